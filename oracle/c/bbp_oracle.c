/* ORACLE -- test infrastructure and timed CPU baseline ONLY; never linked into or called by the shipped library.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Plain-C restatement (64-bit limbs, serial "AVX2 off" arithmetic, one proof per thread) of the reference hot path:
 *   in-tree:   src/gadgets.rs (entire), src/blindbid/mod.rs:7-40, src/blindbid/proof.rs:36-91,
 *              src/blindbid/verify.rs:47-89, src/blindbid/bid.rs:20-29
 *   un-vendored crates (not under /root/reference; algorithms restated from SURVEY.md App. A / RFC 9496):
 *              bulletproofs git develop@4a05305 v1.0.4 (r1cs Prover/Verifier, InnerProductProof, generators),
 *              curve25519-dalek 1.2.3 (Scalar, FieldElement, RistrettoPoint, Straus / Pippenger),
 *              merlin 1.3.0 (STROBE-128 transcript), sha2 0.8.0, sha3 0.8.2.
 * Like the reference it folds the generator vectors every IPA round (the GPU engine does NOT: an independent route to
 * the same group elements).  Pinned against the big-int Python oracle and the golden fixtures in tests/golden/
 * (tests/test_oracle_c.py); the reference itself holds no vectors: parity unpinned by the reference.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;
typedef unsigned __int128 u128;

/* ======================================================== field 2^255-19, radix 2^51 ==================== */
typedef struct { u64 v[5]; } fe;
#define M51 ((1ull << 51) - 1)

static const fe FE_ZERO = {{0, 0, 0, 0, 0}}, FE_ONE = {{1, 0, 0, 0, 0}};

static fe fe_frombytes(const u8 *s) {
    u64 w[4];
    memcpy(w, s, 32);
    fe r;
    r.v[0] = w[0] & M51;
    r.v[1] = ((w[0] >> 51) | (w[1] << 13)) & M51;
    r.v[2] = ((w[1] >> 38) | (w[2] << 26)) & M51;
    r.v[3] = ((w[2] >> 25) | (w[3] << 39)) & M51;
    r.v[4] = (w[3] >> 12) & M51; /* drops bit 255 */
    return r;
}

static fe fe_carry(fe a) {
    for (int pass = 0; pass < 2; pass++) {
        u64 c;
        c = a.v[0] >> 51; a.v[0] &= M51; a.v[1] += c;
        c = a.v[1] >> 51; a.v[1] &= M51; a.v[2] += c;
        c = a.v[2] >> 51; a.v[2] &= M51; a.v[3] += c;
        c = a.v[3] >> 51; a.v[3] &= M51; a.v[4] += c;
        c = a.v[4] >> 51; a.v[4] &= M51; a.v[0] += 19 * c;
    }
    return a;
}

static void fe_tobytes(u8 *out, fe a) {
    a = fe_carry(a);
    /* now each limb < 2^51 + small; compute a + 19 to detect a >= p */
    u64 q = (a.v[0] + 19) >> 51;
    q = (a.v[1] + q) >> 51;
    q = (a.v[2] + q) >> 51;
    q = (a.v[3] + q) >> 51;
    q = (a.v[4] + q) >> 51;
    a.v[0] += 19 * q;
    u64 c;
    c = a.v[0] >> 51; a.v[0] &= M51; a.v[1] += c;
    c = a.v[1] >> 51; a.v[1] &= M51; a.v[2] += c;
    c = a.v[2] >> 51; a.v[2] &= M51; a.v[3] += c;
    c = a.v[3] >> 51; a.v[3] &= M51; a.v[4] += c;
    a.v[4] &= M51;
    u64 w[4];
    w[0] = a.v[0] | (a.v[1] << 51);
    w[1] = (a.v[1] >> 13) | (a.v[2] << 38);
    w[2] = (a.v[2] >> 26) | (a.v[3] << 25);
    w[3] = (a.v[3] >> 39) | (a.v[4] << 12);
    memcpy(out, w, 32);
}

static fe fe_add(fe a, fe b) {
    fe r;
    for (int i = 0; i < 5; i++) r.v[i] = a.v[i] + b.v[i];
    return fe_carry(r);
}

static fe fe_sub(fe a, fe b) {
    /* add 4p so limbs stay non-negative (inputs carried: limbs < 2^52) */
    fe r;
    r.v[0] = a.v[0] + 0x1fffffffffffb4ull - b.v[0];
    for (int i = 1; i < 5; i++) r.v[i] = a.v[i] + 0x1ffffffffffffcull - b.v[i];
    return fe_carry(r);
}

static fe fe_neg(fe a) { return fe_sub(FE_ZERO, a); }

static fe fe_mul(fe a, fe b) {
    u128 t0, t1, t2, t3, t4;
    u64 b1_19 = b.v[1] * 19, b2_19 = b.v[2] * 19, b3_19 = b.v[3] * 19, b4_19 = b.v[4] * 19;
    t0 = (u128)a.v[0] * b.v[0] + (u128)a.v[1] * b4_19 + (u128)a.v[2] * b3_19 + (u128)a.v[3] * b2_19 + (u128)a.v[4] * b1_19;
    t1 = (u128)a.v[0] * b.v[1] + (u128)a.v[1] * b.v[0] + (u128)a.v[2] * b4_19 + (u128)a.v[3] * b3_19 + (u128)a.v[4] * b2_19;
    t2 = (u128)a.v[0] * b.v[2] + (u128)a.v[1] * b.v[1] + (u128)a.v[2] * b.v[0] + (u128)a.v[3] * b4_19 + (u128)a.v[4] * b3_19;
    t3 = (u128)a.v[0] * b.v[3] + (u128)a.v[1] * b.v[2] + (u128)a.v[2] * b.v[1] + (u128)a.v[3] * b.v[0] + (u128)a.v[4] * b4_19;
    t4 = (u128)a.v[0] * b.v[4] + (u128)a.v[1] * b.v[3] + (u128)a.v[2] * b.v[2] + (u128)a.v[3] * b.v[1] + (u128)a.v[4] * b.v[0];
    fe r;
    u64 c;
    r.v[0] = (u64)t0 & M51; c = (u64)(t0 >> 51); t1 += c;
    r.v[1] = (u64)t1 & M51; c = (u64)(t1 >> 51); t2 += c;
    r.v[2] = (u64)t2 & M51; c = (u64)(t2 >> 51); t3 += c;
    r.v[3] = (u64)t3 & M51; c = (u64)(t3 >> 51); t4 += c;
    r.v[4] = (u64)t4 & M51; c = (u64)(t4 >> 51);
    r.v[0] += c * 19;
    c = r.v[0] >> 51; r.v[0] &= M51; r.v[1] += c;
    return r;
}

static fe fe_sq(fe a) { return fe_mul(a, a); }

static fe fe_sqn(fe a, int n) {
    while (n--) a = fe_sq(a);
    return a;
}

static void fe_pow_prefix(fe z, fe *t250, fe *z11) {
    fe z2 = fe_sq(z), z9 = fe_mul(fe_sqn(z2, 2), z);
    *z11 = fe_mul(z9, z2);
    fe a5 = fe_mul(fe_sq(*z11), z9);
    fe a10 = fe_mul(fe_sqn(a5, 5), a5);
    fe a20 = fe_mul(fe_sqn(a10, 10), a10);
    fe a40 = fe_mul(fe_sqn(a20, 20), a20);
    fe a50 = fe_mul(fe_sqn(a40, 10), a10);
    fe a100 = fe_mul(fe_sqn(a50, 50), a50);
    fe a200 = fe_mul(fe_sqn(a100, 100), a100);
    *t250 = fe_mul(fe_sqn(a200, 50), a50);
}

static fe fe_invert(fe z) {
    fe t, z11;
    fe_pow_prefix(z, &t, &z11);
    return fe_mul(fe_sqn(t, 5), z11);
}

static fe fe_pow22523(fe z) {
    fe t, z11;
    fe_pow_prefix(z, &t, &z11);
    return fe_mul(fe_sqn(t, 2), z);
}

static int fe_iszero(fe a) {
    u8 b[32];
    fe_tobytes(b, a);
    u8 o = 0;
    for (int i = 0; i < 32; i++) o |= b[i];
    return o == 0;
}
static int fe_eq(fe a, fe b) { return fe_iszero(fe_sub(a, b)); }
static int fe_isneg(fe a) {
    u8 b[32];
    fe_tobytes(b, a);
    return b[0] & 1;
}
static fe fe_abs(fe a) { return fe_isneg(a) ? fe_neg(a) : a; }

static fe fe_from_hex_le(const char *hex) { /* 64 hex chars, little-endian bytes */
    u8 b[32];
    for (int i = 0; i < 32; i++) {
        unsigned v;
        sscanf(hex + 2 * i, "%2x", &v);
        b[i] = (u8)v;
    }
    return fe_frombytes(b);
}

static fe K_D, K_D2, K_SQRT_M1, K_SQRT_AD_MINUS_ONE, K_INVSQRT_A_MINUS_D, K_ONE_MINUS_D_SQ, K_D_MINUS_ONE_SQ;

/* RFC 9496 4.2 */
static int fe_sqrt_ratio_m1(fe *r, fe u, fe v) {
    fe v3 = fe_mul(fe_sq(v), v), v7 = fe_mul(fe_sq(v3), v);
    fe x = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
    fe check = fe_mul(v, fe_sq(x));
    fe nu = fe_neg(u);
    int correct = fe_eq(check, u), flipped = fe_eq(check, nu), flipped_i = fe_eq(check, fe_mul(nu, K_SQRT_M1));
    if (flipped || flipped_i) x = fe_mul(x, K_SQRT_M1);
    *r = fe_abs(x);
    return correct || flipped;
}

/* ======================================================== scalars mod l, 4x64 Montgomery ================ */
typedef struct { u64 v[4]; } sc;
static const sc SC_L = {{0x5812631a5cf5d3edull, 0x14def9dea2f79cd6ull, 0, 0x1000000000000000ull}};
static const sc SC_RR = {{0xa40611e3449c0f01ull, 0xd00e1ba768859347ull, 0xceec73d217f5be65ull, 0x0399411b7c309a3dull}};
static const sc SC_ZERO = {{0, 0, 0, 0}}, SC_ONE = {{1, 0, 0, 0}};
#define SC_LFACTOR 0xd2b51da312547e1bull

static sc sc_cond_sub(sc a, u64 hi) {
    sc t;
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.v[i] - SC_L.v[i] - (u64)bw;
        t.v[i] = (u64)d;
        bw = (d >> 64) & 1;
    }
    if (hi >= (u64)bw) return t; /* no net borrow */
    return a;
}

static sc sc_add(sc a, sc b) {
    sc r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.v[i] + b.v[i];
        r.v[i] = (u64)c;
        c >>= 64;
    }
    return sc_cond_sub(r, (u64)c);
}

static sc sc_sub(sc a, sc b) {
    sc r;
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.v[i] - b.v[i] - (u64)bw;
        r.v[i] = (u64)d;
        bw = (d >> 64) & 1;
    }
    if (bw) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)r.v[i] + SC_L.v[i];
            r.v[i] = (u64)c;
            c >>= 64;
        }
    }
    return r;
}
static sc sc_neg(sc a) { return sc_sub(SC_ZERO, a); }

static sc sc_montmul(sc a, sc b) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.v[j] * b.v[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (u64)c;
        t[5] = (u64)(c >> 64);
        u64 m = t[0] * SC_LFACTOR;
        c = (u128)m * SC_L.v[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * SC_L.v[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (u64)c;
        t[4] = t[5] + (u64)(c >> 64);
    }
    sc r = {{t[0], t[1], t[2], t[3]}};
    return sc_cond_sub(r, t[4]);
}

static sc sc_mul(sc a, sc b) { return sc_montmul(sc_montmul(a, b), SC_RR); }
static sc sc_reduce256(sc a) { return sc_montmul(sc_montmul(a, SC_RR), SC_ONE); }
static sc sc_frombytes_raw(const u8 *b) {
    sc r;
    memcpy(r.v, b, 32);
    return r;
}
static sc sc_from_wide(const u8 *b64) {
    sc lo = sc_frombytes_raw(b64), hi = sc_frombytes_raw(b64 + 32);
    return sc_add(sc_reduce256(lo), sc_montmul(hi, SC_RR));
}
static sc sc_from_bits(const u8 *b32) { /* src/blindbid/bid.rs:27: bit 255 cleared; used mod l */
    sc a = sc_frombytes_raw(b32);
    a.v[3] &= 0x7fffffffffffffffull;
    return sc_reduce256(a);
}
static int sc_canonical(const u8 *b32) {
    sc a = sc_frombytes_raw(b32);
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.v[i] - SC_L.v[i] - (u64)bw;
        bw = (d >> 64) & 1;
    }
    return (int)bw;
}
static void sc_tobytes(u8 *o, sc a) { memcpy(o, a.v, 32); }
static sc sc_from_u64(u64 x) {
    sc r = {{x, 0, 0, 0}};
    return r;
}
static int sc_iszero(sc a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static sc sc_invert(sc a) {
    static const u64 e[4] = {0x5812631a5cf5d3ebull, 0x14def9dea2f79cd6ull, 0, 0x1000000000000000ull};
    sc am = sc_montmul(a, SC_RR), acc = sc_montmul(SC_ONE, SC_RR);
    for (int i = 255; i >= 0; i--) {
        acc = sc_montmul(acc, acc);
        if ((e[i >> 6] >> (i & 63)) & 1) acc = sc_montmul(acc, am);
    }
    return sc_montmul(acc, SC_ONE);
}

/* ======================================================== points ======================================== */
typedef struct { fe X, Y, Z, T; } ge;
static ge GE_IDENT, GE_BASE;

static ge ge_add(ge p, ge q) {
    fe a = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    fe b = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    fe c = fe_mul(fe_mul(p.T, q.T), K_D2);
    fe zz = fe_mul(p.Z, q.Z);
    fe d = fe_add(zz, zz);
    fe e = fe_sub(b, a), f = fe_sub(d, c), g = fe_add(d, c), h = fe_add(b, a);
    ge r = {fe_mul(e, f), fe_mul(g, h), fe_mul(f, g), fe_mul(e, h)};
    return r;
}
static ge ge_neg(ge p) {
    ge r = {fe_neg(p.X), p.Y, p.Z, fe_neg(p.T)};
    return r;
}
static ge ge_dbl(ge p) {
    fe a = fe_sq(p.X), b = fe_sq(p.Y), zz = fe_sq(p.Z);
    fe c = fe_add(zz, zz), h = fe_add(a, b);
    fe e = fe_sub(h, fe_sq(fe_add(p.X, p.Y))), g = fe_sub(a, b), f = fe_add(c, g);
    ge r = {fe_mul(e, f), fe_mul(g, h), fe_mul(f, g), fe_mul(e, h)};
    return r;
}
static int ge_is_identity(ge p) { return fe_iszero(p.X) || fe_iszero(p.Y); }

static void ge_encode(u8 *out, ge p) { /* RFC 9496 4.3.2 */
    fe u1 = fe_mul(fe_add(p.Z, p.Y), fe_sub(p.Z, p.Y)), u2 = fe_mul(p.X, p.Y), inv;
    fe_sqrt_ratio_m1(&inv, FE_ONE, fe_mul(u1, fe_sq(u2)));
    fe den1 = fe_mul(inv, u1), den2 = fe_mul(inv, u2);
    fe z_inv = fe_mul(fe_mul(den1, den2), p.T);
    fe ix0 = fe_mul(p.X, K_SQRT_M1), iy0 = fe_mul(p.Y, K_SQRT_M1), ench = fe_mul(den1, K_INVSQRT_A_MINUS_D);
    fe x, y, den_inv;
    if (fe_isneg(fe_mul(p.T, z_inv))) {
        x = iy0; y = ix0; den_inv = ench;
    } else {
        x = p.X; y = p.Y; den_inv = den2;
    }
    if (fe_isneg(fe_mul(x, z_inv))) y = fe_neg(y);
    fe_tobytes(out, fe_abs(fe_mul(den_inv, fe_sub(p.Z, y))));
}

static int ge_decode(ge *out, const u8 *b) { /* RFC 9496 4.3.1 */
    fe s = fe_frombytes(b);
    u8 chk[32];
    fe_tobytes(chk, s);
    if (memcmp(chk, b, 32) != 0 || (b[0] & 1)) return 0;
    fe ss = fe_sq(s), u1 = fe_sub(FE_ONE, ss), u2 = fe_add(FE_ONE, ss), u2s = fe_sq(u2);
    fe v = fe_sub(fe_neg(fe_mul(K_D, fe_sq(u1))), u2s), inv;
    int sq = fe_sqrt_ratio_m1(&inv, FE_ONE, fe_mul(v, u2s));
    fe den_x = fe_mul(inv, u2), den_y = fe_mul(fe_mul(inv, den_x), v);
    fe x = fe_abs(fe_mul(fe_add(s, s), den_x)), y = fe_mul(u1, den_y), t = fe_mul(x, y);
    if (!sq || fe_isneg(t) || fe_iszero(y)) return 0;
    out->X = x; out->Y = y; out->Z = FE_ONE; out->T = t;
    return 1;
}

static ge ge_elligator(fe t) { /* RFC 9496 4.3.4 MAP */
    fe r = fe_mul(K_SQRT_M1, fe_sq(t));
    fe u = fe_mul(fe_add(r, FE_ONE), K_ONE_MINUS_D_SQ);
    fe v = fe_mul(fe_sub(fe_neg(FE_ONE), fe_mul(r, K_D)), fe_add(r, K_D));
    fe s;
    int sq = fe_sqrt_ratio_m1(&s, u, v);
    fe s_prime = fe_neg(fe_abs(fe_mul(s, t)));
    fe c = r;
    if (sq) c = fe_neg(FE_ONE); else s = s_prime;
    fe n = fe_sub(fe_mul(fe_mul(c, fe_sub(r, FE_ONE)), K_D_MINUS_ONE_SQ), v);
    fe w0 = fe_mul(fe_add(s, s), v), w1 = fe_mul(n, K_SQRT_AD_MINUS_ONE), ss = fe_sq(s);
    fe w2 = fe_sub(FE_ONE, ss), w3 = fe_add(FE_ONE, ss);
    ge p = {fe_mul(w0, w3), fe_mul(w2, w1), fe_mul(w1, w3), fe_mul(w0, w2)};
    return p;
}
static ge ge_from_uniform(const u8 *b64) { return ge_add(ge_elligator(fe_frombytes(b64)), ge_elligator(fe_frombytes(b64 + 32))); }

/* signed radix-2^w digits of a canonical scalar; returns digit count */
static int sc_digits(const sc *s, int w, int16_t *out) {
    int nd = (256 + w - 1) / w + 1;
    int carry = 0;
    for (int j = 0; j < nd; j++) {
        int o = j * w, word = o >> 6, sh = o & 63;
        u64 raw = 0;
        if (word < 4) {
            raw = s->v[word] >> sh;
            if (sh + w > 64 && word + 1 < 4) raw |= s->v[word + 1] << (64 - sh);
        }
        int d = (int)(raw & ((1u << w) - 1)) + carry;
        carry = d > (1 << (w - 1));
        if (carry) d -= 1 << w;
        out[j] = (int16_t)d;
    }
    return nd;
}

/* vartime multiscalar multiplication: Straus (radix 16) below 190 points, Pippenger above (dalek's split, A.2) */
static ge ge_msm(int n, const sc *scalars, const ge *points) {
    if (n == 0) return GE_IDENT;
    int16_t *dig;
    if (n < 190) {
        const int w = 4;
        int nd = 0;
        dig = malloc(sizeof(int16_t) * (size_t)n * 66);
        ge *tab = malloc(sizeof(ge) * (size_t)n * 8);
        for (int i = 0; i < n; i++) {
            nd = sc_digits(&scalars[i], w, dig + (size_t)i * 66);
            tab[(size_t)i * 8] = points[i];
            for (int k = 1; k < 8; k++) tab[(size_t)i * 8 + k] = ge_add(tab[(size_t)i * 8 + k - 1], points[i]);
        }
        ge acc = GE_IDENT;
        for (int j = nd - 1; j >= 0; j--) {
            if (j != nd - 1) for (int k = 0; k < w; k++) acc = ge_dbl(acc);
            for (int i = 0; i < n; i++) {
                int d = dig[(size_t)i * 66 + j];
                if (d > 0) acc = ge_add(acc, tab[(size_t)i * 8 + d - 1]);
                else if (d < 0) acc = ge_add(acc, ge_neg(tab[(size_t)i * 8 - d - 1]));
            }
        }
        free(dig);
        free(tab);
        return acc;
    }
    int w = n < 500 ? 6 : n < 800 ? 7 : 8;
    int stride = (256 + w - 1) / w + 2;
    dig = malloc(sizeof(int16_t) * (size_t)n * stride);
    int nd = 0;
    for (int i = 0; i < n; i++) nd = sc_digits(&scalars[i], w, dig + (size_t)i * stride);
    int nb = 1 << (w - 1);
    ge *buckets = malloc(sizeof(ge) * (size_t)nb);
    u8 *used = malloc((size_t)nb);
    ge acc = GE_IDENT;
    for (int j = nd - 1; j >= 0; j--) {
        for (int k = 0; k < w; k++) acc = ge_dbl(acc);
        memset(used, 0, (size_t)nb);
        for (int i = 0; i < n; i++) {
            int d = dig[(size_t)i * stride + j];
            if (d == 0) continue;
            int b = (d > 0 ? d : -d) - 1;
            ge p = d > 0 ? points[i] : ge_neg(points[i]);
            if (used[b]) buckets[b] = ge_add(buckets[b], p);
            else { buckets[b] = p; used[b] = 1; }
        }
        ge run = GE_IDENT, tot = GE_IDENT;
        for (int b = nb - 1; b >= 0; b--) {
            if (used[b]) run = ge_add(run, buckets[b]);
            tot = ge_add(tot, run);
        }
        acc = ge_add(acc, tot);
    }
    free(dig); free(buckets); free(used);
    return acc;
}

static ge ge_mul(sc s, ge p) { return ge_msm(1, &s, &p); }

/* ======================================================== hashes ======================================== */
static u64 rotl64(u64 x, int n) { return (x << n) | (x >> (64 - n)); }
static void keccak_f(u64 *s) {
    static const u64 RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
        0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull,
        0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull,
        0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull,
        0x800000008000000Aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int r = 0; r < 24; r++) {
        u64 bc[5], t;
        for (int i = 0; i < 5; i++) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
        for (int i = 0; i < 5; i++) {
            t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) s[j + i] ^= t;
        }
        t = s[1];
        for (int i = 0; i < 24; i++) {
            int j = PIL[i];
            u64 b = s[j];
            s[j] = rotl64(t, ROT[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = s[j + i];
            for (int i = 0; i < 5; i++) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        s[0] ^= RC[r];
    }
}

static void sponge(const u8 *msg, size_t len, unsigned rate, u8 suffix, u8 *out, size_t outlen) {
    u64 st[25];
    memset(st, 0, sizeof st);
    u8 *sb = (u8 *)st;
    size_t pos = 0;
    for (size_t i = 0; i < len; i++) {
        sb[pos++] ^= msg[i];
        if (pos == rate) { keccak_f(st); pos = 0; }
    }
    sb[pos] ^= suffix;
    sb[rate - 1] ^= 0x80;
    keccak_f(st);
    pos = 0;
    for (size_t i = 0; i < outlen; i++) {
        if (pos == rate) { keccak_f(st); pos = 0; }
        out[i] = sb[pos++];
    }
}

static void sha512(const u8 *msg, size_t len, u8 out[64]) {
    static const u64 K[80] = {
        0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull,
        0x59f111f1b605d019ull, 0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull,
        0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull, 0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull,
        0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull, 0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull,
        0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull, 0x983e5152ee66dfabull,
        0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
        0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull,
        0x53380d139d95b3dfull, 0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull,
        0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull, 0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull,
        0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull, 0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull,
        0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull, 0x5b9cca4f7763e373ull,
        0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
        0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull,
        0xd186b8c721c0c207ull, 0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull,
        0x113f9804bef90daeull, 0x1b710b35131c471bull, 0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull,
        0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull, 0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};
    u64 h[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    size_t padded = ((len + 17 + 127) / 128) * 128;
    u8 *m = calloc(padded, 1);
    memcpy(m, msg, len);
    m[len] = 0x80;
    u64 bits = (u64)len * 8;
    for (int i = 0; i < 8; i++) m[padded - 1 - i] = (u8)(bits >> (8 * i));
#define ROTR(x, n) (((x) >> (n)) | ((x) << (64 - (n))))
    for (size_t off = 0; off < padded; off += 128) {
        u64 w[80];
        for (int i = 0; i < 16; i++) {
            w[i] = 0;
            for (int j = 0; j < 8; j++) w[i] = (w[i] << 8) | m[off + 8 * i + j];
        }
        for (int i = 16; i < 80; i++) {
            u64 s0 = ROTR(w[i - 15], 1) ^ ROTR(w[i - 15], 8) ^ (w[i - 15] >> 7);
            u64 s1 = ROTR(w[i - 2], 19) ^ ROTR(w[i - 2], 61) ^ (w[i - 2] >> 6);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        u64 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 80; i++) {
            u64 t1 = hh + (ROTR(e, 14) ^ ROTR(e, 18) ^ ROTR(e, 41)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            u64 t2 = (ROTR(a, 28) ^ ROTR(a, 34) ^ ROTR(a, 39)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    free(m);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) out[8 * i + j] = (u8)(h[i] >> (56 - 8 * j));
}

/* ======================================================== merlin (A.1) ================================== */
typedef struct { u64 st[25]; u32 pos, pos_begin, cur_flags; } strobe;
#define SR 166
static void st_runf(strobe *s) {
    u8 *b = (u8 *)s->st;
    b[s->pos] ^= (u8)s->pos_begin;
    b[s->pos + 1] ^= 0x04;
    b[SR + 1] ^= 0x80;
    keccak_f(s->st);
    s->pos = 0; s->pos_begin = 0;
}
static void st_absorb(strobe *s, const u8 *d, size_t n) {
    u8 *b = (u8 *)s->st;
    for (size_t i = 0; i < n; i++) { b[s->pos++] ^= d[i]; if (s->pos == SR) st_runf(s); }
}
static void st_overwrite(strobe *s, const u8 *d, size_t n) {
    u8 *b = (u8 *)s->st;
    for (size_t i = 0; i < n; i++) { b[s->pos++] = d[i]; if (s->pos == SR) st_runf(s); }
}
static void st_squeeze(strobe *s, u8 *d, size_t n) {
    u8 *b = (u8 *)s->st;
    for (size_t i = 0; i < n; i++) { d[i] = b[s->pos]; b[s->pos++] = 0; if (s->pos == SR) st_runf(s); }
}
static void st_begin(strobe *s, u32 flags, int more) {
    if (more) return;
    u8 hdr[2] = {(u8)s->pos_begin, (u8)flags};
    s->pos_begin = s->pos + 1;
    s->cur_flags = flags;
    st_absorb(s, hdr, 2);
    if ((flags & (4 | 32)) && s->pos != 0) st_runf(s);
}
static void st_meta_ad(strobe *s, const void *d, size_t n, int more) { st_begin(s, 16 | 2, more); st_absorb(s, d, n); }
static void st_ad(strobe *s, const void *d, size_t n, int more) { st_begin(s, 2, more); st_absorb(s, d, n); }
static void st_prf(strobe *s, u8 *d, size_t n, int more) { st_begin(s, 1 | 2 | 4, more); st_squeeze(s, d, n); }
static void st_key(strobe *s, const void *d, size_t n, int more) { st_begin(s, 2 | 4, more); st_overwrite(s, d, n); }

static void tr_append(strobe *t, const char *label, const void *msg, u32 len) {
    st_meta_ad(t, label, strlen(label), 0);
    st_meta_ad(t, &len, 4, 1);
    st_ad(t, msg, len, 0);
}
static void tr_append_u64(strobe *t, const char *label, u64 x) { tr_append(t, label, &x, 8); }
static void tr_init(strobe *t, const char *label) {
    memset(t, 0, sizeof *t);
    static const u8 hdr[18] = {1, 168, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    memcpy(t->st, hdr, 18);
    keccak_f(t->st);
    st_meta_ad(t, "Merlin v1.0", 11, 0);
    tr_append(t, "dom-sep", label, (u32)strlen(label));
}
static void tr_challenge(strobe *t, const char *label, u8 *out, u32 n) {
    st_meta_ad(t, label, strlen(label), 0);
    st_meta_ad(t, &n, 4, 1);
    st_prf(t, out, n, 0);
}
static sc tr_challenge_scalar(strobe *t, const char *label) {
    u8 b[64];
    tr_challenge(t, label, b, 64);
    return sc_from_wide(b);
}
static void tr_point(strobe *t, const char *label, const u8 *p32) { tr_append(t, label, p32, 32); }
static int tr_validate_point(strobe *t, const char *label, const u8 *p32) {
    static const u8 zero[32] = {0};
    if (memcmp(p32, zero, 32) == 0) return 0;
    tr_append(t, label, p32, 32);
    return 1;
}
static void tr_scalar(strobe *t, const char *label, sc s) {
    u8 b[32];
    sc_tobytes(b, s);
    tr_append(t, label, b, 32);
}
static void rng_rekey(strobe *r, const char *label, const void *w, u32 len) {
    st_meta_ad(r, label, strlen(label), 0);
    st_meta_ad(r, &len, 4, 1);
    st_key(r, w, len, 0);
}
static void rng_finalize(strobe *r, const u8 *ent32) {
    st_meta_ad(r, "rng", 3, 0);
    st_key(r, ent32, 32, 0);
}
static sc rng_scalar(strobe *r) {
    u8 b[64];
    u32 n = 64;
    st_meta_ad(r, &n, 4, 0);
    st_prf(r, b, 64, 0);
    return sc_from_wide(b);
}

/* ======================================================== setup ========================================= */
#define CAP_MAX 2048
#define ROUNDS_MAX 90
static ge G_B, G_BBLIND, G_G[CAP_MAX], G_H[CAP_MAX];
static sc MIMC_C[ROUNDS_MAX];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void setup_once(void) {
    K_D = fe_from_hex_le("a3785913ca4deb75abd841414d0a700098e879777940c78c73fe6f2bee6c0352");
    K_D2 = fe_add(K_D, K_D);
    K_SQRT_M1 = fe_from_hex_le("b0a00e4a271beec478e42fad0618432fa7d7fb3d99004d2b0bdfc14f8024832b");
    K_SQRT_AD_MINUS_ONE = fe_from_hex_le("1b2e7b49a0f6977ebd54781b0c8e9daffdd1f531c9fc3c0fac48832bbf316937");
    K_INVSQRT_A_MINUS_D = fe_from_hex_le("ea405d80aafdc899be72415a17162f9d40d801fe917bc216a2fcafcf05896c78");
    K_ONE_MINUS_D_SQ = fe_sub(FE_ONE, fe_sq(K_D));
    K_D_MINUS_ONE_SQ = fe_sq(fe_sub(K_D, FE_ONE));
    GE_IDENT.X = FE_ZERO; GE_IDENT.Y = FE_ONE; GE_IDENT.Z = FE_ONE; GE_IDENT.T = FE_ZERO;
    static const u8 B_ENC[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                 0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
    ge_decode(&GE_BASE, B_ENC); /* RFC 9496 generator encoding */
    G_B = GE_BASE;
    u8 h[64];
    sponge(B_ENC, 32, 72, 0x06, h, 64); /* SHA3-512 */
    G_BBLIND = ge_from_uniform(h);
    u8 *stream = malloc(64 * CAP_MAX);
    for (int which = 0; which < 2; which++) {
        u8 seed[20] = "GeneratorsChain";
        seed[15] = which ? 'H' : 'G';
        seed[16] = seed[17] = seed[18] = seed[19] = 0;
        sponge(seed, 20, 136, 0x1f, stream, 64 * CAP_MAX); /* SHAKE256 */
        for (int i = 0; i < CAP_MAX; i++) (which ? G_H : G_G)[i] = ge_from_uniform(stream + 64 * i);
    }
    free(stream);
    sha512((const u8 *)"blind bid", 9, h); /* src/blindbid/mod.rs:11-20 */
    for (int i = 0; i < ROUNDS_MAX; i++) {
        MIMC_C[i] = sc_from_wide(h);
        u8 cb[32];
        sc_tobytes(cb, MIMC_C[i]);
        sha512(cb, 32, h);
    }
}
static void setup(void) { pthread_once(&g_once, setup_once); }

/* ======================================================== constraint system (A.3) ======================= */
enum { V_COMMITTED, V_L, V_R, V_O, V_ONE };
typedef struct { u32 kind, idx; sc c; } term;
typedef struct { term *t; int n, cap; } lc;

static lc lc_empty(void) { lc r = {NULL, 0, 0}; return r; }
static void lc_push(lc *a, u32 kind, u32 idx, sc c) {
    if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 4; a->t = realloc(a->t, sizeof(term) * (size_t)a->cap); }
    a->t[a->n].kind = kind; a->t[a->n].idx = idx; a->t[a->n].c = c; a->n++;
}
static lc lc_clone(const lc *a) {
    lc r = lc_empty();
    for (int i = 0; i < a->n; i++) lc_push(&r, a->t[i].kind, a->t[i].idx, a->t[i].c);
    return r;
}
static lc lc_var(u32 kind, u32 idx) { lc r = lc_empty(); lc_push(&r, kind, idx, SC_ONE); return r; }
static lc lc_const(sc c) { lc r = lc_empty(); lc_push(&r, V_ONE, 0, c); return r; }
static void lc_add_in(lc *a, const lc *b, int negate) {
    for (int i = 0; i < b->n; i++) lc_push(a, b->t[i].kind, b->t[i].idx, negate ? sc_neg(b->t[i].c) : b->t[i].c);
}
static void lc_free(lc *a) { free(a->t); a->t = NULL; a->n = a->cap = 0; }

typedef struct {
    int prover;
    lc *cons; int ncons, capcons;
    int n_mul, m;
    sc *aL, *aR, *aO; int capmul;  /* prover */
    sc v[4 + 256], vb[4 + 256];    /* prover */
} cs_t;

static void cs_constrain(cs_t *cs, lc l) { /* takes ownership */
    if (cs->ncons == cs->capcons) { cs->capcons = cs->capcons ? 2 * cs->capcons : 1024; cs->cons = realloc(cs->cons, sizeof(lc) * (size_t)cs->capcons); }
    cs->cons[cs->ncons++] = l;
}
static sc cs_eval(const cs_t *cs, const lc *l) {
    sc acc = SC_ZERO;
    for (int i = 0; i < l->n; i++) {
        sc val;
        switch (l->t[i].kind) {
            case V_L: val = cs->aL[l->t[i].idx]; break;
            case V_R: val = cs->aR[l->t[i].idx]; break;
            case V_O: val = cs->aO[l->t[i].idx]; break;
            case V_COMMITTED: val = cs->v[l->t[i].idx]; break;
            default: val = SC_ONE; break;
        }
        acc = sc_add(acc, sc_mul(l->t[i].c, val));
    }
    return acc;
}
/* multiply(left, right): returns multiplier index; takes ownership of both LCs */
static u32 cs_multiply(cs_t *cs, lc left, lc right) {
    u32 i = (u32)cs->n_mul++;
    if (cs->prover) {
        if ((int)i == cs->capmul) {
            cs->capmul = cs->capmul ? 2 * cs->capmul : 2048;
            cs->aL = realloc(cs->aL, sizeof(sc) * (size_t)cs->capmul);
            cs->aR = realloc(cs->aR, sizeof(sc) * (size_t)cs->capmul);
            cs->aO = realloc(cs->aO, sizeof(sc) * (size_t)cs->capmul);
        }
        sc l = cs_eval(cs, &left), r = cs_eval(cs, &right);
        cs->aL[i] = l; cs->aR[i] = r; cs->aO[i] = sc_mul(l, r);
    }
    lc_push(&left, V_L, i, sc_neg(SC_ONE));
    lc_push(&right, V_R, i, sc_neg(SC_ONE));
    cs_constrain(cs, left);
    cs_constrain(cs, right);
    return i;
}
static void cs_free(cs_t *cs) {
    for (int i = 0; i < cs->ncons; i++) lc_free(&cs->cons[i]);
    free(cs->cons); free(cs->aL); free(cs->aR); free(cs->aO);
}

/* ---- gadgets: src/gadgets.rs ---------------------------------------------------------------------------- */
static lc mimc_gadget(cs_t *cs, const lc *left, const lc *right, int rounds) { /* gadgets.rs:37-68 */
    lc x = lc_clone(left);
    for (int i = 0; i < rounds; i++) {
        lc a = x; /* x + key + c[i] */
        lc_add_in(&a, right, 0);
        lc_push(&a, V_ONE, 0, MIMC_C[i]);
        u32 m2 = cs_multiply(cs, lc_clone(&a), lc_clone(&a));
        u32 m3 = cs_multiply(cs, lc_var(V_O, m2), lc_clone(&a));
        u32 m4 = cs_multiply(cs, lc_var(V_O, m2), lc_var(V_O, m2));
        u32 m7 = cs_multiply(cs, lc_var(V_O, m4), lc_var(V_O, m3));
        lc_free(&a);
        x = lc_var(V_O, m7);
    }
    lc_add_in(&x, right, 0);
    return x;
}
static void one_of_many_gadget(cs_t *cs, const lc *x, const u32 *toggle, int n, const sc *items) { /* gadgets.rs:88-140 */
    for (int i = 0; i < n; i++) { /* boolean_gadget */
        lc a = lc_var(V_COMMITTED, toggle[i]);
        lc one_minus = lc_const(SC_ONE);
        lc_push(&one_minus, V_COMMITTED, toggle[i], sc_neg(SC_ONE));
        u32 m = cs_multiply(cs, a, one_minus);
        cs_constrain(cs, lc_var(V_O, m));
    }
    /* running sums; the N-1 constraints prev + cur - cur_sum cancel term by term but still consume a power of z */
    for (int i = 1; i < n; i++) {
        lc c = lc_empty();
        for (int k = 0; k < i; k++) lc_push(&c, V_COMMITTED, toggle[k], SC_ONE);      /* prev_toggle_sum */
        lc_push(&c, V_COMMITTED, toggle[i], SC_ONE);                                   /* + curr_toggle */
        for (int k = 0; k <= i; k++) lc_push(&c, V_COMMITTED, toggle[k], sc_neg(SC_ONE)); /* - curr_toggle_sum */
        cs_constrain(cs, c);
    }
    lc last = lc_empty();
    for (int k = 0; k < n; k++) lc_push(&last, V_COMMITTED, toggle[k], SC_ONE);
    lc_push(&last, V_ONE, 0, sc_neg(SC_ONE));
    cs_constrain(cs, last);
    for (int i = 0; i < n; i++) {
        u32 l = cs_multiply(cs, lc_const(items[i]), lc_var(V_COMMITTED, toggle[i]));
        u32 r = cs_multiply(cs, lc_var(V_COMMITTED, toggle[i]), lc_clone(x));
        lc c = lc_var(V_O, l);
        lc_push(&c, V_O, r, sc_neg(SC_ONE));
        cs_constrain(cs, c);
    }
}
static void proof_gadget(cs_t *cs, u32 vd, u32 vk, u32 vyinv, sc q, sc z_img, sc seed, const u32 *toggle, int n,
                         const sc *items, int rounds) { /* gadgets.rs:6-34 */
    lc d = lc_var(V_COMMITTED, vd), k = lc_var(V_COMMITTED, vk), zero = lc_const(SC_ZERO), sd = lc_const(seed);
    lc m = mimc_gadget(cs, &k, &zero, rounds);
    lc x = mimc_gadget(cs, &d, &m, rounds);
    one_of_many_gadget(cs, &x, toggle, n, items);
    lc y = mimc_gadget(cs, &sd, &x, rounds);
    lc z = mimc_gadget(cs, &sd, &m, rounds);
    lc c = lc_const(z_img);
    lc_add_in(&c, &z, 1);
    cs_constrain(cs, c);
    /* score_gadget gadgets.rs:70-86 */
    u32 one_var = cs_multiply(cs, lc_clone(&y), lc_var(V_COMMITTED, vyinv));
    lc c1 = lc_var(V_O, one_var);
    lc_push(&c1, V_ONE, 0, sc_neg(SC_ONE));
    cs_constrain(cs, c1);
    u32 q_var = cs_multiply(cs, lc_clone(&d), lc_var(V_COMMITTED, vyinv));
    lc c2 = lc_const(q);
    lc_push(&c2, V_O, q_var, sc_neg(SC_ONE));
    cs_constrain(cs, c2);
    lc_free(&d); lc_free(&k); lc_free(&zero); lc_free(&sd); lc_free(&m); lc_free(&x); lc_free(&y); lc_free(&z);
}

static void cs_flatten(const cs_t *cs, sc z, sc *wL, sc *wR, sc *wO, sc *wV, sc *wc) { /* A.5 step 7 */
    for (int i = 0; i < cs->n_mul; i++) wL[i] = wR[i] = wO[i] = SC_ZERO;
    for (int i = 0; i < cs->m; i++) wV[i] = SC_ZERO;
    *wc = SC_ZERO;
    sc e = z;
    for (int k = 0; k < cs->ncons; k++) {
        const lc *l = &cs->cons[k];
        for (int i = 0; i < l->n; i++) {
            sc ec = sc_mul(e, l->t[i].c);
            u32 ix = l->t[i].idx;
            switch (l->t[i].kind) {
                case V_L: wL[ix] = sc_add(wL[ix], ec); break;
                case V_R: wR[ix] = sc_add(wR[ix], ec); break;
                case V_O: wO[ix] = sc_add(wO[ix], ec); break;
                case V_COMMITTED: wV[ix] = sc_sub(wV[ix], ec); break;
                default: *wc = sc_sub(*wc, ec); break;
            }
        }
        e = sc_mul(e, z);
    }
}

static sc inner(const sc *a, const sc *b, int n) {
    sc acc = SC_ZERO;
    for (int i = 0; i < n; i++) acc = sc_add(acc, sc_mul(a[i], b[i]));
    return acc;
}
static int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

/* ======================================================== prove (A.4-A.6, proof.rs:36-91) ============== */
#define ST_OK 0
#define ST_VERIFY 1
#define ST_GENS 2
#define ST_FORMAT 3
#define ST_BADARG 4

typedef struct { u8 y[32], z[32], u[32], x[32], w[32], u_ipp[16][32]; int n_mul, n_cons; } oc_trace;

static ge pedersen(sc v, sc vb) {
    sc s[2] = {v, vb};
    ge p[2] = {G_B, G_BBLIND};
    return ge_msm(2, s, p);
}

int oc_prove(const u8 *scalars7, const u8 *pub_list, u32 N, u64 toggle, const u8 *entropy, int rounds, int cap,
             u8 *record_out, u32 *proof_len, oc_trace *trace) {
    setup();
    if (N == 0 || toggle >= N || N > 256 || rounds < 1 || rounds > ROUNDS_MAX || cap > CAP_MAX) return ST_BADARG;
    sc in7[7];
    for (int i = 0; i < 7; i++) in7[i] = sc_reduce256(sc_frombytes_raw(scalars7 + 32 * i));
    sc d = in7[0], k = in7[1], y_ = in7[2], y_inv = in7[3], q = in7[4], z_img = in7[5], seed = in7[6];
    cs_t cs;
    memset(&cs, 0, sizeof cs);
    cs.prover = 1;
    strobe t;
    tr_init(&t, "BlindBidProofGadget");
    tr_append(&t, "dom-sep", "r1cs v1", 7);
    int m = 4 + (int)N;
    cs.m = m;
    u8 *commit_bytes = record_out; /* filled at the end; keep locally first */
    u8 Vb[260][32];
    sc vals[4] = {d, k, y_, y_inv};
    for (int i = 0; i < m; i++) {
        cs.v[i] = i < 4 ? vals[i] : sc_from_u64((u64)(i - 4) == toggle);
        cs.vb[i] = sc_reduce256(sc_frombytes_raw(entropy + 32 * i));
        ge_encode(Vb[i], pedersen(cs.v[i], cs.vb[i]));
        tr_point(&t, "V", Vb[i]);
    }
    (void)commit_bytes;
    sc *items = malloc(sizeof(sc) * N);
    u32 *tog = malloc(sizeof(u32) * N);
    for (u32 i = 0; i < N; i++) { items[i] = sc_from_bits(pub_list + 32 * i); tog[i] = 4 + i; }
    proof_gadget(&cs, 0, 1, 3, q, z_img, seed, tog, (int)N, items, rounds);
    free(items); free(tog);

    /* Prover::prove */
    tr_append_u64(&t, "m", (u64)m);
    strobe rng = t;
    for (int i = 0; i < m; i++) { u8 b[32]; sc_tobytes(b, cs.vb[i]); rng_rekey(&rng, "v_blinding", b, 32); }
    rng_finalize(&rng, entropy + 32 * m);
    int n1 = cs.n_mul;
    if (cap < n1) { cs_free(&cs); return ST_GENS; }
    sc ib = rng_scalar(&rng), ob = rng_scalar(&rng), sb = rng_scalar(&rng);
    int padded = next_pow2(n1), pad = padded - n1;
    if (cap < padded) { cs_free(&cs); return ST_GENS; }
    sc *sL = malloc(sizeof(sc) * (size_t)padded), *sR = malloc(sizeof(sc) * (size_t)padded);
    for (int i = 0; i < n1; i++) sL[i] = rng_scalar(&rng);
    for (int i = 0; i < n1; i++) sR[i] = rng_scalar(&rng);
    sc *ms = malloc(sizeof(sc) * (size_t)(2 * n1 + 1));
    ge *mp = malloc(sizeof(ge) * (size_t)(2 * padded + 1));
    u8 A_I1[32], A_O1[32], S1[32];
    ms[0] = ib; mp[0] = G_BBLIND;
    for (int i = 0; i < n1; i++) { ms[1 + i] = cs.aL[i]; mp[1 + i] = G_G[i]; ms[1 + n1 + i] = cs.aR[i]; mp[1 + n1 + i] = G_H[i]; }
    ge_encode(A_I1, ge_msm(2 * n1 + 1, ms, mp));
    ms[0] = ob;
    for (int i = 0; i < n1; i++) ms[1 + i] = cs.aO[i];
    ge_encode(A_O1, ge_msm(n1 + 1, ms, mp));
    ms[0] = sb;
    for (int i = 0; i < n1; i++) { ms[1 + i] = sL[i]; ms[1 + n1 + i] = sR[i]; }
    ge_encode(S1, ge_msm(2 * n1 + 1, ms, mp));
    tr_point(&t, "A_I1", A_I1); tr_point(&t, "A_O1", A_O1); tr_point(&t, "S1", S1);
    tr_append(&t, "dom-sep", "r1cs-1phase", 11);
    static const u8 ident[32] = {0};
    tr_point(&t, "A_I2", ident); tr_point(&t, "A_O2", ident); tr_point(&t, "S2", ident);
    sc y = tr_challenge_scalar(&t, "y"), z = tr_challenge_scalar(&t, "z");
    sc *wL = malloc(sizeof(sc) * (size_t)padded), *wR = malloc(sizeof(sc) * (size_t)padded), *wO = malloc(sizeof(sc) * (size_t)padded);
    sc wV[260], wc;
    cs_flatten(&cs, z, wL, wR, wO, wV, &wc);
    sc yinv = sc_invert(y);
    sc *Y = malloc(sizeof(sc) * (size_t)(padded + 1)), *Yi = malloc(sizeof(sc) * (size_t)padded);
    Y[0] = SC_ONE; Yi[0] = SC_ONE;
    for (int i = 1; i <= padded; i++) Y[i] = sc_mul(Y[i - 1], y);
    for (int i = 1; i < padded; i++) Yi[i] = sc_mul(Yi[i - 1], yinv);
    sc *l1 = malloc(sizeof(sc) * (size_t)n1), *r0 = malloc(sizeof(sc) * (size_t)n1), *r1 = malloc(sizeof(sc) * (size_t)n1), *r3 = malloc(sizeof(sc) * (size_t)n1);
    for (int i = 0; i < n1; i++) {
        l1[i] = sc_add(cs.aL[i], sc_mul(Yi[i], wR[i]));
        r0[i] = sc_sub(wO[i], Y[i]);
        r1[i] = sc_add(sc_mul(Y[i], cs.aR[i]), wL[i]);
        r3[i] = sc_mul(Y[i], sR[i]);
    }
    const sc *l2 = cs.aO, *l3 = sL;
    sc t1 = inner(l1, r0, n1);
    sc t2 = sc_add(inner(l1, r1, n1), inner(l2, r0, n1));
    sc t3 = sc_add(inner(l2, r1, n1), inner(l3, r0, n1));
    sc t4 = sc_add(inner(l1, r3, n1), inner(l3, r1, n1));
    sc t5 = inner(l2, r3, n1), t6 = inner(l3, r3, n1);
    sc tb1 = rng_scalar(&rng), tb3 = rng_scalar(&rng), tb4 = rng_scalar(&rng), tb5 = rng_scalar(&rng), tb6 = rng_scalar(&rng);
    u8 T1[32], T3[32], T4[32], T5[32], T6[32];
    ge_encode(T1, pedersen(t1, tb1)); ge_encode(T3, pedersen(t3, tb3)); ge_encode(T4, pedersen(t4, tb4));
    ge_encode(T5, pedersen(t5, tb5)); ge_encode(T6, pedersen(t6, tb6));
    tr_point(&t, "T_1", T1); tr_point(&t, "T_3", T3); tr_point(&t, "T_4", T4); tr_point(&t, "T_5", T5); tr_point(&t, "T_6", T6);
    sc u = tr_challenge_scalar(&t, "u"), x = tr_challenge_scalar(&t, "x");
    sc tb2 = inner(wV, cs.vb, m);
    sc xs[7];
    xs[0] = SC_ONE;
    for (int i = 1; i < 7; i++) xs[i] = sc_mul(xs[i - 1], x);
    sc tc[6] = {t1, t2, t3, t4, t5, t6}, tbc[6] = {tb1, tb2, tb3, tb4, tb5, tb6};
    sc t_x = SC_ZERO, t_xb = SC_ZERO;
    for (int i = 0; i < 6; i++) { t_x = sc_add(t_x, sc_mul(tc[i], xs[i + 1])); t_xb = sc_add(t_xb, sc_mul(tbc[i], xs[i + 1])); }
    sc *lv = malloc(sizeof(sc) * (size_t)padded), *rv = malloc(sizeof(sc) * (size_t)padded);
    for (int i = 0; i < n1; i++) {
        lv[i] = sc_add(sc_add(sc_mul(l1[i], xs[1]), sc_mul(l2[i], xs[2])), sc_mul(l3[i], xs[3]));
        rv[i] = sc_add(sc_add(r0[i], sc_mul(r1[i], xs[1])), sc_mul(r3[i], xs[3]));
    }
    for (int i = n1; i < padded; i++) { lv[i] = SC_ZERO; rv[i] = sc_neg(Y[i]); }
    sc e_bl = sc_mul(x, sc_add(ib, sc_mul(x, sc_add(ob, sc_mul(x, sb)))));
    tr_scalar(&t, "t_x", t_x); tr_scalar(&t, "t_x_blinding", t_xb); tr_scalar(&t, "e_blinding", e_bl);
    sc w = tr_challenge_scalar(&t, "w");
    ge Q = ge_mul(w, G_B);
    if (trace) {
        sc_tobytes(trace->y, y); sc_tobytes(trace->z, z); sc_tobytes(trace->u, u); sc_tobytes(trace->x, x); sc_tobytes(trace->w, w);
        trace->n_mul = n1; trace->n_cons = cs.ncons;
    }
    /* InnerProductProof::create (A.6) */
    sc *Gf = malloc(sizeof(sc) * (size_t)padded), *Hf = malloc(sizeof(sc) * (size_t)padded);
    for (int i = 0; i < padded; i++) { Gf[i] = i < n1 ? SC_ONE : u; Hf[i] = sc_mul(Yi[i], Gf[i]); }
    ge *Gv = malloc(sizeof(ge) * (size_t)padded), *Hv = malloc(sizeof(ge) * (size_t)padded);
    memcpy(Gv, G_G, sizeof(ge) * (size_t)padded);
    memcpy(Hv, G_H, sizeof(ge) * (size_t)padded);
    tr_append(&t, "dom-sep", "ipp v1", 6);
    tr_append_u64(&t, "n", (u64)padded);
    u8 LR[32][64];
    int lg = 0, n = padded, first = 1;
    sc *a = lv, *b = rv;
    sc *tmp_s = malloc(sizeof(sc) * (size_t)(padded + 1));
    while (n != 1) {
        n /= 2;
        sc *aL_ = a, *aR_ = a + n, *bL_ = b, *bR_ = b + n;
        ge *GL = Gv, *GR = Gv + n, *HL = Hv, *HR = Hv + n;
        sc cL = inner(aL_, bR_, n), cR = inner(aR_, bL_, n);
        for (int i = 0; i < n; i++) { tmp_s[i] = first ? sc_mul(aL_[i], Gf[n + i]) : aL_[i]; mp[i] = GR[i]; }
        for (int i = 0; i < n; i++) { tmp_s[n + i] = first ? sc_mul(bR_[i], Hf[i]) : bR_[i]; mp[n + i] = HL[i]; }
        tmp_s[2 * n] = cL; mp[2 * n] = Q;
        ge_encode(LR[lg], ge_msm(2 * n + 1, tmp_s, mp));
        for (int i = 0; i < n; i++) { tmp_s[i] = first ? sc_mul(aR_[i], Gf[i]) : aR_[i]; mp[i] = GL[i]; }
        for (int i = 0; i < n; i++) { tmp_s[n + i] = first ? sc_mul(bL_[i], Hf[n + i]) : bL_[i]; mp[n + i] = HR[i]; }
        tmp_s[2 * n] = cR;
        ge_encode(LR[lg] + 32, ge_msm(2 * n + 1, tmp_s, mp));
        tr_point(&t, "L", LR[lg]); tr_point(&t, "R", LR[lg] + 32);
        sc uj = tr_challenge_scalar(&t, "u"), uji = sc_invert(uj);
        if (trace && lg < 16) sc_tobytes(trace->u_ipp[lg], uj);
        for (int i = 0; i < n; i++) {
            aL_[i] = sc_add(sc_mul(aL_[i], uj), sc_mul(uji, aR_[i]));
            bL_[i] = sc_add(sc_mul(bL_[i], uji), sc_mul(uj, bR_[i]));
            sc s2[2];
            ge p2[2];
            s2[0] = first ? sc_mul(uji, Gf[i]) : uji; s2[1] = first ? sc_mul(uj, Gf[n + i]) : uj;
            p2[0] = GL[i]; p2[1] = GR[i];
            GL[i] = ge_msm(2, s2, p2);
            s2[0] = first ? sc_mul(uj, Hf[i]) : uj; s2[1] = first ? sc_mul(uji, Hf[n + i]) : uji;
            p2[0] = HL[i]; p2[1] = HR[i];
            HL[i] = ge_msm(2, s2, p2);
        }
        first = 0;
        lg++;
    }
    /* R1CSProof::to_bytes, 1-phase compact form (A.8) */
    u8 *o = record_out;
    *o++ = 0;
    memcpy(o, A_I1, 32); o += 32; memcpy(o, A_O1, 32); o += 32; memcpy(o, S1, 32); o += 32;
    memcpy(o, T1, 32); o += 32; memcpy(o, T3, 32); o += 32; memcpy(o, T4, 32); o += 32; memcpy(o, T5, 32); o += 32; memcpy(o, T6, 32); o += 32;
    sc_tobytes(o, t_x); o += 32; sc_tobytes(o, t_xb); o += 32; sc_tobytes(o, e_bl); o += 32;
    for (int j = 0; j < lg; j++) { memcpy(o, LR[j], 64); o += 64; }
    sc_tobytes(o, a[0]); o += 32; sc_tobytes(o, b[0]); o += 32;
    if (proof_len) *proof_len = (u32)(o - record_out);
    for (int i = 0; i < m; i++) { memcpy(o, Vb[i], 32); o += 32; }
    free(sL); free(sR); free(ms); free(mp); free(wL); free(wR); free(wO); free(Y); free(Yi); free(l1); free(r0); free(r1); free(r3);
    free(lv); free(rv); free(Gf); free(Hf); free(Gv); free(Hv); free(tmp_s);
    cs_free(&cs);
    return ST_OK;
}

/* ======================================================== verify (A.7, verify.rs:47-89) ================= */
int oc_verify(const u8 *record, u32 record_len, const u8 *score32, const u8 *z_img32, const u8 *seed32,
              const u8 *pub_list, u32 N, int rounds, int cap, const u8 *entropy32) {
    setup();
    if (N == 0 || N > 256 || rounds < 1 || rounds > ROUNDS_MAX || cap > CAP_MAX) return ST_BADARG;
    u32 tail = 32 * (4 + N);
    if (record_len < tail + 1) return ST_FORMAT;
    u32 plen = record_len - tail;
    const u8 *pr = record, *Vs = record + plen;
    /* R1CSProof::from_bytes */
    if (pr[0] != 0 && pr[0] != 1) return ST_FORMAT;
    if ((plen - 1) % 32 != 0) return ST_FORMAT;
    u32 nel = (plen - 1) / 32, npts = pr[0] == 0 ? 3 : 6;
    if (nel < npts + 5 + 3 + 2) return ST_FORMAT;
    const u8 *p = pr + 1;
    const u8 *A_I1 = p, *A_O1 = p + 32, *S1 = p + 64;
    static const u8 ident[32] = {0};
    const u8 *A_I2 = ident, *A_O2 = ident, *S2 = ident;
    p += 96;
    if (pr[0] == 1) { A_I2 = p; A_O2 = p + 32; S2 = p + 64; p += 96; }
    const u8 *T[5];
    for (int i = 0; i < 5; i++) { T[i] = p; p += 32; }
    if (!sc_canonical(p) || !sc_canonical(p + 32) || !sc_canonical(p + 64)) return ST_FORMAT;
    sc t_x = sc_frombytes_raw(p), t_xb = sc_frombytes_raw(p + 32), e_bl = sc_frombytes_raw(p + 64);
    p += 96;
    u32 ipp_el = nel - npts - 8;
    if (ipp_el < 2 || (ipp_el - 2) % 2 != 0) return ST_FORMAT;
    u32 lg_n = (ipp_el - 2) / 2;
    if (lg_n >= 32) return ST_FORMAT;
    const u8 *LRp = p;
    p += 64 * lg_n;
    if (!sc_canonical(p) || !sc_canonical(p + 32)) return ST_FORMAT;
    sc a = sc_frombytes_raw(p), b = sc_frombytes_raw(p + 32);

    cs_t cs;
    memset(&cs, 0, sizeof cs);
    strobe t;
    tr_init(&t, "BlindBidProofGadget");
    tr_append(&t, "dom-sep", "r1cs v1", 7);
    int m = 4 + (int)N;
    cs.m = m;
    for (int i = 0; i < m; i++) tr_point(&t, "V", Vs + 32 * i);
    sc *items = malloc(sizeof(sc) * N);
    u32 *tog = malloc(sizeof(u32) * N);
    for (u32 i = 0; i < N; i++) { items[i] = sc_from_bits(pub_list + 32 * i); tog[i] = 4 + i; }
    sc score = sc_reduce256(sc_frombytes_raw(score32)), z_img = sc_reduce256(sc_frombytes_raw(z_img32)), seed = sc_reduce256(sc_frombytes_raw(seed32));
    proof_gadget(&cs, 0, 1, 3, score, z_img, seed, tog, (int)N, items, rounds);
    free(items); free(tog);
    int rc = ST_VERIFY;
    tr_append_u64(&t, "m", (u64)m);
    int n1 = cs.n_mul, padded = next_pow2(n1), pad = padded - n1;
    sc *wL = NULL, *wR = NULL, *wO = NULL, *s = NULL, *Yi = NULL, *scal = NULL;
    ge *pts = NULL;
    if (!tr_validate_point(&t, "A_I1", A_I1) || !tr_validate_point(&t, "A_O1", A_O1) || !tr_validate_point(&t, "S1", S1)) goto done;
    tr_append(&t, "dom-sep", "r1cs-1phase", 11);
    if (cap < padded) { rc = ST_GENS; goto done; }
    tr_point(&t, "A_I2", A_I2); tr_point(&t, "A_O2", A_O2); tr_point(&t, "S2", S2);
    sc y = tr_challenge_scalar(&t, "y"), z = tr_challenge_scalar(&t, "z");
    static const char *TL[5] = {"T_1", "T_3", "T_4", "T_5", "T_6"};
    for (int i = 0; i < 5; i++) if (!tr_validate_point(&t, TL[i], T[i])) goto done;
    sc u = tr_challenge_scalar(&t, "u"), x = tr_challenge_scalar(&t, "x");
    tr_scalar(&t, "t_x", t_x); tr_scalar(&t, "t_x_blinding", t_xb); tr_scalar(&t, "e_blinding", e_bl);
    sc w = tr_challenge_scalar(&t, "w");
    wL = calloc((size_t)padded, sizeof(sc)); wR = calloc((size_t)padded, sizeof(sc)); wO = calloc((size_t)padded, sizeof(sc));
    sc wV[260], wc;
    cs_flatten(&cs, z, wL, wR, wO, wV, &wc);
    /* verification_scalars */
    if ((u32)padded != (1u << lg_n)) goto done;
    tr_append(&t, "dom-sep", "ipp v1", 6);
    tr_append_u64(&t, "n", (u64)padded);
    sc ch[32], chi[32], chsq[32], chisq[32];
    for (u32 j = 0; j < lg_n; j++) {
        if (!tr_validate_point(&t, "L", LRp + 64 * j) || !tr_validate_point(&t, "R", LRp + 64 * j + 32)) goto done;
        ch[j] = tr_challenge_scalar(&t, "u");
    }
    sc allinv = SC_ONE;
    for (u32 j = 0; j < lg_n; j++) { chi[j] = sc_invert(ch[j]); allinv = sc_mul(allinv, chi[j]); chsq[j] = sc_mul(ch[j], ch[j]); chisq[j] = sc_mul(chi[j], chi[j]); }
    s = malloc(sizeof(sc) * (size_t)padded);
    s[0] = allinv;
    for (int i = 1; i < padded; i++) {
        int lg_i = 31 - __builtin_clz((unsigned)i);
        s[i] = sc_mul(s[i - (1 << lg_i)], chsq[lg_n - 1 - (u32)lg_i]);
    }
    Yi = malloc(sizeof(sc) * (size_t)padded);
    sc yinv = sc_invert(y);
    Yi[0] = SC_ONE;
    for (int i = 1; i < padded; i++) Yi[i] = sc_mul(Yi[i - 1], yinv);
    sc delta = SC_ZERO;
    int nterm = 6 + m + 5 + 2 + 2 * padded + 2 * (int)lg_n;
    scal = malloc(sizeof(sc) * (size_t)nterm);
    pts = malloc(sizeof(ge) * (size_t)nterm);
    strobe rng = t;
    rng_finalize(&rng, entropy32);
    sc r = rng_scalar(&rng);
    sc xx = sc_mul(x, x), rxx = sc_mul(r, xx), xxx = sc_mul(x, xx);
    int k = 0;
    const u8 *pb[6] = {A_I1, A_O1, S1, A_I2, A_O2, S2};
    sc ps[6] = {x, xx, xxx, sc_mul(u, x), sc_mul(u, xx), sc_mul(u, xxx)};
    for (int i = 0; i < 6; i++) { if (!ge_decode(&pts[k], pb[i])) goto done; scal[k++] = ps[i]; }
    for (int i = 0; i < m; i++) { if (!ge_decode(&pts[k], Vs + 32 * i)) goto done; scal[k++] = sc_mul(wV[i], rxx); }
    sc Ts[5] = {sc_mul(r, x), sc_mul(rxx, x), sc_mul(rxx, xx), sc_mul(rxx, xxx), sc_mul(sc_mul(rxx, xx), xx)};
    for (int i = 0; i < 5; i++) { if (!ge_decode(&pts[k], T[i])) goto done; scal[k++] = Ts[i]; }
    int kB = k;
    k += 2;
    for (int i = 0; i < padded; i++) {
        sc ynw = i < n1 ? sc_mul(wR[i], Yi[i]) : SC_ZERO;
        if (i < n1) delta = sc_add(delta, sc_mul(ynw, wL[i]));
        sc uf = i < n1 ? SC_ONE : u;
        pts[k] = G_G[i];
        scal[k++] = sc_mul(uf, sc_sub(sc_mul(x, ynw), sc_mul(a, s[i])));
    }
    for (int i = 0; i < padded; i++) {
        sc uf = i < n1 ? SC_ONE : u;
        sc inner_ = sc_sub(sc_add(sc_mul(x, wL[i]), wO[i]), sc_mul(b, s[padded - 1 - i]));
        pts[k] = G_H[i];
        scal[k++] = sc_mul(uf, sc_sub(sc_mul(Yi[i], inner_), SC_ONE));
    }
    pts[kB] = G_B;
    scal[kB] = sc_add(sc_mul(w, sc_sub(t_x, sc_mul(a, b))), sc_mul(r, sc_sub(sc_mul(xx, sc_add(wc, delta)), t_x)));
    pts[kB + 1] = G_BBLIND;
    scal[kB + 1] = sc_sub(sc_neg(e_bl), sc_mul(r, t_xb));
    for (u32 j = 0; j < lg_n; j++) { if (!ge_decode(&pts[k], LRp + 64 * j)) goto done; scal[k++] = chsq[j]; }
    for (u32 j = 0; j < lg_n; j++) { if (!ge_decode(&pts[k], LRp + 64 * j + 32)) goto done; scal[k++] = chisq[j]; }
    (void)pad;
    rc = ge_is_identity(ge_msm(k, scal, pts)) ? ST_OK : ST_VERIFY;
done:
    free(wL); free(wR); free(wO); free(s); free(Yi); free(scal); free(pts);
    cs_free(&cs);
    return rc;
}

/* ======================================================== helpers exported for tests / bench ============ */
void oc_generator(u32 index, u8 *out32) { /* engine table order: 0 B_blinding, 1.. G, 2049.. H, 4097 B */
    setup();
    ge p = index == 0 ? G_BBLIND : index <= 2048 ? G_G[index - 1] : index <= 4096 ? G_H[index - 2049] : G_B;
    ge_encode(out32, p);
}
void oc_mimc_constant(u32 i, u8 *out32) { setup(); sc_tobytes(out32, MIMC_C[i]); }

static sc mimc_native(sc l, sc r, int rounds) { /* native image of mimc_gadget, gadgets.rs:45-67 */
    sc x = l;
    for (int i = 0; i < rounds; i++) {
        sc a = sc_add(sc_add(x, r), MIMC_C[i]);
        sc a2 = sc_mul(a, a), a3 = sc_mul(a2, a), a4 = sc_mul(a2, a2);
        x = sc_mul(a4, a3);
    }
    return sc_add(x, r);
}
void oc_witness(const u8 *dks96, int rounds, u8 *out192) { /* m,x,y,y_inv,q,z_img */
    setup();
    sc d = sc_reduce256(sc_frombytes_raw(dks96)), k = sc_reduce256(sc_frombytes_raw(dks96 + 32)), seed = sc_reduce256(sc_frombytes_raw(dks96 + 64));
    sc m = mimc_native(k, SC_ZERO, rounds), x = mimc_native(d, m, rounds), y = mimc_native(seed, x, rounds), z = mimc_native(seed, m, rounds);
    sc yi = sc_invert(y), q = sc_mul(d, yi);
    sc_tobytes(out192, m); sc_tobytes(out192 + 32, x); sc_tobytes(out192 + 64, y); sc_tobytes(out192 + 96, yi); sc_tobytes(out192 + 128, q); sc_tobytes(out192 + 160, z);
}

/* MSM over an engine layout (include/bbp.h): 0 = B_blinding,G[0..m),H[0..m) ; 1 = B_blinding,G[0..m) */
int oc_msm_layout(const u8 *scalars, u32 n, u32 layout, u8 *out32) {
    setup();
    u32 m = layout == 0 ? (n - 1) / 2 : n - 1;
    if (n < 1 || m > CAP_MAX || (layout == 0 && n != 1 + 2 * m) || layout > 1) return ST_BADARG;
    sc *s = malloc(sizeof(sc) * n);
    ge *p = malloc(sizeof(ge) * n);
    for (u32 i = 0; i < n; i++) s[i] = sc_reduce256(sc_frombytes_raw(scalars + 32 * i));
    p[0] = G_BBLIND;
    for (u32 i = 0; i < m; i++) { p[1 + i] = G_G[i]; if (layout == 0) p[1 + m + i] = G_H[i]; }
    ge_encode(out32, ge_msm((int)n, s, p));
    free(s); free(p);
    return ST_OK;
}

/* ---- simple thread pool over independent jobs (one proof per thread: dusk-uds worker model) ---- */
typedef struct { int kind; const void *a, *b, *c, *d, *e; u32 n, layout, N; u64 toggle; int rounds, cap; void *out; int rc; } job_t;
typedef struct { job_t *jobs; int njobs; volatile int next; pthread_mutex_t mu; } pool_t;
static void run_job(job_t *j) {
    switch (j->kind) {
        case 0: j->rc = oc_msm_layout(j->a, j->n, j->layout, j->out); break;
        case 1: { u32 pl; j->rc = oc_prove(j->a, j->b, j->N, j->toggle, j->c, j->rounds, j->cap, j->out, &pl, NULL); break; }
        case 2: j->rc = oc_verify(j->a, j->n, j->b, j->c, j->d, j->e, j->N, j->rounds, j->cap, (const u8 *)"\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0"); break;
    }
}
static void *worker(void *arg) {
    pool_t *p = arg;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        int i = p->next < p->njobs ? p->next++ : -1;
        pthread_mutex_unlock(&p->mu);
        if (i < 0) return NULL;
        run_job(&p->jobs[i]);
    }
}
static void run_pool(job_t *jobs, int njobs, int threads) {
    setup();
    pool_t p = {jobs, njobs, 0, PTHREAD_MUTEX_INITIALIZER};
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, worker, &p);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
}
/* rows: concatenated scalars of each MSM (row i has n[i] terms) */
int oc_msm_layout_many(const u8 *rows, const u32 *n, const u32 *layout, int count, int threads, u8 *out32) {
    job_t *jobs = calloc((size_t)count, sizeof(job_t));
    size_t off = 0;
    for (int i = 0; i < count; i++) { jobs[i].kind = 0; jobs[i].a = rows + off; jobs[i].n = n[i]; jobs[i].layout = layout[i]; jobs[i].out = out32 + 32 * i; off += (size_t)32 * n[i]; }
    run_pool(jobs, count, threads);
    int rc = 0;
    for (int i = 0; i < count; i++) rc |= jobs[i].rc;
    free(jobs);
    return rc;
}
/* in: B * (7*32 + N*32 + 8) ; entropy: B * (32*(4+N)+32) ; out: B * (1121 + 32*(4+N)) (stride rec_stride) */
int oc_prove_many(const u8 *in, const u8 *entropy, u32 B, u32 N, int rounds, int cap, int threads, u8 *out, u32 rec_stride, int *status) {
    job_t *jobs = calloc(B, sizeof(job_t));
    size_t in_stride = 7 * 32 + (size_t)N * 32 + 8, ent_stride = 32 * (4 + (size_t)N) + 32;
    for (u32 i = 0; i < B; i++) {
        const u8 *r = in + in_stride * i;
        u64 tg;
        memcpy(&tg, r + 7 * 32 + (size_t)N * 32, 8);
        jobs[i].kind = 1; jobs[i].a = r; jobs[i].b = r + 7 * 32; jobs[i].N = N; jobs[i].toggle = tg; jobs[i].c = entropy + ent_stride * i;
        jobs[i].rounds = rounds; jobs[i].cap = cap; jobs[i].out = out + (size_t)rec_stride * i;
    }
    run_pool(jobs, (int)B, threads);
    for (u32 i = 0; i < B; i++) status[i] = jobs[i].rc;
    free(jobs);
    return 0;
}
/* in: B * (rec_len + 3*32 + N*32): record || score || z_img || seed || pub_list */
int oc_verify_many(const u8 *in, u32 B, u32 N, u32 rec_len, int rounds, int cap, int threads, int *status) {
    job_t *jobs = calloc(B, sizeof(job_t));
    size_t stride = (size_t)rec_len + 96 + (size_t)N * 32;
    for (u32 i = 0; i < B; i++) {
        const u8 *r = in + stride * i;
        jobs[i].kind = 2; jobs[i].a = r; jobs[i].n = rec_len; jobs[i].b = r + rec_len; jobs[i].c = r + rec_len + 32; jobs[i].d = r + rec_len + 64;
        jobs[i].e = r + rec_len + 96; jobs[i].N = N; jobs[i].rounds = rounds; jobs[i].cap = cap;
    }
    run_pool(jobs, (int)B, threads);
    for (u32 i = 0; i < B; i++) status[i] = jobs[i].rc;
    free(jobs);
    return 0;
}

/* primitive hooks for tests/test_oracle_c.py */
void oc_sc_op(int op, const u8 *a, const u8 *b, u8 *out) { /* 0 add 1 sub 2 mul 3 inv 4 wide(a64) 5 from_bits */
    sc r;
    switch (op) {
        case 0: r = sc_add(sc_frombytes_raw(a), sc_frombytes_raw(b)); break;
        case 1: r = sc_sub(sc_frombytes_raw(a), sc_frombytes_raw(b)); break;
        case 2: r = sc_mul(sc_frombytes_raw(a), sc_frombytes_raw(b)); break;
        case 3: r = sc_invert(sc_frombytes_raw(a)); break;
        case 4: r = sc_from_wide(a); break;
        default: r = sc_from_bits(a); break;
    }
    sc_tobytes(out, r);
}
int oc_scalarmult(const u8 *s32, const u8 *p32, u8 *out32) {
    setup();
    ge p;
    if (!ge_decode(&p, p32)) return 0;
    ge_encode(out32, ge_mul(sc_reduce256(sc_frombytes_raw(s32)), p));
    return 1;
}
void oc_from_uniform(const u8 *in64, u8 *out32) { setup(); ge_encode(out32, ge_from_uniform(in64)); }
void oc_merlin_kat(const char *label, const char *l1, const u8 *m1, u32 m1len, const char *l2, u8 *out, u32 n) {
    strobe t;
    tr_init(&t, label);
    tr_append(&t, l1, m1, m1len);
    tr_challenge(&t, l2, out, n);
}
