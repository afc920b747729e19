// Test-only shim: compiles the PRODUCT's limb arithmetic headers for the host CPU so the not-gpu test
// tier can compare them with the oracle.  Nothing in the shipped library calls this.
#include <string.h>

#include <vector>

#include "../dusk_blindbidproof_amd/csrc/keccak.h"
#include "../dusk_blindbidproof_amd/csrc/keccak_wave.h"
#include "../dusk_blindbidproof_amd/csrc/point.h"
#include "../dusk_blindbidproof_amd/csrc/scalar.h"
#include "../dusk_blindbidproof_amd/csrc/witness.h"

using namespace bbp;

static void ld(u32* w, const uint8_t* b, int nwords) { memcpy(w, b, 4 * nwords); }

extern "C" {

// op: 0 add, 1 sub, 2 mul, 3 sq, 4 invert, 5 canon(a), 6 neg, 7 pow22523, 8 mul_small(a, b[0]), 9 sq2, 10 growth, 11 / 12 threaded-carry multiply
void hc_fe_op(int op, const uint8_t* a32, const uint8_t* b32, uint8_t* out32) {
    u32 wa[8], wb[8];
    ld(wa, a32, 8);
    ld(wb, b32, 8);
    fe a = fe_fromwords(wa), b = fe_fromwords(wb), r;  // bit 255 ignored
    switch (op) {
        case 0: r = fe_add(a, b); break;
        case 1: r = fe_sub(a, b); break;
        case 2: r = fe_mul(a, b); break;
        case 3: r = fe_sq(a); break;
        case 4: r = fe_invert(a); break;
        case 5: r = a; break;
        case 6: r = fe_neg(a); break;
        case 7: r = fe_pow22523(a); break;
        case 8: r = fe_mul_small(a, wb[0] & 0x3ffffffu); break;
        case 9: r = fe_sq2(a); break;
        case 10: {  // worst-case operand growth the point formulas produce: three-term sums into a multiply, twice
            fe m1 = fe_mul(a, b), m2 = fe_sq(b), m3 = fe_mul(b, fe_sq(a));
            fe s = fe_add(fe_add(m1, m1), m2), d = fe_sub(fe_sub(m3, m2), m1);
            r = fe_mul(s, d);
            r = fe_sq(fe_sub(fe_add(r, m1), m3));
            break;
        }
        case 11: r = fe_mul_chain_portable(a, b); break;  // the device path's threaded carry pass (field.h), column sums in plain C
        case 12: {  // the same on maximally loaded operands (three-term sums, as the point formulas feed them), limb bounds checked
            fe m1 = fe_mul_chain_portable(a, b), m2 = fe_mul_chain_portable(b, b), m3 = fe_mul_chain_portable(b, fe_mul_chain_portable(a, a));
            fe s = fe_add(fe_add(m1, m1), m2), d = fe_sub(fe_sub(m3, m2), m1);
            r = fe_mul_chain_portable(s, d);
            for (int i = 0; i < 10; i++) {  // carried: |h_even| <= 2^25, |h_odd| <= 2^24 (+ the one small carry into limb 1)
                const i32 bound = (i & 1) ? (1 << 24) : (1 << 25);
                const i32 v = r.v[i] < 0 ? -r.v[i] : r.v[i];
                if (v > bound + (i == 1 ? (1 << 17) : 0)) r = fe_zero();  // reported as a wrong value
            }
            fe t = fe_sub(fe_add(r, m1), m3);
            r = fe_mul_chain_portable(t, t);
            break;
        }
        default: r = fe_zero(); break;
    }
    fe_tobytes(out32, r);
}

// NAF recoding used by the MSM kernels: writes (position, signed digit) pairs, returns the count (width 12 or 9)
int hc_sc_naf(int width, const uint8_t* a32, int32_t* pos_out, int32_t* digit_out) {
    u32 w[8];
    ld(w, a32, 8);
    int n = 0;
    auto emit = [&](u32 pos, u32 mag, u32 neg) {
        pos_out[n] = (int32_t)pos;
        digit_out[n] = neg ? -(int32_t)mag : (int32_t)mag;
        n++;
    };
    if (width == 12) sc_for_each_naf_digit<12>(w, emit);
    else sc_for_each_naf_digit<9>(w, emit);
    return n;
}

// op: 0 add, 1 sub, 2 mul, 3 invert, 4 from_wide(a64), 5 from_bits(a32), 6 neg
void hc_sc_op(int op, const uint8_t* a, const uint8_t* b32, uint8_t* out32) {
    sc x, y, r;
    u32 w[16];
    if (op == 4) {
        ld(w, a, 16);
        r = sc_from_wide(w);
    } else if (op == 5) {
        ld(w, a, 8);
        r = sc_from_bits(w);
    } else {
        ld(x.v, a, 8);
        ld(y.v, b32, 8);
        switch (op) {
            case 0: r = sc_add(x, y); break;
            case 1: r = sc_sub(x, y); break;
            case 2: r = sc_mul(x, y); break;
            case 3: r = sc_invert(x); break;
            case 7: r = sc_invert_fermat(x); break;
            default: r = sc_neg(x); break;
        }
    }
    sc_tobytes(out32, r);
}

int hc_sc_is_canonical(const uint8_t* a32) {
    u32 w[8];
    ld(w, a32, 8);
    return sc_is_canonical(w) ? 1 : 0;
}

// decode -> (op) -> encode.  op: 0 identity map (round trip), 1 double, 2 add(a,b), 3 sub(a,b),
// 4 madd(a, niels(b)), 5 msub(a, niels(b)).  Returns 0 if a decode failed.
int hc_ge_op(int op, const uint8_t* a32, const uint8_t* b32, uint8_t* out32) {
    u32 w[8];
    ge a, b, r;
    ld(w, a32, 8);
    if (!ge_decode_words(a, w)) return 0;
    ld(w, b32, 8);
    if (op >= 2 && !ge_decode_words(b, w)) return 0;
    switch (op) {
        case 0: r = a; break;
        case 1: r = ge_dbl(a); break;
        case 2: r = ge_add(a, b); break;
        case 3: r = ge_sub(a, b); break;
        case 4: r = ge_madd(a, ge_to_niels(b, fe_invert(b.Z))); break;
        default: r = ge_msub(a, ge_to_niels(b, fe_invert(b.Z))); break;
    }
    ge_encode(out32, r);
    return 1;
}

void hc_from_uniform(const uint8_t* in64, uint8_t* out32) {
    u32 w[16];
    ld(w, in64, 16);
    ge_encode(out32, ge_from_uniform_words(w));
}

void hc_basepoint(uint8_t* out32) { ge_encode(out32, ge_basepoint()); }

// scalar * point by double-and-add (test helper)
int hc_scalarmult(const uint8_t* s32, const uint8_t* p32, uint8_t* out32) {
    u32 w[8], s[8];
    ge p;
    ld(w, p32, 8);
    if (!ge_decode_words(p, w)) return 0;
    ld(s, s32, 8);
    ge acc = ge_identity();
    for (int i = 255; i >= 0; i--) {
        acc = ge_dbl(acc);
        if ((s[i >> 5] >> (i & 31)) & 1u) acc = ge_add(acc, p);
    }
    ge_encode(out32, acc);
    return 1;
}

// The prover's gates a_L | a_R | a_O of one proof on the host, from the product's own code (csrc/witness.h): the compiled gadget
// program interpreted (circuit.h compile + witness_gates_interpret) and the gadget wiring written out (witness_gates_native).
// in_raw: d, k, y, y_inv, q, z_img, seed (7 x 32 B) || N items (32 B each) || toggle (u64), as Proof::prove's inputs reach the engine;
// mimc: the 90 round constants (32 B each).  out_*: 3 * n_mul scalars each (a_L, then a_R, then a_O).  Returns n_mul, or -1 / -2 when
// the two forms do not count the same multipliers / the buffers are too small.
int hc_witness_gates(uint32_t n_items, const uint8_t* in_raw, const uint8_t* mimc, uint8_t* out_interp, uint8_t* out_native, uint32_t cap_mul) {
    const circuit::Compiled c = circuit::compile(n_items);
    if (c.n_mul > cap_mul) return -2;
    std::vector<sc> cst(circuit::cst_count(n_items)), v(4 + n_items);
    u32 w[8];
    sc s7[7];
    for (int i = 0; i < 7; i++) {
        ld(w, in_raw + 32 * i, 8);
        s7[i] = sc_reduce256(w);
    }
    cst[circuit::CST_ONE] = sc_one();
    cst[circuit::CST_ZERO] = sc_zero();
    for (int i = 0; i < circuit::MIMC_ROUNDS; i++) {
        ld(w, mimc + 32 * i, 8);
        cst[circuit::CST_MIMC0 + i] = sc_reduce256(w);
    }
    cst[circuit::CST_SEED] = s7[6];
    cst[circuit::CST_ZIMG] = s7[5];
    cst[circuit::CST_Q] = s7[4];
    for (uint32_t i = 0; i < n_items; i++) {
        ld(w, in_raw + 224 + 32 * i, 8);
        cst[circuit::CST_ITEM0 + i] = sc_from_bits(w);  // bid.rs:27
    }
    uint64_t toggle;
    memcpy(&toggle, in_raw + 224 + 32 * (size_t)n_items, 8);
    v[0] = s7[0];
    v[1] = s7[1];
    v[2] = s7[2];
    v[3] = s7[3];
    for (uint32_t i = 0; i < n_items; i++) v[4 + i] = (uint64_t)i == toggle ? sc_one() : sc_zero();
    std::vector<sc> a(3 * (size_t)c.n_mul), b(3 * (size_t)c.n_mul);
    witness_gates_interpret(c.n_mul, c.w_terms.data(), c.w_loff.data(), c.w_roff.data(), cst.data(), v.data(), a.data(), a.data() + c.n_mul,
                            a.data() + 2 * (size_t)c.n_mul);
    const u32 wrote = witness_gates_native(n_items, cst.data(), v.data(), b.data(), b.data() + c.n_mul, b.data() + 2 * (size_t)c.n_mul);
    if (wrote != c.n_mul) return -1;
    for (size_t i = 0; i < a.size(); i++) {
        memcpy(out_interp + 32 * i, a[i].v, 32);
        memcpy(out_native + 32 * i, b[i].v, 32);
    }
    return (int)c.n_mul;
}

// The bit-interleaved form the one-wavefront Keccak keeps its words in (keccak_wave.h): even / odd bits of x as two 32-bit halves.
// 1 when the halves join back to x AND rotl64(x, r) is what the halves give under the rule kw_setup derives its lane shifts from
// (even r: both halves rotate by r / 2; odd r: the halves change places, odd -> even by (r + 1) / 2, even -> odd by (r - 1) / 2).
int hc_kw_interleave(uint64_t x, int r) {
    const u32 e = kw_half(x, 0), o = kw_half(x, 1);
    if (kw_join(e, o) != x) return 0;
    auto rotl32 = [](u32 v, int k) { k &= 31; return k ? (u32)((v << k) | (v >> (32 - k))) : v; };
    u32 e2, o2;
    if (r % 2 == 0) {
        e2 = rotl32(e, r / 2);
        o2 = rotl32(o, r / 2);
    } else {
        e2 = rotl32(o, (r + 1) / 2);
        o2 = rotl32(e, (r - 1) / 2);
    }
    const uint64_t want = r ? (x << r) | (x >> (64 - r)) : x;
    return kw_join(e2, o2) == want ? 1 : 0;
}

// Host MODEL of the one-wavefront Keccak (keccak_wave.h BBP_KW_ROUND, instruction for instruction on arrays of 64 lanes): the lane
// tables of kw_setup / kw_iota_setup are the product's own, the cross-lane operations are restated from the ISA (DPP row shifts with
// bound_ctrl / bank masks, v_permlane16_swap, v_permlane32_swap, ds_bpermute).  What the CPU tier can say about the kernel: the
// layout, the shift amounts, the gather addresses and the order of operations permute like Keccak-f[1600].
struct KwVec {
    u32 v[64];
};
static KwVec kw_dpp_shl(const KwVec& a, int n) {  // row_shl:n bound_ctrl: lane i reads lane i + n of its 16-lane row, 0 past the row
    KwVec r;
    for (int L = 0; L < 64; L++) r.v[L] = (L & 15) + n < 16 ? a.v[L + n] : 0u;
    return r;
}
static KwVec kw_dpp_shr(const KwVec& a, int n) {
    KwVec r;
    for (int L = 0; L < 64; L++) r.v[L] = (L & 15) - n >= 0 ? a.v[L - n] : 0u;
    return r;
}
static u32 kw_rotr32(u32 x, u32 s) { s &= 31; return s ? (x >> s) | (x << (32 - s)) : x; }  // v_alignbit_b32 x, x, s
void hc_kw_keccak_f(uint8_t* st200) {
    u64 st[25];
    memcpy(st, st200, 200);
    kw_lane c[64];
    kw_iota k[64];
    KwVec x;
    for (u32 L = 0; L < 64; L++) {
        c[L] = kw_setup(L);
        k[L] = kw_iota_setup(L);
        x.v[L] = c[L].live ? kw_half(st[c[L].word], c[L].half) : 0u;
    }
    for (int r = 0; r < 24; r++) {
        KwVec t0, t1, t2, t3, t4;
        for (int L = 0; L < 64; L++) t0.v[L] = x.v[L] ^ k[L].v[r];                      // a = x ^ pending iota
        const KwVec s5 = kw_dpp_shl(x, 5), s10 = kw_dpp_shl(x, 10), r5 = kw_dpp_shr(x, 5), r10 = kw_dpp_shr(x, 10);
        for (int L = 0; L < 64; L++) {
            t1.v[L] = x.v[L] ^ s5.v[L] ^ r10.v[L];
            t2.v[L] = s10.v[L];
            t3.v[L] = r5.v[L];
            t4.v[L] = t1.v[L] ^ t2.v[L] ^ t3.v[L];
            t1.v[L] = t4.v[L];
        }
        // v_permlane16_swap t1, t4: rows (a0, a1, a2, a3), (b0, b1, b2, b3) -> (a0, b0, a2, b2), (a1, b1, a3, b3)
        KwVec A = t1, Bv = t4;
        for (int L = 0; L < 16; L++) {
            t1.v[16 + L] = Bv.v[L];       t4.v[L] = A.v[16 + L];
            t1.v[48 + L] = Bv.v[32 + L];  t4.v[32 + L] = A.v[48 + L];
        }
        for (int L = 0; L < 64; L++) {
            t2.v[L] = t1.v[L] ^ t4.v[L] ^ k[L].cp[r];                                  // C of this lane's half
            t3.v[L] = kw_rotr32(t2.v[L], c[L].sh_theta);
        }
        KwVec um = kw_dpp_shr(t2, 1), nx = kw_dpp_shl(t3, 1);
        for (int L = 0; L < 64; L++) {
            const int pos = L & 15;
            if (pos < 4) um.v[L] = t2.v[L + 4];   // row_shl:4 bank_mask:0x1 (source always inside the row)
            if (pos >= 8) nx.v[L] = t3.v[L - 4];  // row_shr:4 bank_mask:0xc
        }
        // v_permlane32_swap t4 (= nx), t2 (= copy): (lo, hi), (lo', hi') -> (lo, lo'), (hi, hi')
        KwVec P = nx, Q = nx;
        for (int L = 0; L < 32; L++) {
            P.v[32 + L] = nx.v[L];  // vdst upper half <- src lower half
            Q.v[L] = nx.v[32 + L];  // src lower half <- vdst upper half
        }
        for (int L = 0; L < 64; L++) {
            const u32 d = um.v[L] ^ nx.v[L] ^ P.v[L] ^ Q.v[L];
            t0.v[L] = kw_rotr32(t0.v[L] ^ (d & c[L].live), c[L].sh_rho);
        }
        for (int L = 0; L < 64; L++) {
            const u32 b0 = t0.v[c[L].s0 / 4], b1 = t0.v[c[L].s1 / 4], b2 = t0.v[c[L].s2 / 4];
            x.v[L] = b0 ^ (~b1 & b2);
        }
    }
    for (u32 L = 0; L < 32; L++)
        if (c[L].live) st[c[L].word] = kw_join(x.v[L] ^ k[L].v[24], x.v[L + 32] ^ k[L + 32].v[24]);
    memcpy(st200, st, 200);
}

void hc_keccak_f(uint8_t* st200) {
    u64 s[25];
    memcpy(s, st200, 200);
    keccak_f1600(s);
    memcpy(st200, s, 200);
}

// merlin: Transcript(label) ; append_message(l1, m1) ; challenge_bytes(l2, n)
void hc_merlin_kat(const uint8_t* label, int label_len, const uint8_t* l1, int l1_len, const uint8_t* m1, int m1_len,
                   const uint8_t* l2, int l2_len, uint8_t* out, int n) {
    merlin_transcript t;
    merlin_init(t, label, label_len);
    merlin_append(t, l1, l1_len, m1, m1_len);
    merlin_challenge(t, l2, l2_len, out, n);
}

// bulk 64-byte draws vs the generic byte-wise path: out_generic / out_bulk = (1 + count + 1) * 64 bytes each
int hc_merlin_rng_bulk(const uint8_t* w, int w_len, const uint8_t* ent32, int count, uint8_t* out_generic, uint8_t* out_bulk) {
    merlin_transcript t;
    merlin_init(t, (const uint8_t*)"BlindBidProofGadget", 19);
    merlin_transcript a = t;
    merlin_rng_rekey(a, (const uint8_t*)"v_blinding", 10, w, w_len);
    merlin_rng_finalize(a, ent32);
    merlin_transcript b = a;
    for (int i = 0; i < count + 2; i++) merlin_rng_fill(a, out_generic + 64 * i, 64);
    merlin_rng_fill(b, out_bulk, 64);
    u32* words = new u32[16 * count];
    bool ok = merlin_rng_fill64_bulk(b, count, words);
    memcpy(out_bulk + 64, words, 64 * (size_t)count);
    delete[] words;
    merlin_rng_fill(b, out_bulk + 64 * (count + 1), 64);
    return ok ? 1 : 0;
}

// TranscriptRng: Transcript(label); rng = build_rng().rekey(wl, w).finalize(ent32); fill n bytes twice
void hc_merlin_rng(const uint8_t* label, int label_len, const uint8_t* wl, int wl_len, const uint8_t* w, int w_len,
                   const uint8_t* ent32, uint8_t* out, int n) {
    merlin_transcript t;
    merlin_init(t, label, label_len);
    merlin_transcript r = t;
    merlin_rng_rekey(r, wl, wl_len, w, w_len);
    merlin_rng_finalize(r, ent32);
    merlin_rng_fill(r, out, n);
    merlin_rng_fill(r, out + n, n);
}
}
