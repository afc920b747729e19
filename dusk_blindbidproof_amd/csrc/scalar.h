// Scalar arithmetic mod l = 2^252 + 27742317777372353535851937790883648493 for gfx950.
//
// 8 saturated 32-bit limbs, canonical (< l) between operations; products go through two 8x8 CIOS
// Montgomery passes (R = 2^256) so every stored scalar is in plain form and byte-compatible with the
// 32-byte little-endian encoding the reference uses on the wire and in the transcript.
//
// Replaces the role of curve25519-dalek 1.2.3 `Scalar` (un-vendored):
// from_bytes_mod_order_wide (src/blindbid/mod.rs:16), from_bits (src/blindbid/bid.rs:27,
// src/blindbid/verify.rs:115), invert, + - * as used throughout bulletproofs' r1cs prover/verifier.
#pragma once
#include "field.h"

namespace bbp {

struct sc {
    u32 v[8];
};

#define BBP_SC_LIT(a0, a1, a2, a3, a4, a5, a6, a7) \
    sc { { a0, a1, a2, a3, a4, a5, a6, a7 } }

BBP_HD sc sc_zero() { return BBP_SC_LIT(0, 0, 0, 0, 0, 0, 0, 0); }
BBP_HD sc sc_one() { return BBP_SC_LIT(1, 0, 0, 0, 0, 0, 0, 0); }
BBP_HD sc sc_l() { return BBP_SC_LIT(0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0, 0, 0, 0x10000000u); }
BBP_HD sc sc_rr() { return BBP_SC_LIT(0x449c0f01u, 0xa40611e3u, 0x68859347u, 0xd00e1ba7u, 0x17f5be65u, 0xceec73d2u, 0x7c309a3du, 0x0399411bu); }
BBP_HD sc sc_r() { return BBP_SC_LIT(0x8d98951du, 0xd6ec3174u, 0x737dcf70u, 0xc6ef5bf4u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0x0fffffffu); }
#define BBP_SC_LFACTOR 0x12547e1bu

BBP_HD sc sc_from_u32(u32 x) { return BBP_SC_LIT(x, 0, 0, 0, 0, 0, 0, 0); }
BBP_HD sc sc_from_u64(u64 x) { return BBP_SC_LIT((u32)x, (u32)(x >> 32), 0, 0, 0, 0, 0, 0); }

// r = a - l if a >= l (a < 2l, optional 9th limb `hi`)
BBP_HD sc sc_cond_sub_l(const sc& a, u32 hi) {
    sc l = sc_l(), t;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)l.v[i];
        t.v[i] = (u32)c;
        c >>= 32;
    }
    c += hi;  // borrow (-1) cancels against hi == 1
    u32 keep_a = (u32)(c < 0);
    u32 m = 0u - keep_a;
    sc r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (a.v[i] & m) | (t.v[i] & ~m);
    return r;
}

BBP_HD sc sc_add(const sc& a, const sc& b) {
    sc r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)a.v[i] + b.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    return sc_cond_sub_l(r, (u32)c);
}

BBP_HD sc sc_sub(const sc& a, const sc& b) {
    sc r, l = sc_l();
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)b.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    u32 m = (u32)c;  // 0 or 0xffffffff
    u64 k = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        k += (u64)r.v[i] + (l.v[i] & m);
        r.v[i] = (u32)k;
        k >>= 32;
    }
    return r;
}

BBP_HD sc sc_neg(const sc& a) { return sc_sub(sc_zero(), a); }

// Montgomery product a*b*2^-256 mod l; needs a*b < 2^256 * l; result canonical
BBP_HD sc sc_montmul(const sc& a, const sc& b) {
    const sc l = sc_l();
    u32 t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (u64)a.v[j] * b.v[i] + t[j];
            t[j] = (u32)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (u32)c;
        t[9] = (u32)(c >> 32);
        u32 m = t[0] * BBP_SC_LFACTOR;
        c = (u64)m * l.v[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (u64)m * l.v[j] + t[j];
            t[j - 1] = (u32)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (u32)c;
        c >>= 32;
        t[8] = t[9] + (u32)c;
    }
    sc r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return sc_cond_sub_l(r, t[8]);
}

BBP_HD sc sc_to_mont(const sc& a) { return sc_montmul(a, sc_rr()); }
BBP_HD sc sc_from_mont(const sc& a) { return sc_montmul(a, sc_one()); }
BBP_HD sc sc_mul(const sc& a, const sc& b) { return sc_montmul(sc_montmul(a, b), sc_rr()); }
BBP_HD sc sc_sq(const sc& a) { return sc_mul(a, a); }
// a*b + c
BBP_HD sc sc_muladd(const sc& a, const sc& b, const sc& c) { return sc_add(sc_mul(a, b), c); }

// any 256-bit value (8 LE words) -> canonical scalar (a < 2^256 < 16 l is allowed into montmul)
BBP_HD sc sc_reduce256(const u32* w) {
    sc a;
#pragma unroll
    for (int i = 0; i < 8; i++) a.v[i] = w[i];
    return sc_montmul(sc_montmul(a, sc_rr()), sc_one());
}

// Scalar::from_bytes_mod_order_wide: 16 LE words
BBP_HD sc sc_from_wide(const u32* w) {
    sc lo, hi;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        lo.v[i] = w[i];
        hi.v[i] = w[i + 8];
    }
    sc lo_mod = sc_montmul(sc_montmul(lo, sc_rr()), sc_one());  // lo mod l
    sc hi_r = sc_montmul(hi, sc_rr());                         // hi * 2^256 mod l
    return sc_add(lo_mod, hi_r);
}

// Scalar::from_bits semantics as the reference uses them (bit 255 cleared, then used mod l)
BBP_HD sc sc_from_bits(const u32* w) {
    u32 t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = w[i];
    t[7] &= 0x7fffffffu;
    return sc_reduce256(t);
}

BBP_HD bool sc_is_canonical(const u32* w) {
    sc l = sc_l();
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)w[i] - (int64_t)l.v[i];
        c >>= 32;
    }
    return c < 0;
}

BBP_HD bool sc_iszero(const sc& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}

BBP_HD bool sc_eq(const sc& a, const sc& b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// a^(l-2): square-and-multiply over the constant exponent, in Montgomery form (kept as the cross-check of sc_invert)
BBP_HD_NOINLINE sc sc_invert_fermat(const sc& a) {
    const u32 e[8] = {0x5cf5d3ebu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0, 0, 0, 0x10000000u};
    sc am = sc_to_mont(a);
    sc acc = sc_r();  // 1 in Montgomery form
    for (int i = 255; i >= 0; i--) {
        acc = sc_montmul(acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1u) acc = sc_montmul(acc, am);
    }
    return sc_from_mont(acc);
}

// Modular inverse by the Bernstein-Yang "safegcd" divsteps (the formulation libsecp256k1's modinv32 popularised), branch-free:
// 20 batches of 30 divsteps on the low words build a 2x2 transition matrix each, applied to (f, g) and, modulo l, to (d, e);
// 600 divsteps are enough for any 256-bit input.  About 12 k instructions against 80 k for the Fermat ladder above -- the
// inversion of each round's challenge sits on the prover's serial path.  Numbers are 9 signed limbs of 30 bits.
// inverse of 0 is 0 (like dalek's Scalar::invert on zero, which the Fermat ladder also yields).
struct sc_s30 {
    i32 v[9];
};

BBP_HD sc sc_invert(const sc& a) {
    const i32 M30 = (i32)((1u << 30) - 1u);
    const i32 MOD[9] = {0x1cf5d3ed, 0x20498c69, 0x2f79cd65, 0x37be77a8, 0x14, 0, 0, 0, 0x1000};
    const u32 MOD_INV30 = 0x2dab81e5u;  // l^-1 mod 2^30
    sc_s30 d, e, f, g;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        d.v[i] = 0;
        e.v[i] = i == 0;
        f.v[i] = MOD[i];
        // bits [30 i, 30 i + 30) of the 256-bit input
        const int o = 30 * i, w = o >> 5, sh = o & 31;
        u64 two = (u64)a.v[w] | (w + 1 < 8 ? (u64)a.v[w + 1] << 32 : 0ull);
        g.v[i] = (i32)((u32)(two >> sh) & (u32)M30);
    }
    i32 zeta = -1;
    for (int it = 0; it < 20; it++) {
        // 30 divsteps on the low limbs -> transition matrix (u v; q r), scaled by 2^30
        u32 u = 1, v = 0, q = 0, r = 1;
        u32 ff = (u32)f.v[0] | ((u32)f.v[1] << 30), gg = (u32)g.v[0] | ((u32)g.v[1] << 30);
        for (int i = 0; i < 30; i++) {
            u32 c1 = (u32)(zeta >> 31), c2 = 0u - (gg & 1u);
            u32 x = (ff ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
            gg += x & c2;
            q += y & c2;
            r += z & c2;
            c1 &= c2;
            zeta = (i32)((u32)zeta ^ c1) - 1;
            ff += gg & c1;
            u += q & c1;
            v += r & c1;
            gg >>= 1;
            u <<= 1;
            v <<= 1;
        }
        const i64 tu = (i32)u, tv = (i32)v, tq = (i32)q, tr = (i32)r;
        // (d, e) <- (u d + v e, q d + r e) / 2^30 mod l, staying in (-2 l, l)
        {
            const i32 sd = d.v[8] >> 31, se = e.v[8] >> 31;
            i32 md = ((i32)tu & sd) + ((i32)tv & se), me = ((i32)tq & sd) + ((i32)tr & se);
            i64 cd = tu * d.v[0] + tv * e.v[0], ce = tq * d.v[0] + tr * e.v[0];
            md -= (i32)((MOD_INV30 * (u32)cd + (u32)md) & (u32)M30);
            me -= (i32)((MOD_INV30 * (u32)ce + (u32)me) & (u32)M30);
            cd += (i64)MOD[0] * md;
            ce += (i64)MOD[0] * me;
            cd >>= 30;
            ce >>= 30;
#pragma unroll
            for (int i = 1; i < 9; i++) {
                const i64 di = d.v[i], ei = e.v[i];
                cd += tu * di + tv * ei + (i64)MOD[i] * md;
                ce += tq * di + tr * ei + (i64)MOD[i] * me;
                d.v[i - 1] = (i32)cd & M30;
                cd >>= 30;
                e.v[i - 1] = (i32)ce & M30;
                ce >>= 30;
            }
            d.v[8] = (i32)cd;
            e.v[8] = (i32)ce;
        }
        // (f, g) <- (u f + v g, q f + r g) / 2^30, exact
        {
            i64 cf = tu * f.v[0] + tv * g.v[0], cg = tq * f.v[0] + tr * g.v[0];
            cf >>= 30;
            cg >>= 30;
#pragma unroll
            for (int i = 1; i < 9; i++) {
                const i64 fi = f.v[i], gi = g.v[i];
                cf += tu * fi + tv * gi;
                cg += tq * fi + tr * gi;
                f.v[i - 1] = (i32)cf & M30;
                cf >>= 30;
                g.v[i - 1] = (i32)cg & M30;
                cg >>= 30;
            }
            f.v[8] = (i32)cf;
            g.v[8] = (i32)cg;
        }
    }
    // now g = 0 and f = +-gcd = +-1 (or +-l for input 0): the inverse is d * sign(f), brought from (-2 l, l) to [0, l)
    {
        i32 cond_add = d.v[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; i++) d.v[i] += MOD[i] & cond_add;
        const i32 cond_neg = f.v[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; i++) d.v[i] = (d.v[i] ^ cond_neg) - cond_neg;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            d.v[i + 1] += d.v[i] >> 30;
            d.v[i] &= M30;
        }
        cond_add = d.v[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; i++) d.v[i] += MOD[i] & cond_add;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            d.v[i + 1] += d.v[i] >> 30;
            d.v[i] &= M30;
        }
    }
    sc out;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        // word w = bits [32 w, 32 w + 32): from limbs floor(32 w / 30) and the next
        const int o = 32 * w, li = o / 30, sh = o % 30;
        u64 two = (u64)(u32)d.v[li] | (li + 1 < 9 ? (u64)(u32)d.v[li + 1] << 30 : 0ull);
        if (li + 2 < 9) two |= (u64)(u32)d.v[li + 2] << 60;
        out.v[w] = (u32)(two >> sh);
    }
    return out;
}

BBP_HD void sc_tobytes(uint8_t* out, const sc& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        out[4 * i + 0] = (uint8_t)(a.v[i]);
        out[4 * i + 1] = (uint8_t)(a.v[i] >> 8);
        out[4 * i + 2] = (uint8_t)(a.v[i] >> 16);
        out[4 * i + 3] = (uint8_t)(a.v[i] >> 24);
    }
}

// Width-WID non-adjacent form of a 256-bit little-endian value < 2^253 (the MSM kernels' scalar recoding), least significant
// digit first: calls f(position, |digit|, negative) for every digit; digits are odd, |digit| < 2^(WID-1), consecutive positions
// at least WID apart, last position <= 253, on average one digit per WID + 1 bits.
// An "effective" bit is bit + carry: runs where bit == carry are skipped with one find-first-set.
template <int WID, class F>
BBP_HD void sc_for_each_naf_digit(const u32 (&s)[8], F&& f) {
    u32 carry = 0;
    int off = 0;  // bit offset into the current 32-bit word (may run past it)
#pragma unroll
    for (int wi = 0; wi < 8; wi++) {
        const u64 win = (u64)s[wi] | (wi + 1 < 8 ? (u64)s[wi + 1] << 32 : 0ull);
        while (off < 32) {
            const u64 rest = (win ^ (carry ? ~0ull : 0ull)) >> off;
            if (rest == 0) {
                off = 64;
                break;
            }
            off += __builtin_ctzll(rest);
            if (off >= 32) break;
            const u32 t = ((u32)(win >> off) & ((1u << WID) - 1u)) + carry;  // odd, <= 2^WID - 1
            carry = t >> (WID - 1);
            f((u32)(32 * wi + off), carry ? (1u << WID) - t : t, carry);
            off += WID;
        }
        off -= 32;
    }
}

}  // namespace bbp
