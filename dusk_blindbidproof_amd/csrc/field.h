// Field arithmetic mod p = 2^255 - 19 for gfx950, written for 32-bit VALU lanes.
//
// Representation: 8 saturated 32-bit limbs, value kept in [0, 2^256) and only congruent mod p
// (2^256 == 38 mod p, so every carry-out folds back as +38).  A field element is 8 VGPRs, a point 32.
// The multiplier is an 8x8 operand scan on v_mad_u64_u32 (32x32+64 -> 64).
//
// Replaces the role of curve25519-dalek 1.2.3 `FieldElement` (un-vendored crate).
// The same header compiles for the host (tests/host_check.cpp) so the limb code can be checked against
// the oracle without a GPU; the shipped library only ever runs it on the device.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BBP_HD __host__ __device__ __forceinline__
#define BBP_HD_NOINLINE __host__ __device__ inline __attribute__((noinline))
#else
#define BBP_HD inline
#define BBP_HD_NOINLINE inline
#endif

namespace bbp {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

struct fe {
    u32 v[8];
};

#define BBP_FE_LIT(a0, a1, a2, a3, a4, a5, a6, a7) \
    fe { { a0, a1, a2, a3, a4, a5, a6, a7 } }

BBP_HD fe fe_zero() { return BBP_FE_LIT(0, 0, 0, 0, 0, 0, 0, 0); }
BBP_HD fe fe_one() { return BBP_FE_LIT(1, 0, 0, 0, 0, 0, 0, 0); }
BBP_HD fe fe_d() { return BBP_FE_LIT(0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu); }
BBP_HD fe fe_d2() { return BBP_FE_LIT(0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu); }
BBP_HD fe fe_sqrt_m1() { return BBP_FE_LIT(0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u); }
BBP_HD fe fe_sqrt_ad_minus_one() { return BBP_FE_LIT(0x497b2e1bu, 0x7e97f6a0u, 0x1b7854bdu, 0xaf9d8e0cu, 0x31f5d1fdu, 0x0f3cfcc9u, 0x2b8348acu, 0x376931bfu); }
BBP_HD fe fe_invsqrt_a_minus_d() { return BBP_FE_LIT(0x805d40eau, 0x99c8fdaau, 0x5a4172beu, 0x9d2f1617u, 0xfe01d840u, 0x16c27b91u, 0xcfaffca2u, 0x786c8905u); }
BBP_HD fe fe_one_minus_d_sq() { return BBP_FE_LIT(0x945fc176u, 0xe27c09c1u, 0xcd5e350fu, 0x2c81a138u, 0xbe70dfe4u, 0x9994abddu, 0xb2b3e0d7u, 0x029072a8u); }
BBP_HD fe fe_d_minus_one_sq() { return BBP_FE_LIT(0x44ed4d20u, 0x31ad5aaau, 0xb01e1999u, 0xd29e4a2cu, 0x529b4eebu, 0x4cdcd32fu, 0xf66c2241u, 0x5968b37au); }

BBP_HD fe fe_add(const fe& a, const fe& b) {
    fe r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)a.v[i] + b.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    u64 k = c * 38u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        k += r.v[i];
        r.v[i] = (u32)k;
        k >>= 32;
    }
    r.v[0] += (u32)k * 38u;  // second wrap leaves r < 38: no further carry
    return r;
}

BBP_HD fe fe_sub(const fe& a, const fe& b) {
    fe r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)b.v[i];
        r.v[i] = (u32)c;
        c >>= 32;  // arithmetic: 0 or -1
    }
    int64_t k = c * 38;  // 0 or -38
#pragma unroll
    for (int i = 0; i < 8; i++) {
        k += r.v[i];
        r.v[i] = (u32)k;
        k >>= 32;
    }
    r.v[0] += (u32)((int32_t)k * 38);  // second wrap leaves r >= 2^256-38: no further borrow
    return r;
}

BBP_HD fe fe_neg(const fe& a) { return fe_sub(fe_zero(), a); }

// fold a 512-bit product t[0..15] to [0, 2^256)
BBP_HD fe fe_fold512(const u32* t) {
    fe r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)t[i + 8] * 38u + t[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    u64 k = c * 38u;  // c <= 38
#pragma unroll
    for (int i = 0; i < 8; i++) {
        k += r.v[i];
        r.v[i] = (u32)k;
        k >>= 32;
    }
    r.v[0] += (u32)k * 38u;
    return r;
}

BBP_HD fe fe_mul(const fe& a, const fe& b) {
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (u64)a.v[i] * b.v[j] + t[i + j];
            t[i + j] = (u32)c;
            c >>= 32;
        }
        t[i + 8] = (u32)c;
    }
    return fe_fold512(t);
}

BBP_HD fe fe_sq(const fe& a) {
    // off-diagonal products once, doubled, plus the diagonal: 36 multiplies instead of 64
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = i + 1; j < 8; j++) {
            c += (u64)a.v[i] * a.v[j] + t[i + j];
            t[i + j] = (u32)c;
            c >>= 32;
        }
        t[i + 8] = (u32)c;
    }
    // double
    u32 top = 0;
#pragma unroll
    for (int i = 1; i < 16; i++) {
        u32 nt = t[i] >> 31;
        t[i] = (t[i] << 1) | top;
        top = nt;
    }
    // add diagonal
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 sq = (u64)a.v[i] * a.v[i];
        c += (u64)t[2 * i] + (u32)sq;
        t[2 * i] = (u32)c;
        c >>= 32;
        c += (u64)t[2 * i + 1] + (u32)(sq >> 32);
        t[2 * i + 1] = (u32)c;
        c >>= 32;
    }
    return fe_fold512(t);
}

BBP_HD fe fe_mul_small(const fe& a, u32 s) {  // s < 2^26
    fe r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)a.v[i] * s;
        r.v[i] = (u32)c;
        c >>= 32;
    }
    u64 k = c * 38u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        k += r.v[i];
        r.v[i] = (u32)k;
        k >>= 32;
    }
    r.v[0] += (u32)k * 38u;
    return r;
}

// canonical representative in [0, p)
BBP_HD fe fe_canon(const fe& a) {
    fe r = a;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        u32 top = r.v[7] >> 31;
        r.v[7] &= 0x7fffffffu;
        u64 c = (u64)top * 19u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += r.v[i];
            r.v[i] = (u32)c;
            c >>= 32;
        }
    }
    // now r < 2^255; subtract p if r >= p  <=>  r + 19 >= 2^255
    fe t;
    u64 c = 19;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += r.v[i];
        t.v[i] = (u32)c;
        c >>= 32;
    }
    u32 ge = t.v[7] >> 31;
    t.v[7] &= 0x7fffffffu;
    u32 mask = 0u - ge;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (t.v[i] & mask) | (r.v[i] & ~mask);
    return r;
}

BBP_HD void fe_tobytes(uint8_t* out, const fe& a) {
    fe c = fe_canon(a);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        out[4 * i + 0] = (uint8_t)(c.v[i]);
        out[4 * i + 1] = (uint8_t)(c.v[i] >> 8);
        out[4 * i + 2] = (uint8_t)(c.v[i] >> 16);
        out[4 * i + 3] = (uint8_t)(c.v[i] >> 24);
    }
}

// words = 8 little-endian u32 (callers load bytes as words); bit 255 is ignored like dalek's from_bytes
BBP_HD fe fe_fromwords(const u32* w) {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = w[i];
    r.v[7] &= 0x7fffffffu;
    return r;
}

BBP_HD bool fe_iszero(const fe& a) {
    fe c = fe_canon(a);
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= c.v[i];
    return o == 0;
}

BBP_HD bool fe_eq(const fe& a, const fe& b) { return fe_iszero(fe_sub(a, b)); }
BBP_HD bool fe_isneg(const fe& a) { return fe_canon(a).v[0] & 1u; }

BBP_HD fe fe_select(const fe& a, const fe& b, bool pick_b) {
    fe r;
    u32 m = 0u - (u32)pick_b;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (b.v[i] & m) | (a.v[i] & ~m);
    return r;
}

BBP_HD fe fe_abs(const fe& a) { return fe_select(a, fe_neg(a), fe_isneg(a)); }

BBP_HD fe fe_sqn(fe a, int n) {
    for (int i = 0; i < n; i++) a = fe_sq(a);
    return a;
}

// a^(2^250 - 1) and a^11, the shared prefix of inversion and pow22523
BBP_HD void fe_pow_prefix(const fe& z, fe& t250, fe& z11) {
    fe z2 = fe_sq(z);
    fe z9 = fe_mul(fe_sqn(z2, 2), z);
    z11 = fe_mul(z9, z2);
    fe z2_5_0 = fe_mul(fe_sq(z11), z9);
    fe z2_10_0 = fe_mul(fe_sqn(z2_5_0, 5), z2_5_0);
    fe z2_20_0 = fe_mul(fe_sqn(z2_10_0, 10), z2_10_0);
    fe z2_40_0 = fe_mul(fe_sqn(z2_20_0, 20), z2_20_0);
    fe z2_50_0 = fe_mul(fe_sqn(z2_40_0, 10), z2_10_0);
    fe z2_100_0 = fe_mul(fe_sqn(z2_50_0, 50), z2_50_0);
    fe z2_200_0 = fe_mul(fe_sqn(z2_100_0, 100), z2_100_0);
    t250 = fe_mul(fe_sqn(z2_200_0, 50), z2_50_0);
}

BBP_HD_NOINLINE fe fe_invert(const fe& z) {  // z^(p-2) = z^(2^255 - 21)
    fe t250, z11;
    fe_pow_prefix(z, t250, z11);
    return fe_mul(fe_sqn(t250, 5), z11);
}

BBP_HD_NOINLINE fe fe_pow22523(const fe& z) {  // z^((p-5)/8) = z^(2^252 - 3)
    fe t250, z11;
    fe_pow_prefix(z, t250, z11);
    return fe_mul(fe_sqn(t250, 2), z);
}

// RFC 9496 4.2 SQRT_RATIO_M1(u, v): returns was_square, r = |sqrt(u/v)| or |sqrt(i*u/v)|
BBP_HD bool fe_sqrt_ratio_m1(fe& r, const fe& u, const fe& v) {
    fe v3 = fe_mul(fe_sq(v), v);
    fe v7 = fe_mul(fe_sq(v3), v);
    r = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
    fe check = fe_mul(v, fe_sq(r));
    fe neg_u = fe_neg(u);
    bool correct = fe_eq(check, u);
    bool flipped = fe_eq(check, neg_u);
    bool flipped_i = fe_eq(check, fe_mul(neg_u, fe_sqrt_m1()));
    fe r_prime = fe_mul(r, fe_sqrt_m1());
    r = fe_select(r, r_prime, flipped || flipped_i);
    r = fe_abs(r);
    return correct || flipped;
}

}  // namespace bbp
