// Field arithmetic mod p = 2^255 - 19 for gfx950, written for 32-bit VALU lanes.
//
// Representation: ten signed limbs in radix 2^25.5 (26, 25, 26, 25, ... bits), unsaturated.
// Why this form on CDNA4: a point is 40 VGPRs, so the hot kernels run at 2 waves/SIMD and cannot hide dependent-issue
// latency.  With saturated 32-bit limbs every add/sub/multiply is one long carry chain; here an addition is ten
// independent v_add_u32, the schoolbook product is ten INDEPENDENT columns of ten v_mad_i64_i32 (pure multiply-
// accumulate, no carry handling inside), and one short interleaved carry pass normalises at the end of a multiply.
// Measured on MI355X (tools/ubench.py): mixed point addition 3.09e10/s in this form vs 1.79e10/s with 8x32 saturated limbs.
//
// Bounds (the discipline of the public-domain ref10 layout): multiply / square accept limbs up to |f_even| <= 1.65*2^26,
// |f_odd| <= 1.65*2^25 and return |h_even| <= 1.01*2^25, |h_odd| <= 1.01*2^24 ("carried"), so a sum or difference of up
// to three carried values may feed the next multiply without normalising.  Every formula in point.h stays inside that.
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
typedef int32_t i32;
typedef int64_t i64;

struct fe {
    i32 v[10];
};

// 8 little-endian 32-bit words (bit 255 ignored, like dalek's FieldElement::from_bytes) -> limbs.
// With literal arguments this folds at compile time, so constants cost nothing.
BBP_HD fe fe_fromwords(const u32* w) {
    fe r;
    r.v[0] = (i32)(w[0] & 0x3ffffffu);                                   // bits   0..25
    r.v[1] = (i32)(((w[0] >> 26) | (w[1] << 6)) & 0x1ffffffu);           // bits  26..50
    r.v[2] = (i32)(((w[1] >> 19) | (w[2] << 13)) & 0x3ffffffu);          // bits  51..76
    r.v[3] = (i32)(((w[2] >> 13) | (w[3] << 19)) & 0x1ffffffu);          // bits  77..101
    r.v[4] = (i32)((w[3] >> 6) & 0x3ffffffu);                            // bits 102..127
    r.v[5] = (i32)(w[4] & 0x1ffffffu);                                   // bits 128..152
    r.v[6] = (i32)(((w[4] >> 25) | (w[5] << 7)) & 0x3ffffffu);           // bits 153..178
    r.v[7] = (i32)(((w[5] >> 19) | (w[6] << 13)) & 0x1ffffffu);          // bits 179..203
    r.v[8] = (i32)(((w[6] >> 12) | (w[7] << 20)) & 0x3ffffffu);          // bits 204..229
    r.v[9] = (i32)((w[7] >> 6) & 0x1ffffffu);                            // bits 230..254
    return r;
}

BBP_HD fe fe_from8(u32 a0, u32 a1, u32 a2, u32 a3, u32 a4, u32 a5, u32 a6, u32 a7) {
    const u32 w[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
    return fe_fromwords(w);
}
#define BBP_FE_LIT(a0, a1, a2, a3, a4, a5, a6, a7) fe_from8(a0, a1, a2, a3, a4, a5, a6, a7)

BBP_HD fe fe_zero() { return fe{{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
BBP_HD fe fe_one() { return fe{{1, 0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
BBP_HD fe fe_d() { return BBP_FE_LIT(0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu); }
BBP_HD fe fe_d2() { return BBP_FE_LIT(0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu); }
BBP_HD fe fe_sqrt_m1() { return BBP_FE_LIT(0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u); }
BBP_HD fe fe_sqrt_ad_minus_one() { return BBP_FE_LIT(0x497b2e1bu, 0x7e97f6a0u, 0x1b7854bdu, 0xaf9d8e0cu, 0x31f5d1fdu, 0x0f3cfcc9u, 0x2b8348acu, 0x376931bfu); }
BBP_HD fe fe_invsqrt_a_minus_d() { return BBP_FE_LIT(0x805d40eau, 0x99c8fdaau, 0x5a4172beu, 0x9d2f1617u, 0xfe01d840u, 0x16c27b91u, 0xcfaffca2u, 0x786c8905u); }
BBP_HD fe fe_one_minus_d_sq() { return BBP_FE_LIT(0x945fc176u, 0xe27c09c1u, 0xcd5e350fu, 0x2c81a138u, 0xbe70dfe4u, 0x9994abddu, 0xb2b3e0d7u, 0x029072a8u); }
BBP_HD fe fe_d_minus_one_sq() { return BBP_FE_LIT(0x44ed4d20u, 0x31ad5aaau, 0xb01e1999u, 0xd29e4a2cu, 0x529b4eebu, 0x4cdcd32fu, 0xf66c2241u, 0x5968b37au); }

BBP_HD fe fe_add(const fe& a, const fe& b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}

BBP_HD fe fe_sub(const fe& a, const fe& b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] - b.v[i];
    return r;
}

BBP_HD fe fe_neg(const fe& a) {
    fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = -a.v[i];
    return r;
}

// interleaved carry pass over ten 64-bit columns -> carried limbs
BBP_HD fe fe_carry64(i64 (&h)[10]) {
    i64 c;
#define BBP_CARRY(i, bits, nxt, mul)                 \
    c = (h[i] + ((i64)1 << ((bits) - 1))) >> (bits); \
    h[nxt] += c * (mul);                             \
    h[i] -= c << (bits);
    BBP_CARRY(0, 26, 1, 1)
    BBP_CARRY(4, 26, 5, 1)
    BBP_CARRY(1, 25, 2, 1)
    BBP_CARRY(5, 25, 6, 1)
    BBP_CARRY(2, 26, 3, 1)
    BBP_CARRY(6, 26, 7, 1)
    BBP_CARRY(3, 25, 4, 1)
    BBP_CARRY(7, 25, 8, 1)
    BBP_CARRY(4, 26, 5, 1)
    BBP_CARRY(8, 26, 9, 1)
    BBP_CARRY(9, 25, 0, 19)
    BBP_CARRY(0, 26, 1, 1)
#undef BBP_CARRY
    fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = (i32)h[i];
    return r;
}

// Carry pass for columns that were PRE-BIASED: the multiply kernels start column k at 2^(bits_k - 1) instead of 0 (a constant
// addend of the first multiply-add, so free), which turns the round-to-nearest carry c = (h + 2^(bits-1)) >> bits of the ten
// first-time carries into a plain shift, and the signed remainder into (H & mask) - 2^(bits-1).  Same carry order and the same
// results as fe_carry64; the two second-time carries (columns 4 and 0) work on small unbiased values.
BBP_HD fe fe_carry64_prebiased(i64 (&h)[10]) {
    fe r;
    i64 c;
#define BBP_PB(i, bits, nxt)                                                          \
    c = h[i] >> (bits);                                                               \
    h[nxt] += c;                                                                      \
    r.v[i] = (i32)((u32)h[i] & ((1u << (bits)) - 1u)) - (i32)(1u << ((bits) - 1));
    BBP_PB(0, 26, 1)
    BBP_PB(4, 26, 5)
    BBP_PB(1, 25, 2)
    BBP_PB(5, 25, 6)
    BBP_PB(2, 26, 3)
    BBP_PB(6, 26, 7)
    c = h[3] >> 25;  // column 3 -> column 4, which has been carried once already
    r.v[3] = (i32)((u32)h[3] & ((1u << 25) - 1u)) - (i32)(1u << 24);
    i64 x4 = (i64)r.v[4] + c;
    BBP_PB(7, 25, 8)
    c = (x4 + ((i64)1 << 25)) >> 26;
    r.v[4] = (i32)(x4 - (c << 26));
    r.v[5] += (i32)c;  // |c| < 2^12
    BBP_PB(8, 26, 9)
    c = h[9] >> 25;  // column 9 wraps to column 0 times 19
    r.v[9] = (i32)((u32)h[9] & ((1u << 25) - 1u)) - (i32)(1u << 24);
    i64 x0 = (i64)r.v[0] + 19 * c;
    c = (x0 + ((i64)1 << 25)) >> 26;
    r.v[0] = (i32)(x0 - (c << 26));
    r.v[1] += (i32)c;  // |c| < 2^17
#undef BBP_PB
    return r;
}

#define BBP_FE_BIAS(k) ((i64)1 << (((k) & 1) ? 24 : 25))  // 2^(bits_k - 1): even columns hold 26 bits, odd ones 25
// The bias is meant to ride in as the 64-bit addend of the column's FIRST multiply-add (free).  Written as a C sum the compiler
// reassociates it: the constant is added at the END of the chain with a v_lshl_add_u64 of its own -- ten extra VOP3 instructions per
// multiplication (3.45 cycles each per SIMD at two waves per SIMD against 4.38 for the multiply-add itself, tools/exp_issue.hip: 5 %
// of a mixed addition; an operand the optimiser cannot see through is reassociated just the same, and so is a sum whose first
// term alone is opaque).  So on the device a column is ONE inline-assembly block: v_mad_i64_i32 with the bias from an SGPR pair,
// then the remaining terms accumulating in place.  One block per column, not one statement per multiply-add: the hazard recogniser
// pads every pair of adjacent asm statements with an s_nop.  The chain inside a block is dependent, which costs nothing here: one
// wave alone cannot issue these faster than one per 8.75 cycles whatever their independence, two waves share the SIMD at 4.38.
// (-DBBP_FE_MAD_C keeps the plain C form for A/B runs.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BBP_FE_MAD_C)
#define BBP_FE_MAD_ASM 1
#ifndef BBP_FE_NO_CHAIN
#define BBP_FE_CHAIN 1  // carries threaded through the column sums (below): measured 21.30 k against 20.88 k proofs/s, exclusive accumulate time -4.5 %
#endif
#define BBP_MAD0 "v_mad_i64_i32 %0, vcc, %2, %3, %1\n\t"
#define BBP_MADN(a, b) "v_mad_i64_i32 %0, vcc, %" #a ", %" #b ", %0\n\t"
__device__ __forceinline__ i64 fe_col10(i64 bias, i32 a0, i32 b0, i32 a1, i32 b1, i32 a2, i32 b2, i32 a3, i32 b3, i32 a4, i32 b4, i32 a5, i32 b5, i32 a6,
                                        i32 b6, i32 a7, i32 b7, i32 a8, i32 b8, i32 a9, i32 b9) {
    i64 d;
    asm(BBP_MAD0 BBP_MADN(4, 5) BBP_MADN(6, 7) BBP_MADN(8, 9) BBP_MADN(10, 11) BBP_MADN(12, 13) BBP_MADN(14, 15) BBP_MADN(16, 17) BBP_MADN(18, 19) BBP_MADN(20, 21)
        : "=&v"(d)
        : "s"(bias), "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6), "v"(a7),
          "v"(b7), "v"(a8), "v"(b8), "v"(a9), "v"(b9)
        : "vcc");
    return d;
}
__device__ __forceinline__ i64 fe_col6(i64 bias, i32 a0, i32 b0, i32 a1, i32 b1, i32 a2, i32 b2, i32 a3, i32 b3, i32 a4, i32 b4, i32 a5, i32 b5) {
    i64 d;
    asm(BBP_MAD0 BBP_MADN(4, 5) BBP_MADN(6, 7) BBP_MADN(8, 9) BBP_MADN(10, 11) BBP_MADN(12, 13)
        : "=&v"(d)
        : "s"(bias), "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5)
        : "vcc");
    return d;
}
__device__ __forceinline__ i64 fe_col5(i64 bias, i32 a0, i32 b0, i32 a1, i32 b1, i32 a2, i32 b2, i32 a3, i32 b3, i32 a4, i32 b4) {
    i64 d;
    asm(BBP_MAD0 BBP_MADN(4, 5) BBP_MADN(6, 7) BBP_MADN(8, 9) BBP_MADN(10, 11)
        : "=&v"(d)
        : "s"(bias), "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4)
        : "vcc");
    return d;
}
#define BBP_MAD0V "v_mad_i64_i32 %0, vcc, %2, %3, %1\n\t"
// the same with the first addend in a VGPR pair (the carry of the previous column: BBP_FE_CHAIN)
__device__ __forceinline__ i64 fe_col10v(i64 addend, i32 a0, i32 b0, i32 a1, i32 b1, i32 a2, i32 b2, i32 a3, i32 b3, i32 a4, i32 b4, i32 a5, i32 b5, i32 a6,
                                         i32 b6, i32 a7, i32 b7, i32 a8, i32 b8, i32 a9, i32 b9) {
    i64 d;
    asm(BBP_MAD0V BBP_MADN(4, 5) BBP_MADN(6, 7) BBP_MADN(8, 9) BBP_MADN(10, 11) BBP_MADN(12, 13) BBP_MADN(14, 15) BBP_MADN(16, 17) BBP_MADN(18, 19) BBP_MADN(20, 21)
        : "=&v"(d)
        : "v"(addend), "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6), "v"(a7),
          "v"(b7), "v"(a8), "v"(b8), "v"(a9), "v"(b9)
        : "vcc");
    return d;
}
__device__ __forceinline__ i64 fe_col5v(i64 addend, i32 a0, i32 b0, i32 a1, i32 b1, i32 a2, i32 b2, i32 a3, i32 b3, i32 a4, i32 b4) {
    i64 d;
    asm(BBP_MAD0V BBP_MADN(4, 5) BBP_MADN(6, 7) BBP_MADN(8, 9) BBP_MADN(10, 11)
        : "=&v"(d)
        : "v"(addend), "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4)
        : "vcc");
    return d;
}
#undef BBP_MAD0V
#undef BBP_MAD0
#undef BBP_MADN
#endif

// BBP_FE_CHAIN (default; -DBBP_FE_NO_CHAIN for the separate carry pass): the carry pass threaded THROUGH the column sums.  Columns are summed in order; an even column's seed constant is its own
// rounding bias plus the next (odd) column's bias shifted up by its 26 bits, so that its carry h >> 26 already contains that bias and can
// be the FIRST ADDEND of the odd column's chain (free); the odd column's carry is added to the next even column with one 64-bit add.  Four
// explicit 64-bit adds per multiplication instead of nine, and no second visit of columns 4 and 0 beyond the wrap (19 c9 into limb 0).
// The carry order differs from fe_carry64_prebiased, so limbs may differ -- the field element does not (every output is carried).
constexpr i64 FE_SEED_EVEN = ((i64)1 << 25) + ((i64)1 << 50);
BBP_HD void fe_chain_step(fe& r, i64& c, int k, i64 hk) {  // hk: column k with its seed / carry-in already inside
    if (k & 1) {
        c = hk >> 25;
        r.v[k] = (i32)((u32)hk & ((1u << 25) - 1u)) - (i32)(1u << 24);
    } else {
        c = hk >> 26;  // (for k < 9 this carries the next column's rounding bias with it)
        r.v[k] = (i32)((u32)hk & ((1u << 26) - 1u)) - (i32)(1u << 25);
    }
}
BBP_HD void fe_chain_wrap(fe& r, i64 c9) {  // column 9 wraps to column 0 times 19, then one small carry 0 -> 1
    const i64 x0 = (i64)r.v[0] + 19 * c9;
    const i64 c = (x0 + ((i64)1 << 25)) >> 26;
    r.v[0] = (i32)(x0 - (c << 26));
    r.v[1] += (i32)c;
}

// The same threaded carry pass over column sums computed in plain C: what the device path does with its assembly chains, for the
// host (tests/host_check.cpp op 11 / 12 drives it with the product's own fe_chain_step / fe_chain_wrap: seed constants, carry order
// and limb bounds are checked on the CPU tier; the GPU tier checks the assembly against the oracle).
BBP_HD fe fe_mul_chain_portable(const fe& f, const fe& g) {
    i32 g19[10], f2[10];
    for (int i = 0; i < 10; i++) {
        g19[i] = 19 * g.v[i];
        f2[i] = 2 * f.v[i];
    }
    fe r;
    i64 c = 0;
    for (int k = 0; k < 10; k++) {
        i64 hk = (k & 1) ? c : FE_SEED_EVEN + (k ? c : 0);
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            const i32 a = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            const i32 b = (i + j >= 10) ? g19[j] : g.v[j];
            hk += (i64)a * b;
        }
        fe_chain_step(r, c, k, hk);
    }
    fe_chain_wrap(r, c);
    return r;
}

// term f_i g_j lands in column (i+j) mod 10, times 19 when i+j >= 10 (2^255 = 19), times 2 when i and j are both odd
BBP_HD fe fe_mul(const fe& f, const fe& g) {
    i32 g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        g19[i] = 19 * g.v[i];
        f2[i] = 2 * f.v[i];
    }
#if defined(BBP_FE_MAD_ASM) && defined(BBP_FE_CHAIN)
    {
        fe r;
        i64 c = 0;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            i32 a[10], b[10];
#pragma unroll
            for (int i = 0; i < 10; i++) {
                const int j = (k - i + 10) % 10;
                a[i] = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
                b[i] = (i + j >= 10) ? g19[j] : g.v[j];
            }
            i64 hk;
            if (k & 1)
                hk = fe_col10v(c, a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], a[6], b[6], a[7], b[7], a[8], b[8], a[9], b[9]);
            else {
                hk = fe_col10(FE_SEED_EVEN, a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], a[6], b[6], a[7], b[7], a[8], b[8], a[9], b[9]);
                if (k) hk += c;
            }
            fe_chain_step(r, c, k, hk);
        }
        fe_chain_wrap(r, c);
        return r;
    }
#endif
    i64 h[10];
#ifdef BBP_FE_MAD_ASM
#pragma unroll
    for (int k = 0; k < 10; k++) {
        i32 a[10], b[10];  // column k: terms (i, j) with i + j = k or k + 10
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            a[i] = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            b[i] = (i + j >= 10) ? g19[j] : g.v[j];
        }
        h[k] = fe_col10(BBP_FE_BIAS(k), a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], a[6], b[6], a[7], b[7], a[8], b[8], a[9], b[9]);
    }
#else
#pragma unroll
    for (int k = 0; k < 10; k++) h[k] = BBP_FE_BIAS(k);
#pragma unroll
    for (int i = 0; i < 10; i++) {
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const int k = i + j;
            const i32 fi = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            const i32 gj = (k >= 10) ? g19[j] : g.v[j];
            h[k % 10] += (i64)fi * gj;
        }
    }
#endif
    return fe_carry64_prebiased(h);
}

// dedicated squaring: 55 multiplies (off-diagonal terms once, doubled)
// coefficient of f_i f_j (i <= j, k = i + j): (i == j ? 1 : 2) * (both odd ? 2 : 1) * (k >= 10 ? 19 : 1); the factors are split so that every
// 32-bit operand stays below 2^31: 2 goes on f_i, 19 / 38 on f_j (38 only for odd j)
#define BBP_SQ_TERM(i, j, a, b)                                                                                       \
    do {                                                                                                              \
        const bool both_odd_ = ((i) & 1) && ((j) & 1);                                                                \
        if ((i) == (j)) {                                                                                             \
            a = both_odd_ ? f2[i] : f.v[i];                                                                           \
            b = ((i) + (j) >= 10) ? f19[j] : f.v[j];                                                                  \
        } else {                                                                                                      \
            a = f2[i];                                                                                                \
            b = both_odd_ ? (((i) + (j) >= 10) ? f38[j] : f2[j]) : (((i) + (j) >= 10) ? f19[j] : f.v[j]);             \
        }                                                                                                             \
    } while (0)
BBP_HD void fe_sq_columns(const fe& f, i64 (&h)[10]) {
    i32 f2[10], f19[10], f38[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        f2[i] = 2 * f.v[i];
        f19[i] = 19 * f.v[i];
        f38[i] = 38 * f.v[i];
    }
#ifdef BBP_FE_MAD_ASM
    // by column (fe_col5 / fe_col6 above): even columns have six terms, odd ones five; the bias rides in the first multiply-add
#pragma unroll
    for (int c = 0; c < 10; c++) {
        i32 a[6] = {0, 0, 0, 0, 0, 0}, b[6] = {0, 0, 0, 0, 0, 0};
        int n = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
#pragma unroll
            for (int j = i; j < 10; j++) {
                if ((i + j) % 10 != c) continue;
                i32 aa, bb;
                BBP_SQ_TERM(i, j, aa, bb);
                a[n] = aa;
                b[n] = bb;
                n++;
            }
        }
        h[c] = (c & 1) ? fe_col5(BBP_FE_BIAS(c), a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4])
                       : fe_col6(BBP_FE_BIAS(c), a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5]);
    }
#else
#pragma unroll
    for (int i = 0; i < 10; i++) {
#pragma unroll
        for (int j = i; j < 10; j++) {
            i32 a, b;
            BBP_SQ_TERM(i, j, a, b);
            h[(i + j) % 10] += (i64)a * b;
        }
    }
#endif
}

#if defined(BBP_FE_MAD_ASM) && defined(BBP_FE_CHAIN)
__device__ __forceinline__ fe fe_sq_chain(const fe& f) {
    i32 f2[10], f19[10], f38[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        f2[i] = 2 * f.v[i];
        f19[i] = 19 * f.v[i];
        f38[i] = 38 * f.v[i];
    }
    fe r;
    i64 c = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        i32 a[6] = {0, 0, 0, 0, 0, 0}, b[6] = {0, 0, 0, 0, 0, 0};
        int n = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
#pragma unroll
            for (int j = i; j < 10; j++) {
                if ((i + j) % 10 != k) continue;
                i32 aa, bb;
                BBP_SQ_TERM(i, j, aa, bb);
                a[n] = aa;
                b[n] = bb;
                n++;
            }
        }
        i64 hk;
        if (k & 1)
            hk = fe_col5v(c, a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4]);
        else {
            hk = fe_col6(FE_SEED_EVEN, a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5]);
            if (k) hk += c;
        }
        fe_chain_step(r, c, k, hk);
    }
    fe_chain_wrap(r, c);
    return r;
}
#endif

BBP_HD fe fe_sq(const fe& f) {
#if defined(BBP_FE_MAD_ASM) && defined(BBP_FE_CHAIN)
    return fe_sq_chain(f);
#endif
    i64 h[10];
#ifndef BBP_FE_MAD_ASM
#pragma unroll
    for (int k = 0; k < 10; k++) h[k] = BBP_FE_BIAS(k);
#endif
    fe_sq_columns(f, h);
    return fe_carry64_prebiased(h);
}

#undef BBP_SQ_TERM

// 2 * f^2, carried (point doubling needs it inside the multiply bounds)
BBP_HD fe fe_sq2(const fe& f) {
    fe g = fe_sq(f);
    i64 h[10];
#pragma unroll
    for (int i = 0; i < 10; i++) h[i] = 2 * (i64)g.v[i];
    return fe_carry64(h);
}

BBP_HD fe fe_mul_small(const fe& a, u32 s) {  // s < 2^26
    i64 h[10];
#pragma unroll
    for (int i = 0; i < 10; i++) h[i] = (i64)a.v[i] * (i64)s;
    return fe_carry64(h);
}

// canonical 8 little-endian words (value in [0, p)); accepts limbs bounded by 1.1 * 2^26 / 2^25
BBP_HD void fe_towords(u32* w, const fe& f) {
    i32 h[10];
#pragma unroll
    for (int i = 0; i < 10; i++) h[i] = f.v[i];
    i32 q = (19 * h[9] + ((i32)1 << 24)) >> 25;
    q = (h[0] + q) >> 26;
    q = (h[1] + q) >> 25;
    q = (h[2] + q) >> 26;
    q = (h[3] + q) >> 25;
    q = (h[4] + q) >> 26;
    q = (h[5] + q) >> 25;
    q = (h[6] + q) >> 26;
    q = (h[7] + q) >> 25;
    q = (h[8] + q) >> 26;
    q = (h[9] + q) >> 25;
    h[0] += 19 * q;
    i32 c;
    c = h[0] >> 26; h[1] += c; h[0] -= c << 26;
    c = h[1] >> 25; h[2] += c; h[1] -= c << 25;
    c = h[2] >> 26; h[3] += c; h[2] -= c << 26;
    c = h[3] >> 25; h[4] += c; h[3] -= c << 25;
    c = h[4] >> 26; h[5] += c; h[4] -= c << 26;
    c = h[5] >> 25; h[6] += c; h[5] -= c << 25;
    c = h[6] >> 26; h[7] += c; h[6] -= c << 26;
    c = h[7] >> 25; h[8] += c; h[7] -= c << 25;
    c = h[8] >> 26; h[9] += c; h[8] -= c << 26;
    c = h[9] >> 25; h[9] -= c << 25;
    const u32 u0 = (u32)h[0], u1 = (u32)h[1], u2 = (u32)h[2], u3 = (u32)h[3], u4 = (u32)h[4];
    const u32 u5 = (u32)h[5], u6 = (u32)h[6], u7 = (u32)h[7], u8_ = (u32)h[8], u9 = (u32)h[9];
    w[0] = u0 | (u1 << 26);
    w[1] = (u1 >> 6) | (u2 << 19);
    w[2] = (u2 >> 13) | (u3 << 13);
    w[3] = (u3 >> 19) | (u4 << 6);
    w[4] = u5 | (u6 << 25);
    w[5] = (u6 >> 7) | (u7 << 19);
    w[6] = (u7 >> 13) | (u8_ << 12);
    w[7] = (u8_ >> 20) | (u9 << 6);
}

BBP_HD void fe_tobytes(uint8_t* out, const fe& a) {
    u32 w[8];
    fe_towords(w, a);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        out[4 * i + 0] = (uint8_t)(w[i]);
        out[4 * i + 1] = (uint8_t)(w[i] >> 8);
        out[4 * i + 2] = (uint8_t)(w[i] >> 16);
        out[4 * i + 3] = (uint8_t)(w[i] >> 24);
    }
}

BBP_HD bool fe_iszero(const fe& a) {
    u32 w[8];
    fe_towords(w, a);
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= w[i];
    return o == 0;
}

BBP_HD bool fe_eq(const fe& a, const fe& b) { return fe_iszero(fe_sub(a, b)); }

BBP_HD bool fe_isneg(const fe& a) {
    u32 w[8];
    fe_towords(w, a);
    return w[0] & 1u;
}

BBP_HD fe fe_select(const fe& a, const fe& b, bool pick_b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = pick_b ? b.v[i] : a.v[i];
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
