// Keccak-f[1600] on ONE wavefront, third form (round 4): one 32-bit HALF of a state word per lane, bit-interleaved.
//
// Why: the prover's TranscriptRng draw chain (one permutation per 64-byte draw, 2 + 2 n1 of them in sequence: merlin 1.3.0
// `TranscriptRng::fill_bytes`, driven by bulletproofs' prover -- SURVEY.md App. A.5 step 4; reference call site
// src/blindbid/proof.rs:88) is what a single proof waits for: 9.3 of 18.3 ms.  A lone wavefront issues one instruction per ~5
// cycles whatever its active lanes, so the chain's time is its instruction count.  The second form (prover.hip coop8_*: word (x, y)
// in lane 8 y + x, two registers per word) spends 38 vector + 6 LDS-crossbar instructions and 8 pads per round, every operation
// once per half.  Here the 50 halves sit in 50 lanes and every operation is issued once:
//
//   * words are kept BIT-INTERLEAVED (even bits in lanes 0..31, odd bits in lanes 32..63 of the same position), so a 64-bit
//     rotation is a 32-bit rotation of each half (by an odd amount: the halves also change places -- the rho-pi gather simply
//     reads the other half's lane).  One v_alignbit per round instead of four plus two selects;
//   * a half (x, y) lives at position 16 (y / 3) + 5 (y % 3) + x of its 32 lanes: three planes per DPP row (lane 15 idle), two in
//     the second row.  Column parities by four row shifts (+-5, +-10) come out PERIODIC along the row, so the neighbours x - 1 and
//     x + 1 are plain row shifts by one with a second, bank-masked shift by -+4 covering the row's ends -- no selects;
//   * v_permlane16_swap joins the two rows of a half, v_permlane32_swap hands theta's rotated neighbour to the other half;
//   * rho-pi-chi: one v_alignbit, three ds_bpermute, one v_bitop3; iota is one XOR with a per-round lane vector.
//
// 20 vector + 3 crossbar instructions per round.  Idle lanes hold zero and stay zero (theta's update is masked).
// Bytes are those of the other forms: the draws are written as (even, odd) pairs and k_reduce_draws joins them.
#pragma once
#include "keccak.h"

namespace bbp {

// bits 0, 2, 4, ... of x
BBP_HD u32 kw_even_bits(u64 x) {
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
    x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
    x = (x | (x >> 16)) & 0x00000000ffffffffull;
    return (u32)x;
}
BBP_HD u64 kw_spread(u32 v) {
    u64 x = v;
    x = (x | (x << 16)) & 0x0000ffff0000ffffull;
    x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
BBP_HD u32 kw_half(u64 x, u32 h) { return kw_even_bits(h ? x >> 1 : x); }
BBP_HD u64 kw_join(u32 even, u32 odd) { return kw_spread(even) | (kw_spread(odd) << 1); }

struct kw_lane {
    int s0, s1, s2;    // ds_bpermute byte addresses of the (pre-rotated) sources of B[x][y], B[x+1][y], B[x+2][y], this lane's half
    u32 sh_rho;        // v_alignbit shift of this half's share of the rho rotation (applied at the source)
    u32 sh_theta;      // 31 in the odd-bit lanes (theta's rotl by one turns odd bits into even bits one place up), 0 in the even-bit lanes
    u32 live;          // all ones in the 50 state lanes (kept opaque: the masked update is one v_bitop3)
    u32 word, half;    // which state word (x + 5 y) and which half this lane holds (state lanes only)
    bool lower;        // lane < 32
};

BBP_HD u32 kw_lane_of(u32 x, u32 y, u32 h) { return 32 * h + 16 * (y / 3) + 5 * (y % 3) + x; }

BBP_HD kw_lane kw_setup(u32 L) {
    const u32 h = L >> 5, pos = L & 15u, rr = (L >> 4) & 1u, x = pos % 5, y = 3 * rr + pos / 5;
    const bool live = pos < 15 && y < 5;
    const u32 RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    kw_lane c;
    // pi: B[X][Y] is the rotated word of source (xs, ys) = ((X + 3 Y) mod 5, X); a rotation by an odd amount takes the other half
    auto src = [&](u32 X) -> int {
        if (!live) return 4 * (int)L;  // idle lanes read themselves (zero)
        const u32 xs = (X + 3 * y) % 5, ys = X, r = RHO[xs + 5 * ys];
        return 4 * (int)kw_lane_of(xs, ys, h ^ (r & 1u));
    };
    c.s0 = src(x % 5);
    c.s1 = src((x + 1) % 5);
    c.s2 = src((x + 2) % 5);
    // as a source: rotl64 by r = 2k: both halves by k; by 2k + 1: odd -> even by k + 1, even -> odd by k
    const u32 r = live ? RHO[(x + 5 * y) % 25] : 0u;
    const u32 amt = (r & 1u) ? (h ? (r + 1) / 2 : (r - 1) / 2) : r / 2;
    c.sh_rho = (32u - amt) & 31u;
    c.sh_theta = h ? 31u : 0u;
    c.live = live ? ~0u : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(c.live));
#endif
    c.word = (x + 5 * y) % 25;
    c.half = h;
    c.lower = L < 32;
    return c;
}

// iota: lane vectors of the 24 round constants -- v: even bits in the lane of word 0's even half, odd bits in its odd half's;
// cp: the same constant as it shows in the column parities (every lane of column 0, by half).  Theta's shifts read the state BEFORE
// iota (two wait states after chi otherwise), and the constant's share enters with the row join.
struct kw_iota {
    u32 v[25], cp[25];  // [r]: the constant PENDING at the start of round r (none at r = 0); v[24]: the last round's, applied at the end
};
BBP_HD kw_iota kw_iota_setup(u32 L) {
    const u64 RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                        0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                        0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                        0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                        0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    kw_iota k;
    const u32 pos = L & 15u;
    const bool col0 = pos < 15 && pos % 5 == 0;
    k.v[0] = k.cp[0] = 0u;
#pragma unroll
    for (int r = 0; r < 24; r++) {  // the constants fold at compile time
        const u32 e = kw_half(RC[r], 0), o = kw_half(RC[r], 1);
        k.v[r + 1] = L == 0 ? e : L == 32 ? o : 0u;
        k.cp[r + 1] = !col0 ? 0u : L < 32 ? e : o;
    }
    return k;
}

#if defined(__HIPCC__)
#define BBP_KW_ROW_SHL(n) (0x100 + (n))
#define BBP_KW_ROW_SHR(n) (0x110 + (n))

// The rounds, scheduled by hand (the compiler re-associates the XOR tree into a chain of dependent DPP operations with a
// two-wait-state pad after each -- a pad costs a lone wavefront as much as an instruction, tools/exp_wave_latency.hip -- and pads
// every boundary between asm blocks, hence eight rounds per block).
// x: the state before the previous round's iota (in), the state before this round's iota (out); IV / CP: the previous round's
// constant as a lane vector and as it shows in the column parities.  21 vector + 3 crossbar instructions, 6 pad slots per round.
//   t0 = a, t1..t4 scratch.  bitop3 tables: 0x96 = xor3, 0x78 = s0 ^ (s1 & s2), 0xd2 = s0 ^ (~s1 & s2).
#define BBP_KW_DPP " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define BBP_KW_ROUND(IV, CP)                                                                                                   \
    "v_xor_b32 %[t0], %[x], %[" #IV "]\n\t" /* a = x ^ iota (fills one of chi's two wait states) */                           \
    "s_nop 0\n\t" /* theta: column parities of the row's planes, periodic in the lane position */                              \
    "v_xor_b32_dpp %[t1], %[x], %[x] row_shl:5" BBP_KW_DPP                                                                     \
    "v_mov_b32_dpp %[t2], %[x] row_shl:10" BBP_KW_DPP                                                                          \
    "v_mov_b32_dpp %[t3], %[x] row_shr:5" BBP_KW_DPP                                                                           \
    "v_xor_b32_dpp %[t1], %[x], %[t1] row_shr:10" BBP_KW_DPP                                                                   \
    "v_bitop3_b32 %[t4], %[t1], %[t2], %[t3] bitop3:0x96\n\t"                                                                 \
    "v_bitop3_b32 %[t1], %[t1], %[t2], %[t3] bitop3:0x96\n\t"                                                                 \
    "s_nop 1\n\t"                                                                                                             \
    "v_permlane16_swap_b32 %[t1], %[t4]\n\t" /* (r0, r0, r2, r2), (r1, r1, r3, r3) */                                         \
    "v_bitop3_b32 %[t2], %[t1], %[t4], %[" #CP "] bitop3:0x96\n\t" /* C of this lane's half (+ the pending iota's share) */   \
    "v_alignbit_b32 %[t3], %[t2], %[t2], %[sht]\n\t" /* what the other half needs of C[x + 1] */                              \
    "s_nop 0\n\t"                                                                                                             \
    "v_mov_b32_dpp %[t1], %[t2] row_shr:1" BBP_KW_DPP /* C[x - 1]: lane - 1 ... */                                             \
    "v_mov_b32_dpp %[t4], %[t3] row_shl:1" BBP_KW_DPP /* C[x + 1]: lane + 1 ... */                                             \
    "s_nop 0\n\t"                                                                                                             \
    "v_mov_b32_dpp %[t1], %[t2] row_shl:4 row_mask:0xf bank_mask:0x1\n\t" /* ... or lane + 4 in the row's first bank */       \
    "v_mov_b32_dpp %[t4], %[t3] row_shr:4 row_mask:0xf bank_mask:0xc\n\t" /* ... or lane - 4 in its last two banks */         \
    "v_mov_b32 %[t2], %[t4]\n\t"                                                                                              \
    "v_xor_b32 %[t3], %[t1], %[t4]\n\t"                                                                                       \
    "s_nop 0\n\t"                                                                                                             \
    "v_permlane32_swap_b32 %[t4], %[t2]\n\t" /* (lower, lower), (upper, upper) */                                             \
    "v_bitop3_b32 %[t3], %[t3], %[t4], %[t2] bitop3:0x96\n\t" /* D = C[x - 1] ^ own ^ lower ^ upper: the OTHER half's C[x + 1] */ \
    "v_bitop3_b32 %[t0], %[t0], %[t3], %[live] bitop3:0x78\n\t" /* a ^= D in the state lanes */                               \
    "v_alignbit_b32 %[t0], %[t0], %[t0], %[shr]\n\t" /* rho at the source; pi and chi's two neighbours in one gather phase */  \
    "ds_bpermute_b32 %[t1], %[s0], %[t0]\n\t"                                                                                 \
    "ds_bpermute_b32 %[t2], %[s1], %[t0]\n\t"                                                                                 \
    "ds_bpermute_b32 %[t3], %[s2], %[t0]\n\t"                                                                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                                \
    "v_bitop3_b32 %[x], %[t1], %[t2], %[t3] bitop3:0xd2\n\t" /* chi */

// eight rounds; iv[i] / cp[i]: the constant pending at the start of round i
__device__ __forceinline__ void kw_rounds8(u32& x, const u32* iv, const u32* cp, const kw_lane& c) {
    u32 t0, t1, t2, t3, t4;
    asm volatile(BBP_KW_ROUND(i0, c0) BBP_KW_ROUND(i1, c1) BBP_KW_ROUND(i2, c2) BBP_KW_ROUND(i3, c3) BBP_KW_ROUND(i4, c4) BBP_KW_ROUND(i5, c5)
                     BBP_KW_ROUND(i6, c6) BBP_KW_ROUND(i7, c7)
                 : [x] "+v"(x), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4)
                 : [i0] "v"(iv[0]), [c0] "v"(cp[0]), [i1] "v"(iv[1]), [c1] "v"(cp[1]), [i2] "v"(iv[2]), [c2] "v"(cp[2]), [i3] "v"(iv[3]), [c3] "v"(cp[3]),
                   [i4] "v"(iv[4]), [c4] "v"(cp[4]), [i5] "v"(iv[5]), [c5] "v"(cp[5]), [i6] "v"(iv[6]), [c6] "v"(cp[6]), [i7] "v"(iv[7]), [c7] "v"(cp[7]),
                   [sht] "v"(c.sh_theta), [shr] "v"(c.sh_rho), [live] "v"(c.live), [s0] "v"(c.s0), [s1] "v"(c.s1), [s2] "v"(c.s2));
}

// one permutation of the sponge spread over the wavefront (a = this lane's half)
__device__ __forceinline__ u32 kw_keccak_f(u32 a, const kw_lane& c, const kw_iota& k) {
    kw_rounds8(a, k.v, k.cp, c);
    kw_rounds8(a, k.v + 8, k.cp + 8, c);
    kw_rounds8(a, k.v + 16, k.cp + 16, c);
    return a ^ k.v[24];
}
#if defined(BBP_KECCAK_WAVE)
// The permutation of ONE transcript that all 64 lanes of a wavefront hold in lockstep (merlin_transcript::wave; the transcript kernels of
// small launches, prover.hip): the state goes through a 200-byte LDS slot per wavefront -- lane 0 writes its 25 words, every lane picks
// its half-word, the rounds run as in the draw chain, the even-half lanes write the joined words back and every lane reads all 25.
// ~2.4 us instead of 15 for the one-lane permutation.  Must be reached by the whole wavefront (uniform control flow).
__device__ __noinline__ void keccak_f1600_wave(u64* s) {
    __shared__ u64 kx[16][25];  // per wavefront of the workgroup (at most 1024 lanes)
    const u32 L = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (L == 0) {
#pragma unroll
        for (int i = 0; i < 25; i++) kx[w][i] = s[i];
    }
    __syncthreads();
    const kw_lane c = kw_setup(L);
    const kw_iota k = kw_iota_setup(L);
    u32 a = c.live ? kw_half(kx[w][c.word], c.half) : 0u;
    __syncthreads();
    a = kw_keccak_f(a, c, k);
    const auto q = __builtin_amdgcn_permlane32_swap(a, a, false, false);  // the odd halves, for the lanes of the even ones
    if (c.live && c.lower) kx[w][c.word] = kw_join(a, q[1]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 25; i++) s[i] = kx[w][i];
    __syncthreads();
}
#endif
#endif

}  // namespace bbp
