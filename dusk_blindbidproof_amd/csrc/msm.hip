// K1: batched fixed-base multiscalar multiplication over the resident generator table (Pippenger bucketing).
//
// Replaces the point work bulletproofs does through RistrettoPoint::{multiscalar_mul, vartime_multiscalar_mul}
// (dalek constant-time Straus / vartime Straus-or-Pippenger) under Prover::prove and
// Verifier::verify, reached from src/blindbid/proof.rs:88 and src/blindbid/verify.rs:88.  Any evaluation order
// gives the same group element, and the ristretto encoding is canonical, so outputs are byte-identical.
//
// The table holds EVERY power-of-two multiple 2^b * P_i (b = 0..255) of every generator as a 128-byte row of ready-to-use
// limbs, so a scalar can be recoded in width-12 non-adjacent form: odd digits |d| < 2048 at arbitrary bit positions, one
// non-zero digit per 13 bits on average (19.85 per scalar instead of the 24 of aligned 11-bit windows), 1024 buckets, and
// no doublings anywhere.  The hot loop is instruction-issue bound (measured: serving all rows from cache changes its time
// by 4 %), so fewer additions and no unpacking are what count; the generators' 134 MB of rows (275 MB with the padded-range and merged bases) are HBM/MALL resident.
//
// One MSM = one workgroup in each of THREE kernels:
//   k_msm_sort[_staged]  (1024 thin lanes)
//     A. NAF digits of every scalar -> LDS histogram over bucket (|d| + 1) / 2
//     B. exclusive scan -> bucket offsets
//     C. counting-sort scatter of (row index, sign) into the MSM's HBM scratch slice (staged through an LDS image and written out
//        as full lines for MSMs of at most 3000 terms); bucket end offsets to HBM
//   k_msm_acc            (256 fat lanes, 151 registers, two waves per SIMD)
//     D1. the sorted entry array is cut into 256 equal chunks, one per lane: mixed additions of gathered table rows into bucket
//         sums (HBM); a chunk that starts inside a bucket parks its leading partial sum
//   k_msm_fold_half / k_msm_fold  (half a wavefront per MSM for launches of >= 512 MSMs, else 128 lanes)
//     P.  chunk-leading partials into their buckets
//     D2. running-sum fold over the lane's buckets
//     E.  cross-lane fold (shuffles): W = sum_k k S_k and S = sum_k S_k; result = sum_k (2k - 1) S_k = 2 W - S
// A launch of fewer than 128 MSMs cuts each into sub-MSMs (msm_split) that take the same three kernels in their small geometry
// (msm_geom<2>: width-9 digits, 128 buckets, one bucket per fold lane); k_msm_reduce sums the partial results.
#include "context.h"

namespace bbp {

#ifndef BBP_FOLD_PRIO
#define BBP_FOLD_PRIO 2  // wave priority of the fold kernels (accumulate: 0, the other thin kernels: 3); 0 / 1 / 3 measured: no difference, 49.4-49.8 ms per batch either way
#endif
// resident waves per SIMD the register allocator aims the MSM kernels at (512 VGPRs / waves).  Measured on the blind-bid
// batch: 2 (no spills) beats 3 and 4 (spills in the cold phases, nothing gained in the issue-bound hot loop).
#ifndef BBP_MSM_WAVES
#define BBP_MSM_WAVES 2
#endif
// Since the fold phases moved into k_msm_fold (round 2) the accumulate kernel needs 156 VGPRs and no stack at all.  It is still
// held to TWO waves per SIMD (2, 2): they leave 200 VGPRs per SIMD lane for the thin waves of the other streams' kernels
// (transcript, encode, scalar, sort) to run BESIDE the MSM instead of between MSMs; allowing a third accumulate wave was
// measured slower (18.2-18.3 k vs 18.5-18.6 k proofs/s; (2, 4), which lets occupancy float: 18.0-18.1 k).
#ifndef BBP_MSM_WAVES_MAX
#define BBP_MSM_WAVES_MAX 2
#endif

// -DBBP_MSM_PROF (experiments only): per-phase wall-clock (100 MHz) totals of lane 0 of every workgroup, printed every 16 launches
#ifdef BBP_MSM_PROF
__device__ unsigned long long g_msm_prof[8];
#define MSM_PROF_BEGIN() unsigned long long prof_t = wall_clock64()
#define MSM_PROF_MARK(i)                                                    \
    if (tid == 0) {                                                         \
        unsigned long long now_ = wall_clock64();                           \
        atomicAdd(&g_msm_prof[i], now_ - prof_t);                           \
        prof_t = now_;                                                      \
    }
#else
#define MSM_PROF_BEGIN()
#define MSM_PROF_MARK(i)
#endif

// Cross-lane exchange of a point: within a wavefront through ds_bpermute (no LDS storage), between the two wavefronts of a
// workgroup through one 40-word LDS slot.
__device__ __forceinline__ ge ge_shfl_down(const ge& p, int d) {
    ge r;
    const u32* w = reinterpret_cast<const u32*>(&p);
    u32* o = reinterpret_cast<u32*>(&r);
#pragma unroll
    for (int i = 0; i < GE_WORDS; i++) o[i] = (u32)__shfl_down((int)w[i], d, 64);
    return r;
}

__device__ __forceinline__ void xch_put(u32* xch, const ge& p) {
    const u32* w = reinterpret_cast<const u32*>(&p);
#pragma unroll
    for (int i = 0; i < GE_WORDS; i++) xch[i] = w[i];
}

__device__ __forceinline__ ge xch_get(const u32* xch) {
    ge p;
    u32* w = reinterpret_cast<u32*>(&p);
#pragma unroll
    for (int i = 0; i < GE_WORDS; i++) w[i] = xch[i];
    return p;
}

// The index of a gathered row is clamped: a corrupted scratch entry must never turn into an out-of-bounds gather (a GPU fault here
// takes the whole node down); one v_min_u32 per ~1400-instruction iteration.  (row_regs / load_row_at / ge_madd_row: point.h)
__device__ __forceinline__ row_regs load_row(const niels_row* __restrict__ tab, u32 entry, u32* __restrict__ fault) {
    u32 row = entry & 0x7fffffffu;
    if (row > (u32)(TAB_BASES * MSM_POS - 1)) {  // never seen on sound scratch: clamp (no fault) AND say so (bbp_check_health)
        row = (u32)(TAB_BASES * MSM_POS - 1);
        atomicOr(fault, 1u);
    }
#ifdef BBP_EXP_ROWMASK  // experiment (wrong results): alias all gathers onto a cache-resident slice of the table
    row &= BBP_EXP_ROWMASK;
#endif
    return load_row_at(tab + row, entry >> 31);
}

// -DBBP_ACC_COOP: wave-cooperative row fetch (round 4).  The plain path has every lane pull its own 128-byte row in ten pieces of
// 8-16 bytes: each of those wave-instructions touches 64 different cache lines (MI355X_MICROARCH.md, access shape: the slowest
// one).  Here eight consecutive lanes fetch ONE row's 8 x 16 bytes, so a wave-instruction covers 8 whole lines, eight instructions
// a wave's 64 rows -- straight into a per-wave LDS image by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), one
// iteration ahead; a lane then reads its own row with ds_read_b128 / b64.  No workgroup barrier: the image belongs to the wave.
//   image: slot L (128 bytes) = row of lane L; logical 16-byte piece s sits at position s ^ ((L >> 1) & 7), so that the sixteen
//   lanes an LDS b128 read serves together hit sixteen different bank groups (the swizzle is applied on the SOURCE address: an
//   LDS-DMA writes wave-uniform base + lane * 16).
#ifdef BBP_ACC_COOP
constexpr int COOP_IMG = 64 * 128;  // bytes per wavefront
__device__ __forceinline__ u32 coop_clamp_row(u32 entry, u32* __restrict__ fault) {
    u32 row = entry & 0x7fffffffu;
    if (row > (u32)(TAB_BASES * MSM_POS - 1)) {
        row = (u32)(TAB_BASES * MSM_POS - 1);
        atomicOr(fault, 1u);
    }
    return row;
}
// all 64 lanes, EXEC full: lane l fetches piece (l & 7) ^ k of the row that lane 8 j + (l >> 3) will add next
__device__ __forceinline__ void coop_issue(const niels_row* __restrict__ tab, u8* wb, u32 my_row, int lane, u32 srcoff) {
    const u8* t = reinterpret_cast<const u8*>(tab);
    u32 r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = (u32)__shfl((int)my_row, 8 * j + (lane >> 3), 64);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u8* src = t + (size_t)r[j] * 128 + (srcoff ^ (u32)((j & 1) << 6));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(wb + j * 1024), 16, 0, 0);
    }
}
// this lane's row out of the image; a negative digit swaps y+x and y-x (pieces s and s ^ 4), 2dxy stays where it is
__device__ __forceinline__ row_regs coop_read(const u8* wb, u32 A, u32 neg) {
    const u32 An = A ^ (neg << 6);
    const uint4 a0 = *reinterpret_cast<const uint4*>(wb + An), a1 = *reinterpret_cast<const uint4*>(wb + (An ^ 0x10u));
    const uint2 a2 = *reinterpret_cast<const uint2*>(wb + (An ^ 0x20u));
    const uint4 b0 = *reinterpret_cast<const uint4*>(wb + (An ^ 0x40u)), b1 = *reinterpret_cast<const uint4*>(wb + (An ^ 0x50u));
    const uint2 b2 = *reinterpret_cast<const uint2*>(wb + (An ^ 0x60u));
    const uint2 x0 = *reinterpret_cast<const uint2*>(wb + (A ^ 0x20u) + 8);
    const uint4 x1 = *reinterpret_cast<const uint4*>(wb + (A ^ 0x30u));
    const uint2 x2 = *reinterpret_cast<const uint2*>(wb + (A ^ 0x60u) + 8), x3 = *reinterpret_cast<const uint2*>(wb + (A ^ 0x70u));
    row_regs r;
    r.ypx = fe{{(i32)a0.x, (i32)a0.y, (i32)a0.z, (i32)a0.w, (i32)a1.x, (i32)a1.y, (i32)a1.z, (i32)a1.w, (i32)a2.x, (i32)a2.y}};
    r.ymx = fe{{(i32)b0.x, (i32)b0.y, (i32)b0.z, (i32)b0.w, (i32)b1.x, (i32)b1.y, (i32)b1.z, (i32)b1.w, (i32)b2.x, (i32)b2.y}};
    r.xy2d = fe{{(i32)x0.x, (i32)x0.y, (i32)x1.x, (i32)x1.y, (i32)x1.z, (i32)x1.w, (i32)x2.x, (i32)x2.y, (i32)x3.x, (i32)x3.y}};
    return r;
}
#endif

// The cold phases (bucket fold, cross-lane reduction) call ONE out-of-line copy of the point addition / doubling: inlining
// them at every site made the kernel 124 KB, and with workgroups of a CU in different phases the 64 KB instruction cache
// thrashed under the hot mixed-addition loop (measured: that loop ran 1.7x slower than the same code in isolation).
__device__ __noinline__ void ge_add_nc(ge& r, const ge& a, const ge& b) { r = ge_add(a, b); }
__device__ __noinline__ void ge_dbl_nc(ge& r, const ge& a) { r = ge_dbl(a); }

// ---------------------------------------------------------------------------------------------------------------
// The MSM runs as TWO kernels.  The sort half (recoding, histogram, scan, scatter) is latency work -- LDS atomics with return,
// 4-byte scattered stores -- and wants many thin waves; the accumulate half is issue-bound and wants two fat waves per SIMD.
// Fused in one kernel the sort phases ran at the fat kernel's occupancy and took 21 % of its time.
//
// MODE 0: generic MSM -- scalars [n_msm][n], base rows from an index list, key = (|d| + 1) / 2                  (K = 1024)
// MODE 1: generator fold for the IPA tail -- MSM 2p / 2p+1 = the G / H side of proof p, scalars g[p] / h[p], all 2048
//         generators of the side, COMPOSITE key = class(i) * 128 + (|d| + 1) / 2 over width-9 NAF digits        (K = 4096)
// Materialising folded generators (MODE 1): from round FOLD_ROUND on the IPA vectors are at most 32 long, and two 2049-term
// MSMs per round cost far more than working with the 32 + 32 explicit folded generators F_G[i] = sum_{k = i mod 32} g[k] G[k]
// (F_H likewise); one Pippenger pass per side computes all 32 sums, a running-sum fold per class (4 lanes x 32 keys) ends it.
// ---------------------------------------------------------------------------------------------------------------
#ifndef BBP_SORT_T
#define BBP_SORT_T 1024
#endif
constexpr int SORT_T = BBP_SORT_T;  // lanes of the sort workgroup: the kernel is latency-bound, so one scalar or two per lane (measured on
                                    // the 1024-proof batch: 256 lanes 64.1 ms, 512 lanes 62.4 ms, 1024 lanes 61.2 ms per batch)
template <int MODE> struct msm_geom;
template <> struct msm_geom<0> { static constexpr int K = MSM_K, NAF = MSM_NAF, W = MSM_W; };
template <> struct msm_geom<1> { static constexpr int K = FOLD_K, NAF = FOLD_NAF, W = FOLD_W; };
template <> struct msm_geom<2> { static constexpr int K = SMALL_K, NAF = SMALL_NAF, W = SMALL_W; };  // MODE 0's layout with fewer buckets: split MSMs

template <int MODE>
__global__ __launch_bounds__(SORT_T) void k_msm_sort(const u32* __restrict__ scal_a, const u32* __restrict__ aux, u32 n_total, u32 n_idx_sets,
                                                      u32 n_sub, u32 split, u32* __restrict__ sorted_all, u32* __restrict__ cursor_all,
                                                      const u32* __restrict__ msm_map, const u32* __restrict__ n_active, u32* __restrict__ ticket) {
    constexpr int K = msm_geom<MODE>::K, NAF = msm_geom<MODE>::NAF, W = msm_geom<MODE>::W, G = K >= SORT_T ? K / SORT_T : 1;  // buckets per lane of the scan (lanes past K idle)
    __shared__ u32 cursor[K + 1];  // histogram, then bucket start offsets, then (after the scatter) bucket end offsets
    __shared__ u32 part[SORT_T];
    const int tid = threadIdx.x;
    if (ticket && blockIdx.x == 0 && tid == 0) *ticket = 0;  // work counter of the accumulate launch that follows on this stream (BBP_ACC_PERSIST)
    // device-sized launches (MODE 0, split = 1): the grid covers the largest possible number of MSMs, *n_active of them exist
    // (the whole workgroup leaves together), and MSM j takes its scalars from row msm_map[j] of the scalar array
    if (n_active && blockIdx.x >= *n_active) return;
    // small batches: every MSM is cut into `split` sub-MSMs over n_sub consecutive terms (one workgroup each, summed afterwards)
    const size_t work = blockIdx.x, msm = work / split;
    const u32 i0 = (u32)(work % split) * n_sub;
    const u32 n = i0 < n_total ? min(n_sub, n_total - i0) : 0u;
    __builtin_amdgcn_s_setprio(3);  // thin kernel: see BBP_THIN_PRIO in prover.hip
    const u32* sbase;
    const u32* base_idx = nullptr;
    u32 base0 = 0;
    if (MODE != 1) {
        const size_t src = msm_map ? (size_t)msm_map[msm] : msm;
        sbase = scal_a + (src * (size_t)n_total + i0) * 8;
        base_idx = aux + (size_t)(msm % n_idx_sets) * n_total + i0;
    } else {
        const u32 side = (u32)msm & 1u;  // 0: G with g[], 1: H with h[]
        sbase = (side ? aux : scal_a) + ((msm >> 1) * (size_t)2048 + i0) * 8;
        base0 = (side ? BBP_BASE_H0 : BBP_BASE_G0) + i0;
    }
    u32* sorted = sorted_all + work * (size_t)n_sub * W;
    auto key = [&](u32 i, u32 mag) -> u32 { return (MODE != 1 ? 0u : ((i0 + i) & (FOLD_CLS - 1)) * FOLD_M) + ((mag + 1) >> 1); };

    for (int k = tid; k <= K; k += SORT_T) cursor[k] = 0;
    __syncthreads();
    // A. histogram
    for (u32 i = tid; i < n; i += SORT_T) {
        const uint4* sp = reinterpret_cast<const uint4*>(sbase + (size_t)i * 8);
        uint4 lo = sp[0], hi = sp[1];
        const u32 s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (MODE != 1 && base_idx[i] == MSM_SKIP_BASE) continue;  // this scalar rides on another term's merged base
        sc_for_each_naf_digit<NAF>(s, [&](u32, u32 mag, u32) { atomicAdd(&cursor[key(i, mag)], 1u); });
    }
    __syncthreads();
    // B. offsets (in place: count -> exclusive prefix)
    {
        u32 local = 0;
        for (int r = 1; r <= G; r++)
            if (tid * G + r <= K) local += cursor[tid * G + r];
        // block-wide exclusive scan of `local`: inclusive scan inside each wavefront by shuffles, wave totals through LDS
        // (a single lane looping over 1024 partial sums used to cost a fifth of this kernel)
        const int lane = tid & 63, wave = tid >> 6;
        u32 incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 up = (u32)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) part[wave] = incl;
        __syncthreads();
        if (tid < 64) {
            constexpr int NW = SORT_T / 64;
            u32 v = tid < NW ? part[tid] : 0u, sc_ = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u32 up = (u32)__shfl_up((int)sc_, d, 64);
                if (tid >= d) sc_ += up;
            }
            if (tid < NW) part[tid] = sc_ - v;  // exclusive prefix of the wave totals
        }
        __syncthreads();
        u32 base = part[wave] + incl - local;
        for (int r = 1; r <= G; r++) {
            if (tid * G + r > K) break;
            const u32 c = cursor[tid * G + r];
            cursor[tid * G + r] = base;
            base += c;
        }
    }
    __syncthreads();
    // C. scatter
    for (u32 i = tid; i < n; i += SORT_T) {
        const uint4* sp = reinterpret_cast<const uint4*>(sbase + (size_t)i * 8);
        uint4 lo = sp[0], hi = sp[1];
        const u32 s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (MODE != 1 && base_idx[i] == MSM_SKIP_BASE) continue;
        const u32 tb = (MODE != 1 ? base_idx[i] : base0 + i) * MSM_POS;
        sc_for_each_naf_digit<NAF>(s, [&](u32 pos, u32 mag, u32 neg) {
            u32 at = atomicAdd(&cursor[key(i, mag)], 1u);
            sorted[at] = (tb + pos) | (neg << 31);
        });
    }
    __syncthreads();
    u32* cur_out = cursor_all + work * (size_t)(K + 1);
    for (int k = tid; k <= K; k += SORT_T) cur_out[k] = k ? cursor[k] : 0u;  // cursor[k] = end offset of bucket k
}

// k_msm_sort with the counting-sort scatter STAGED THROUGH LDS.  The plain kernel's scatter is 72 % of its time: every entry is a 4-byte
// store to its own cache line, bound by L2 write requests (~2 * 10^11 /s).  Here the buckets are walked in windows of at most SORT_CAP
// entries (consecutive buckets; the window boundaries come from the prefix sums, so every lane finds the same ones): a pass re-walks the
// lane's NAF digits (cheap), places the entries of the window's buckets in an LDS image with the same LDS atomics as before, and the
// workgroup then writes the image out as full cache lines (image: BBP_SORT_CAP entries, 84 KB).  A bucket larger than the image (adversarial scalars only) takes a pass of
// its own with direct stores.  Entry order inside a bucket differs from the plain kernel's; sums do not care.
#ifndef BBP_SORT_CAP
#define BBP_SORT_CAP 21504  // two windows for a 2049-term MSM (40.7 k entries), three for 2933 terms; 16 384 (three / four, two workgroups per CU) measured 0.5 % slower per batch
#endif
template <int MODE> struct sort_cap { static constexpr u32 V = BBP_SORT_CAP; };  // image entries (dynamic LDS: 4 bytes each) of an ordinary launch
template <> struct sort_cap<1> { static constexpr u32 V = 8192; };
constexpr u32 SORT_CAP_WIDE = 32768;  // ... of MSMs with more than SORT_WIDE_FROM terms (one workgroup per CU then)
constexpr u32 SORT_WIDE_FROM = 3000;

template <int MODE>
__global__ __launch_bounds__(SORT_T) void k_msm_sort_staged(const u32* __restrict__ scal_a, const u32* __restrict__ aux, u32 n_total, u32 n_idx_sets,
                                                             u32 n_sub, u32 split, u32* __restrict__ sorted_all, u32* __restrict__ cursor_all,
                                                             const u32* __restrict__ msm_map, const u32* __restrict__ n_active, u32 CAP, u32* __restrict__ ticket) {
    constexpr int K = msm_geom<MODE>::K, NAF = msm_geom<MODE>::NAF, W = msm_geom<MODE>::W, G = K >= SORT_T ? K / SORT_T : 1;  // buckets per lane of the scan (lanes past K idle)
    extern __shared__ u32 stage[];  // [CAP] entries: the image of one window
    __shared__ u32 start[K + 2];  // start[k] = position of bucket k's first entry (k = 1..K), start[K + 1] = number of entries
    __shared__ u32 fill[K + 1];   // histogram, then entries placed so far per bucket
    __shared__ u32 part[SORT_T];
    const int tid = threadIdx.x;
    if (ticket && blockIdx.x == 0 && tid == 0) *ticket = 0;
    if (n_active && blockIdx.x >= *n_active) return;
    const size_t work = blockIdx.x, msm = work / split;
    const u32 i0 = (u32)(work % split) * n_sub;
    const u32 n = i0 < n_total ? min(n_sub, n_total - i0) : 0u;
    __builtin_amdgcn_s_setprio(3);
    const u32* sbase;
    const u32* base_idx = nullptr;
    u32 base0 = 0;
    if (MODE != 1) {
        const size_t src = msm_map ? (size_t)msm_map[msm] : msm;
        sbase = scal_a + (src * (size_t)n_total + i0) * 8;
        base_idx = aux + (size_t)(msm % n_idx_sets) * n_total + i0;
    } else {
        const u32 side = (u32)msm & 1u;
        sbase = (side ? aux : scal_a) + ((msm >> 1) * (size_t)2048 + i0) * 8;
        base0 = (side ? BBP_BASE_H0 : BBP_BASE_G0) + i0;
    }
    u32* sorted = sorted_all + work * (size_t)n_sub * W;
    auto key = [&](u32 i, u32 mag) -> u32 { return (MODE != 1 ? 0u : ((i0 + i) & (FOLD_CLS - 1)) * FOLD_M) + ((mag + 1) >> 1); };

    for (int k = tid; k <= K; k += SORT_T) fill[k] = 0;
    __syncthreads();
    // A. histogram
    for (u32 i = tid; i < n; i += SORT_T) {
        const uint4* sp = reinterpret_cast<const uint4*>(sbase + (size_t)i * 8);
        uint4 lo = sp[0], hi = sp[1];
        const u32 s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (MODE != 1 && base_idx[i] == MSM_SKIP_BASE) continue;
        sc_for_each_naf_digit<NAF>(s, [&](u32, u32 mag, u32) { atomicAdd(&fill[key(i, mag)], 1u); });
    }
    __syncthreads();
    // B. bucket start offsets (block-wide exclusive scan as in k_msm_sort), counters back to zero
    {
        u32 local = 0;
        for (int r = 1; r <= G; r++)
            if (tid * G + r <= K) local += fill[tid * G + r];
        const int lane = tid & 63, wave = tid >> 6;
        u32 incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 up = (u32)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) part[wave] = incl;
        __syncthreads();
        if (tid < 64) {
            constexpr int NW = SORT_T / 64;
            u32 v = tid < NW ? part[tid] : 0u, sc_ = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u32 up = (u32)__shfl_up((int)sc_, d, 64);
                if (tid >= d) sc_ += up;
            }
            if (tid < NW) part[tid] = sc_ - v;
        }
        __syncthreads();
        u32 base = part[wave] + incl - local;
        for (int r = 1; r <= G; r++) {
            if (tid * G + r > K) break;
            const u32 c = fill[tid * G + r];
            start[tid * G + r] = base;
            fill[tid * G + r] = 0;
            base += c;
        }
        if (tid == SORT_T - 1) start[K + 1] = base;  // (lanes past K carry the grand total: `local` is 0 for them)
        if (tid == 0) start[0] = 0;
    }
    __syncthreads();
    // C. scatter, window by window
    u32 kb = 1;
    while (kb <= (u32)K) {
        const u32 base = start[kb];
        const bool direct = start[kb + 1] - base > CAP;  // one bucket larger than the image
        u32 ke = kb + 1;
        if (!direct) {  // largest ke in [kb + 1, K + 1] with start[ke] - base <= CAP
            u32 lo = kb + 1, hi = (u32)K + 1;
            while (lo < hi) {
                const u32 mid = (lo + hi + 1) >> 1;
                if (start[mid] - base <= CAP) lo = mid; else hi = mid - 1;
            }
            ke = lo;
        }
        if (start[ke] != base) {  // (windows of empty buckets have nothing to place)
            for (u32 i = tid; i < n; i += SORT_T) {
                const uint4* sp = reinterpret_cast<const uint4*>(sbase + (size_t)i * 8);
                uint4 lo4 = sp[0], hi4 = sp[1];
                const u32 s[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
                if (MODE != 1 && base_idx[i] == MSM_SKIP_BASE) continue;
                const u32 tb = (MODE != 1 ? base_idx[i] : base0 + i) * MSM_POS;
                sc_for_each_naf_digit<NAF>(s, [&](u32 pos, u32 mag, u32 neg) {
                    const u32 k = key(i, mag);
                    if (k >= kb && k < ke) {
                        const u32 at = start[k] + atomicAdd(&fill[k], 1u);
                        const u32 ent = (tb + pos) | (neg << 31);
                        if (direct) sorted[at] = ent; else stage[at - base] = ent;
                    }
                });
            }
            __syncthreads();
            if (!direct) {
                const u32 cnt = start[ke] - base;
                for (u32 i = tid; i < cnt; i += SORT_T) sorted[base + i] = stage[i];
            }
            __syncthreads();
        }
        kb = ke;
    }
    u32* cur_out = cursor_all + work * (size_t)(K + 1);
    for (int k = tid; k <= K; k += SORT_T) cur_out[k] = k ? start[k + 1] : 0u;  // end offset of bucket k
}

// Workgroups of the accumulate kernel: ACC_WG lanes each, ACC_T / ACC_WG of them per MSM (chunk = global lane index within the MSM).
// Measured: one wavefront per workgroup (64) lets the dispatcher place accumulate waves SIMD by SIMD and shortens the accumulate
// launches by 7 %, but the thin kernels beside them lose as much and more (52.9 vs 51.6 ms per batch): the machine is saturated, a
// kernel only gains what another loses.  One workgroup per MSM stays.
#ifndef BBP_ACC_WG
#define BBP_ACC_WG 256
#endif
constexpr int ACC_WG = BBP_ACC_WG, ACC_WGS = ACC_T / ACC_WG;
static_assert(ACC_T % ACC_WG == 0 && ACC_WG % 64 == 0, "accumulate workgroup geometry");

template <int MODE>
__global__ __launch_bounds__(ACC_WG) __attribute__((amdgpu_waves_per_eu(BBP_MSM_WAVES, BBP_MSM_WAVES_MAX)))
void k_msm_acc(const niels_row* __restrict__ ptable, const u32* __restrict__ sorted_all, const u32* __restrict__ cursor_all, u32 n /* sorted stride */,
               ge* __restrict__ bsum_all, ge* __restrict__ psum_all, ge* __restrict__ out, const u32* __restrict__ n_active, u32* __restrict__ fault,
               u32 n_work, u32* __restrict__ ticket) {
    constexpr int K = msm_geom<MODE>::K, W = msm_geom<MODE>::W, G = K / MSM_T;  // (G: unused here since the fold moved out)
    __shared__ u32 cursor[K + 1];
#ifdef BBP_ACC_PERSIST
    // experiment: at most 2 x CUs workgroups stay resident and draw MSMs from a ticket counter (zeroed by the sort kernel that runs
    // before this launch on the same stream) until none is left: no workgroup dispatch between the MSMs of a launch.  Every wave
    // of a workgroup sees the same ticket (LDS, behind a barrier), so the exit is uniform.
    static_assert(ACC_WGS == 1, "the persistent form draws whole MSMs");
    __shared__ u32 s_ticket;
  for (;;) {
    __syncthreads();  // the previous MSM's cursor[] is no longer read by any wave
    if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const size_t msm = s_ticket;
    if (msm >= n_work || (n_active && msm >= *n_active)) return;
    const int tid = (int)threadIdx.x;
#else
    const size_t msm = blockIdx.x / ACC_WGS;
    const int tid = (int)(blockIdx.x % ACC_WGS) * ACC_WG + (int)threadIdx.x;  // chunk index within the MSM
    if (n_active && msm >= *n_active) return;  // device-sized launch (see k_msm_sort)
#endif
    const u32* sorted = sorted_all + msm * (size_t)n * W;
#ifdef BBP_MSM_PRIO
    __builtin_amdgcn_s_setprio(BBP_MSM_PRIO);
#endif
    MSM_PROF_BEGIN();
    {
        const u32* cur_in = cursor_all + msm * (size_t)(K + 1);
        for (int k = threadIdx.x; k <= K; k += ACC_WG) cursor[k] = cur_in[k];
    }
    __syncthreads();
    MSM_PROF_MARK(1);

    // D1. balanced bucket accumulation: the sorted entry array is cut into 128 EQUAL chunks, one per lane, whatever the
    //     bucket sizes are (a scalar repeated hundreds of times -- the padding rows of the first IPA round -- would
    //     otherwise serialise one lane).  A chunk that starts inside a bucket parks that leading partial sum in psum[lane];
    //     every other bucket (or bucket head) it meets goes to bsum[bucket].
    const u32 E = cursor[K];  // may be 0 (an empty sub-MSM): then no chunk has entries and no bucket is non-empty, E never divides
    ge* bsum = bsum_all + msm * (size_t)K;               // [K] bucket k at index k-1
    ge* psum = psum_all + msm * (size_t)ACC_T;           // [ACC_T]
#ifdef BBP_KO_ACC_PERMILLE  // timing experiment (wrong results): every lane walks only this share of its chunk
    const u32 c0 = (u32)(((u64)tid * E) / ACC_T), c1 = c0 + (u32)(((u64)((u32)(((u64)(tid + 1) * E) / ACC_T) - c0) * BBP_KO_ACC_PERMILLE) / 1000);
#else
    const u32 c0 = (u32)(((u64)tid * E) / ACC_T), c1 = (u32)(((u64)(tid + 1) * E) / ACC_T);
#endif
#ifdef BBP_MSM_PROF
    const unsigned long long clk0 = clock64();
#endif
    u32 k_first = 0;         // bucket holding this lane's first entry
    bool inside = false;     // the chunk starts strictly inside it: its leading partial sum went to psum[tid]
#ifdef BBP_ACC_COOP
    {
        // Same walk, but the trip count is uniform over the workgroup (chunks differ by at most one entry: n_it = ceil(E / ACC_T);
        // a lane whose chunk is through idles in the last trip) because every lane takes part in every cooperative fetch.
        __shared__ __attribute__((aligned(1024))) u8 rowimg[ACC_WG / 64][COOP_IMG];
        const int lane = (int)threadIdx.x & 63;
        u8* wb = rowimg[threadIdx.x >> 6];
        const u32 A = (u32)lane * 128u + ((((u32)lane >> 1) & 7u) << 4);            // reader: slot | swizzle
        const u32 srcoff = (((u32)lane & 7u) ^ (((u32)lane >> 3) >> 1)) << 4;        // loader: piece of the row this lane fetches (j even)
        const u32 n_it = (E + (u32)ACC_T - 1) / (u32)ACC_T, len = c1 - c0;
        u32 k = 1, kend = 0;
        ge* dest = &psum[tid];
        if (len) {
            u32 lo = 1, hi = K;  // bucket containing entry c0: smallest k with cursor[k] > c0
            while (lo < hi) {
                u32 mid = (lo + hi) >> 1;
                if (cursor[mid] > c0) hi = mid; else lo = mid + 1;
            }
            k = lo, kend = cursor[k];
            k_first = lo;
            inside = cursor[k - 1] < c0;
            if (!inside) dest = &bsum[k - 1];
        }
        ge acc = ge_identity();
        u32 ent_cur = len ? sorted[c0] : 0u;
        u32 ent_nxt = len > 1 ? sorted[c0 + 1] : 0u;
        if (n_it) coop_issue(ptable, wb, coop_clamp_row(ent_cur, fault), lane, srcoff);
        for (u32 it = 0; it < n_it; it++) {
            const u32 e = c0 + it;
            const row_regs cur = coop_read(wb, A, ent_cur >> 31);  // (the compiler waits for the LDS-DMA of the previous trip here)
            const bool neg = ent_cur >> 31;
            ent_cur = ent_nxt;
            const bool live = it < len;
            if (live && e == kend) {  // crossed into the next non-empty bucket
                *dest = acc;
                acc = ge_identity();
                do { k++; kend = cursor[k]; } while (kend == e);
                dest = &bsum[k - 1];
            }
            // the image is overwritten by the next trip's rows: this trip's reads must have RETURNED first (an LDS-DMA write is
            // not ordered behind an earlier ds_read by anything but the wave's own wait)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (it + 1 < n_it) coop_issue(ptable, wb, coop_clamp_row(ent_cur, fault), lane, srcoff);
            if (it + 2 < len) ent_nxt = sorted[e + 2]; else ent_nxt = 0u;
            if (live) acc = ge_madd_row(acc, cur, neg);
        }
        if (len) *dest = acc;
    }
#else
    if (c0 < c1) {
        // bucket containing entry c0: smallest k with cursor[k] > c0
        u32 lo = 1, hi = K;
        while (lo < hi) {
            u32 mid = (lo + hi) >> 1;
            if (cursor[mid] > c0) hi = mid; else lo = mid + 1;
        }
        u32 k = lo, kend = cursor[k];
        k_first = lo;
        inside = cursor[k - 1] < c0;
        ge* dest = inside ? &psum[tid] : &bsum[k - 1];
        ge acc = ge_identity();
        // software pipeline, two deep: the entry index is fetched two iterations ahead and the 128-byte table row one
        // iteration ahead, so neither load is waited for before a full mixed addition (~1300 instructions) has run
        u32 ent_cur = sorted[c0];
        u32 ent_nxt = (c0 + 1 < c1) ? sorted[c0 + 1] : 0u;
        row_regs row = load_row(ptable, ent_cur, fault);
        for (u32 e = c0; e < c1; e++) {
            if (e == kend) {  // crossed into the next non-empty bucket
                *dest = acc;
                acc = ge_identity();
                do { k++; kend = cursor[k]; } while (kend == e);
                dest = &bsum[k - 1];
            }
            const row_regs cur = row;
            const bool neg = ent_cur >> 31;
            ent_cur = ent_nxt;
            if (e + 1 < c1) row = load_row(ptable, ent_cur, fault);
            if (e + 2 < c1) ent_nxt = sorted[e + 2];
            acc = ge_madd_row(acc, cur, neg);
        }
        *dest = acc;
    }
#endif
#ifdef BBP_MSM_PROF
    if (tid == 0) atomicAdd(&g_msm_prof[6], clock64() - clk0);
#endif
    MSM_PROF_MARK(2);
    __threadfence_block();
    __syncthreads();
    MSM_PROF_MARK(3);
#ifdef BBP_ACC_PERSIST
  }
#endif
}

// The cold half of an MSM as a kernel of its own: chunk-leading partial sums into their buckets (P), running-sum fold over each
// lane's G buckets (D2), cross-lane fold (E).  It used to be the tail of k_msm_acc, where every point addition had to go through
// ONE out-of-line copy (operands through the stack: 35 calls per lane and MSM, ~19 GB of scratch writes per 1024-proof batch) to
// keep the hot loop's instruction-cache and register footprint; on its own the fold inlines its additions, and k_msm_acc ends
// when its last table row is added.
template <int MODE>
__global__ __launch_bounds__(MSM_T) void k_msm_fold(const u32* __restrict__ cursor_all, ge* __restrict__ bsum_all, const ge* __restrict__ psum_all,
                                                     ge* __restrict__ out, const u32* __restrict__ n_active) {
    constexpr int K = msm_geom<MODE>::K, G = K / MSM_T;
    __shared__ u32 cursor[K + 1];
    __shared__ u32 xch[GE_WORDS];
    const int tid = threadIdx.x;
    if (n_active && blockIdx.x >= *n_active) return;
    __builtin_amdgcn_s_setprio(BBP_FOLD_PRIO);
    const size_t msm = blockIdx.x;
    {
        const u32* cur_in = cursor_all + msm * (size_t)(K + 1);
        for (int k = tid; k <= K; k += MSM_T) cursor[k] = cur_in[k];
    }
    __syncthreads();
    const u32 E = cursor[K];
    ge* bsum = bsum_all + msm * (size_t)K;
    const ge* psum = psum_all + msm * (size_t)ACC_T;
#ifdef BBP_KO_FOLD  // timing experiment (wrong results): the fold does nothing but write some point per output
    if (MODE != 1) { if (tid == 0) out[msm] = bsum[0]; }
    else if (tid < FOLD_CLS) out[msm * FOLD_CLS + tid] = bsum[tid];
    return;
#endif
    // P. the chunk-leading partial sums of the accumulate kernel's ACC_T chunks go into their buckets.  A chunk that started
    //    strictly inside a bucket left that leading partial in psum[chunk]; when several consecutive chunks start inside the same
    //    (heavy) bucket, the LAST of them adds the whole run -- one writer per bucket.  Chunks are dealt to the MSM_T lanes.
    for (u32 ch = tid; ch < (u32)ACC_T; ch += MSM_T) {
        const u32 c0 = (u32)(((u64)ch * E) / ACC_T), c1 = (u32)(((u64)(ch + 1) * E) / ACC_T);
        if (c0 >= c1) continue;
        u32 lo = 1, hi = K;  // the bucket this chunk started in, as k_msm_acc found it
        while (lo < hi) {
            u32 mid = (lo + hi) >> 1;
            if (cursor[mid] > c0) hi = mid; else lo = mid + 1;
        }
        const u32 k_first = lo;
        const bool inside = cursor[lo - 1] < c0;
        const bool last_of_run = inside && !(c1 < E && c1 > cursor[k_first - 1] && c1 < cursor[k_first]);
        if (last_of_run) {
            ge part = psum[ch];
            for (int u = (int)ch - 1; u >= 0; u--) {  // earlier chunks that also start strictly inside this bucket (skewed inputs only)
                const u32 cu = (u32)(((u64)u * E) / ACC_T), cu1 = (u32)(((u64)(u + 1) * E) / ACC_T);
                if (cu <= cursor[k_first - 1]) break;
                if (cu1 > cu) ge_add_nc(part, part, psum[u]);
            }
            ge head = bsum[k_first - 1];
            ge_add_nc(head, head, part);
            bsum[k_first - 1] = head;
        }
    }
    __threadfence_block();
    __syncthreads();

    // D2. running-sum fold over this lane's G buckets (high to low), all lanes in lockstep
    ge running = ge_identity(), total = ge_identity();
    if constexpr (G == 1) {  // one bucket per lane: nothing to run over (two additions to the identity were 13 us of a 122 us fold)
        if (cursor[tid + 1] != cursor[tid]) running = bsum[tid];
        total = running;
    } else {
#pragma unroll 1
        for (int r = G; r >= 1; r--) {
            const u32 k = tid * G + r;
            if (cursor[k] != cursor[k - 1]) running = ge_add(running, bsum[k - 1]);
            total = ge_add(total, running);
        }
    }

    if constexpr (MODE != 1) {
        // E. cross-lane fold: W = sum_k k S_k = sum_t total_t + G * sum_{t>=1} suffix_t, suffix_t = sum_{u>=t} running_u
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll 1
        for (int d = 1; d < 64; d <<= 1) {  // suffix scan inside each wavefront
            ge other = ge_shfl_down(running, d);
            if (lane + d < 64) running = ge_add(running, other);
        }
        if (MSM_T > 64) {
            if (tid == 64) xch_put(xch, running);  // = sum over the upper wavefront
            __syncthreads();
            if (wave == 0) {
                ge other = xch_get(xch);
                ge_add_nc(running, running, other);
            }
        }
        ge x = total;
        if (tid >= 1) {
            ge s = running;
            constexpr int LOG_G = G == 1 ? 0 : G == 2 ? 1 : G == 4 ? 2 : G == 8 ? 3 : G == 16 ? 4 : -1;  // G buckets per lane: W = sum total_t + G * sum_{t>=1} suffix_t
            static_assert(LOG_G >= 0, "buckets per fold lane must be a power of two up to 16");
            for (int i = 0; i < LOG_G; i++) ge_dbl_nc(s, s);
            ge_add_nc(x, x, s);
        }
#pragma unroll 1
        for (int d = 32; d >= 1; d >>= 1) {
            ge other = ge_shfl_down(x, d);
            if (lane < d) x = ge_add(x, other);
        }
        if (MSM_T > 64) {
            __syncthreads();
            if (tid == 64) xch_put(xch, x);
            __syncthreads();
        }
        if (tid == 0) {
            if (MSM_T > 64) {
                ge other = xch_get(xch);
                ge_add_nc(x, x, other);
            }
            ge other;
            // bucket k holds the digit 2k - 1: result = 2 W - S, S = suffix_0 = lane 0's running
            ge_dbl_nc(x, x);
            other = ge_neg(running);
            ge_add_nc(x, x, other);
            out[msm] = x;
        }
    } else {
        // class (LPC lanes x G keys): W = sum_q total_q + G * sum_{q>=1} suffix_q, S = suffix_0; result = 2 W - S
        constexpr int LPC = FOLD_M / G, LOG_G2 = G == 32 ? 5 : 6;
        static_assert(G == 32 || G == 64, "class reduction written for 4 or 2 lanes per class");
        const int q = tid & (LPC - 1);
        ge suf = running;
        for (int d = 1; d < LPC; d++) {
            ge other = ge_shfl_down(running, d);
            if (q + d < LPC) ge_add_nc(suf, suf, other);
        }
        ge x = total;
        if (q >= 1) {
            ge sg = suf;
            for (int i = 0; i < LOG_G2; i++) ge_dbl_nc(sg, sg);
            ge_add_nc(x, x, sg);
        }
        ge accq = x;
        for (int d = 1; d < LPC; d++) {
            ge other = ge_shfl_down(x, d);
            if (q == 0) ge_add_nc(accq, accq, other);
        }
        if (q == 0) {
            ge_dbl_nc(accq, accq);
            ge other = ge_neg(suf);
            ge_add_nc(accq, accq, other);
            out[msm * FOLD_CLS + (tid / LPC)] = accq;
        }
    }
}

// k_msm_fold<0> on HALF a wavefront per MSM (two MSMs per 64-lane workgroup), for launches with many MSMs.  The fold is integer-issue
// work like the accumulation (general additions, 10 multiplications each), and what it costs the pipeline is wave-instructions:
// 128 lanes per MSM spend 2 wavefronts x 38 dependent additions = 76 wave-additions per MSM, most of them in the cross-lane phases that
// only exist because the buckets are spread over many lanes; 32 lanes with 32 buckets each spend (8 + 64 + 17) / 2 = 44.5.  The chain
// per wavefront is 2.3 times longer, so small launches (latency-bound: single proofs, split MSMs) keep the wide kernel.
constexpr int FOLD2_L = 32;                       // lanes per MSM
constexpr int FOLD2_G = MSM_K / FOLD2_L;          // 32 consecutive buckets per lane
constexpr int FOLD2_LOG_G = 5;
static_assert(FOLD2_G == 32 && ACC_T % FOLD2_L == 0, "fold geometry");

__device__ __forceinline__ ge ge_shfl_down32(const ge& p, int d) {  // within each 32-lane half of the wavefront
    ge r;
    const u32* w = reinterpret_cast<const u32*>(&p);
    u32* o = reinterpret_cast<u32*>(&r);
#pragma unroll
    for (int i = 0; i < GE_WORDS; i++) o[i] = (u32)__shfl_down((int)w[i], d, 32);
    return r;
}

// p + q where p2d = 2 d * p.T is already known (an operand that takes part in several additions pays that product once): 8M
__device__ __forceinline__ ge ge_add_t2d(const ge& p, const fe& p2d, const ge& q) {
    fe a = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    fe b = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    fe c = fe_mul(p2d, q.T);
    fe zz = fe_mul(p.Z, q.Z);
    fe d = fe_add(zz, zz);
    fe e = fe_sub(b, a), f = fe_sub(d, c), g = fe_add(d, c), h = fe_add(b, a);
    ge r;
    r.X = fe_mul(e, f);
    r.Y = fe_mul(g, h);
    r.Z = fe_mul(g, f);  // (g first: the four products then share 2 x {e, g} and 19 x {f, h})
    r.T = fe_mul(e, h);
    return r;
}

__device__ __forceinline__ void park_put(u32* park, int tid, const ge& p) {
    const u32* w = reinterpret_cast<const u32*>(&p);
#pragma unroll
    for (int i = 0; i < GE_WORDS; i++) park[i * 64 + tid] = w[i];
}

__device__ __forceinline__ ge park_get(const u32* park, int tid) {
    ge p;
    u32* w = reinterpret_cast<u32*>(&p);
#pragma unroll
    for (int i = 0; i < GE_WORDS; i++) w[i] = park[i * 64 + tid];
    return p;
}

__device__ __forceinline__ ge ld_ge(const ge* p) {
    ge r;
    const uint4* s = reinterpret_cast<const uint4*>(p);
    u32* o = reinterpret_cast<u32*>(&r);
#pragma unroll
    for (int i = 0; i < GE_WORDS / 4; i++) {
        const uint4 v = s[i];
        o[4 * i] = v.x, o[4 * i + 1] = v.y, o[4 * i + 2] = v.z, o[4 * i + 3] = v.w;
    }
    return r;
}

#ifndef BBP_FOLD_WAVES
#define BBP_FOLD_WAVES 2
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BBP_FOLD_WAVES, BBP_FOLD_WAVES))) void k_msm_fold_half(const u32* __restrict__ cursor_all, ge* __restrict__ bsum_all, const ge* __restrict__ psum_all,
                                                       ge* __restrict__ out, u32 n_work, const u32* __restrict__ n_active) {
    constexpr int K = MSM_K, G = FOLD2_G;
    __shared__ u32 cursor2[2][K + 1];
    __shared__ u32 park[GE_WORDS * 64];  // one point per lane, word-major (conflict-free)
    const int tid = threadIdx.x, half = tid >> 5, l = tid & 31;
    const u32 n_tot = n_active ? min(*n_active, n_work) : n_work;
    if (2u * blockIdx.x >= n_tot) return;
    __builtin_amdgcn_s_setprio(BBP_FOLD_PRIO);
    const size_t msm = 2 * (size_t)blockIdx.x + half;
    const bool valid = msm < n_tot;  // an odd launch leaves the last upper half idle: it sees an all-empty cursor and adds identities
    u32* cursor = cursor2[half];
    {
        const u32* cur_in = cursor_all + msm * (size_t)(K + 1);
        for (int k = l; k <= K; k += FOLD2_L) cursor[k] = valid ? cur_in[k] : 0u;
    }
    __syncthreads();
    const u32 E = cursor[K];
    ge* bsum = bsum_all + (valid ? msm : 0) * (size_t)K;
    const ge* psum = psum_all + (valid ? msm : 0) * (size_t)ACC_T;
    // P. chunk-leading partial sums into their buckets (see k_msm_fold): ACC_T / 32 chunks per lane
#pragma unroll 1
    for (u32 ch = l; ch < (u32)ACC_T; ch += FOLD2_L) {
        const u32 c0 = (u32)(((u64)ch * E) / ACC_T), c1 = (u32)(((u64)(ch + 1) * E) / ACC_T);
        bool last_of_run = false;
        u32 k_first = 1;
        if (c0 < c1) {
            u32 lo = 1, hi = K;
            while (lo < hi) {
                u32 mid = (lo + hi) >> 1;
                if (cursor[mid] > c0) hi = mid; else lo = mid + 1;
            }
            k_first = lo;
            const bool inside = cursor[lo - 1] < c0;
            last_of_run = inside && !(c1 < E && c1 > cursor[k_first - 1] && c1 < cursor[k_first]);
        }
        if (last_of_run) {
            ge part = ld_ge(&psum[ch]);
            for (int u = (int)ch - 1; u >= 0; u--) {  // earlier chunks that also start strictly inside this bucket (skewed inputs only)
                const u32 cu = (u32)(((u64)u * E) / ACC_T), cu1 = (u32)(((u64)(u + 1) * E) / ACC_T);
                if (cu <= cursor[k_first - 1]) break;
                if (cu1 > cu) {  // (same addition site as below: the head is added last)
                    const ge more = ld_ge(&psum[u]);
                    part = ge_add(part, more);
                }
            }
            const ge head = ld_ge(&bsum[k_first - 1]);
            bsum[k_first - 1] = ge_add(head, part);
        }
    }
    __threadfence_block();
    __syncthreads();

    // D2. running-sum fold over this lane's 32 buckets (high to low).  Registers are the scarce thing here (this wavefront should fit
    //     beside two accumulate waves: 160 VGPRs): `total` is only touched once per bucket, so it lives in an LDS slot between uses.
    //     `running` is an operand of both additions of a step: its 2d T product is computed once per change and shared (8 + 8 + 1
    //     multiplications per bucket instead of 9 + 9).
    ge running = ge_identity();
    fe r2d = fe_zero();  // 2 d * running.T
    park_put(park, tid, ge_identity());
#pragma unroll 1
    for (int r = G; r >= 1; r--) {
        const u32 k = (u32)l * G + r;
        if (cursor[k] != cursor[k - 1]) {
            const ge cur = ld_ge(&bsum[k - 1]);
            running = ge_add_t2d(running, r2d, cur);
            r2d = fe_mul(running.T, fe_d2());
        }
        const ge total = park_get(park, tid);
        park_put(park, tid, ge_add_t2d(running, r2d, total));
    }

    // E. cross-lane fold inside the half: W = sum_t total_t + G * sum_{t>=1} suffix_t, suffix_t = sum_{u>=t} running_u; result = 2 W - S.
    //    ONE loop with one inlined addition and one inlined doubling: steps 0-4 are the suffix scan of `running`; step 5 is
    //    x = total + 32 * suffix (total leaves the LDS slot, the suffix -- lane 0's is S -- takes it); steps 6-10 the tree sum of x;
    //    step 11 the result 2 x - S in lane 0.
    ge v = running;
#pragma unroll 1
    for (int it = 0; it < 12; it++) {
        ge q;
        bool take;
        int n_dbl = 0;
        if (it == 5) {
            q = v;
            v = park_get(park, tid);
            park_put(park, tid, q);
            n_dbl = FOLD2_LOG_G;
            take = l >= 1;
        } else if (it == 11) {
            q = ge_neg(park_get(park, tid));
            take = true;
        } else {
            const int d = it < 5 ? (1 << it) : (16 >> (it - 6));
            q = ge_shfl_down32(v, d);
            take = it < 5 ? (l + d < 32) : (l < d);
        }
        if (it == 11) {
            v = ge_dbl(v);
        } else {
#pragma unroll 1
            for (int i = 0; i < n_dbl; i++) q = ge_dbl(q);
        }
        if (take) v = ge_add(v, q);
    }
    if (l == 0 && valid) out[msm] = v;
}

// sums the `split` (<= 16) partial results of every MSM: out[o] = sum_j tmp[((o / items) * split + j) * items + o % items].
// Sixteen lanes per output, one partial each, then a shuffle tree: four dependent additions instead of fifteen (round 4: this kernel
// sits on the serial chain of every small call -- a single proof has 2 outputs x 16 partials per IPA round -- and one lane adding
// fifteen points in a row was 40-60 us of it).  All lanes of a wavefront stay in the loop (the shuffles need them); lanes past n_out or
// past `split` carry the identity.
constexpr int REDUCE_L = 16;
__global__ __launch_bounds__(64) void k_msm_reduce(u32 n_out, u32 split, u32 items, const ge* __restrict__ tmp, ge* __restrict__ out) {
    __builtin_amdgcn_s_setprio(3);
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 o = t / REDUCE_L, j = t % REDUCE_L;
    const u32 msm = o / items, c = o % items;
    ge acc = ge_identity();
    if (o < n_out && j < split) acc = tmp[((size_t)msm * split + j) * items + c];
    for (u32 jj = j + REDUCE_L; o < n_out && jj < split; jj += REDUCE_L)  // (split <= 16 today; correct for more)
        acc = ge_add(acc, tmp[((size_t)msm * split + jj) * items + c]);
#pragma unroll 1
    for (int d = REDUCE_L / 2; d >= 1; d >>= 1) {
        const ge other = ge_shfl_down(acc, d);  // within the 16-lane group: lane j + d < 16 whenever j < d
        if ((int)j < d) acc = ge_add(acc, other);
    }
    if (o < n_out && j == 0) out[o] = acc;
}

// how many workgroups an MSM of n terms is cut into when the launch has only n_msm of them (fills the GPU for small batches)
static u32 msm_split(u32 n_msm, u32 n_terms) {
    // every sub-MSM pays a bucket fold of its own (k_msm_fold), so splitting only pays while the GPU would otherwise be mostly
    // empty.  BBP_MSM_SPLIT_BELOW (launches with fewer MSMs than this are split) / BBP_MSM_SPLIT_TARGET (into about this many
    // workgroups) are read once.
    static const u32 below = [] { const char* e = getenv("BBP_MSM_SPLIT_BELOW"); return e ? (u32)atoi(e) : 128u; }();
    static const u32 target = [] { const char* e = getenv("BBP_MSM_SPLIT_TARGET"); return e ? (u32)atoi(e) : 512u; }();
    if (n_msm >= below) return 1;
    u32 s = target / n_msm;
    if (s > 16) s = 16;
    while (s > 1 && n_terms / s < 128) s--;
    return s ? s : 1;
}

// scratch of one launch: sorted entries (n * W u32 per MSM) | bucket end offsets (K + 1 u32) | bucket sums (K points) |
// chunk-leading partial sums (T points)
struct MsmScratch {
    u32 *sorted, *cursor;
    ge *bsum, *psum, *tmp;  // tmp: partial results of split MSMs
    u32* ticket;            // (BBP_ACC_PERSIST) the accumulate launch's work counter: zeroed by the sort kernel ahead of it
    size_t bytes;
};
static MsmScratch msm_scratch_layout(void* base, size_t n_msm, size_t n_terms, size_t W, size_t K, size_t out_items = 1) {
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    MsmScratch m;
    size_t o = 0;
    m.sorted = reinterpret_cast<u32*>(static_cast<u8*>(base) + o);
    o += up(n_msm * n_terms * W * sizeof(u32));
    m.cursor = reinterpret_cast<u32*>(static_cast<u8*>(base) + o);
    o += up(n_msm * (K + 1) * sizeof(u32));
    m.bsum = reinterpret_cast<ge*>(static_cast<u8*>(base) + o);
    o += n_msm * K * sizeof(ge);
    m.psum = reinterpret_cast<ge*>(static_cast<u8*>(base) + o);
    o += n_msm * ACC_T * sizeof(ge);
    m.tmp = reinterpret_cast<ge*>(static_cast<u8*>(base) + o);
    o += up(n_msm * out_items * sizeof(ge));
    m.ticket = reinterpret_cast<u32*>(static_cast<u8*>(base) + o);
    o += 256;
    m.bytes = o;
    return m;
}
size_t msm_scratch_bytes(uint32_t n_msm, uint32_t n_terms) {
    const u32 split = msm_split(n_msm, n_terms), n_sub = (n_terms + split - 1) / split;
    size_t a = msm_scratch_layout(nullptr, (size_t)n_msm * split, n_sub, MSM_W, MSM_K).bytes;
    if (split > 1) a = std::max(a, msm_scratch_layout(nullptr, (size_t)n_msm * split, n_sub, SMALL_W, SMALL_K).bytes);  // split MSMs: the small geometry (or, BBP_MSM_SMALL=0, the large one)
    const size_t b = msm_scratch_layout(nullptr, n_msm, n_terms, MSM_W, MSM_K).bytes;  // the unsplit layout of a device-sized launch
    return a > b ? a : b;
}

// workgroups of an accumulate launch: one per MSM (x ACC_WGS); the persistent experiment keeps at most two per CU and lets them draw
static inline u32 acc_grid(bbp_ctx*, u32 n_work) {
#ifdef BBP_ACC_PERSIST
    static const u32 cap = [] { const char* e = getenv("BBP_ACC_PERSIST_GRID"); return e ? (u32)atoi(e) : 512u; }();
    return n_work < cap ? n_work : cap;
#else
    return n_work * ACC_WGS;
#endif
}

int32_t fold_generators_launch(bbp_ctx* ctx, uint32_t n_proofs, const sc* g_dev, const sc* h_dev, ge* out_dev, hipStream_t stream,
                               int scratch_slot) {
    if (n_proofs == 0) return BBP_OK;
    const size_t n_msm = 2 * (size_t)n_proofs;
    const u32 split = msm_split((u32)n_msm, 2048), n_sub = (2048 + split - 1) / split;
    const size_t n_work = n_msm * split;
    DevBuf& scratch = ctx->slice_fold[scratch_slot];
    int32_t rc = dev_reserve(ctx, scratch, msm_scratch_layout(nullptr, n_work, n_sub, FOLD_W, FOLD_K, FOLD_CLS).bytes);
    if (rc) return rc;
    const MsmScratch m = msm_scratch_layout(scratch.p, n_work, n_sub, FOLD_W, FOLD_K, FOLD_CLS);
    {
        ScopedEvent ev(ctx, TAG_MSM_SORT, stream);
        if (ctx->sort_staged & 2) {
            if (!ctx->sort_lds_attr1) {  // 36 KB static + 32 KB dynamic LDS: above the 64 KB a kernel gets without opting in (per context = per device)
                BBP_HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_msm_sort_staged<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sort_cap<1>::V * 4)));
                ctx->sort_lds_attr1 = true;
            }
            hipLaunchKernelGGL(k_msm_sort_staged<1>, dim3((u32)n_work), dim3(SORT_T), sort_cap<1>::V * 4, stream, (const u32*)g_dev, (const u32*)h_dev, 2048u, 1u,
                               n_sub, split, m.sorted, m.cursor, (const u32*)nullptr, (const u32*)nullptr, sort_cap<1>::V, m.ticket);
        } else
            hipLaunchKernelGGL(k_msm_sort<1>, dim3((u32)n_work), dim3(SORT_T), 0, stream, (const u32*)g_dev, (const u32*)h_dev, 2048u, 1u, n_sub, split,
                               m.sorted, m.cursor, (const u32*)nullptr, (const u32*)nullptr, m.ticket);
        BBP_HIP_TRY(ctx, hipGetLastError());
    }
    {
        ScopedEvent ev(ctx, TAG_MSM, stream);
        hipLaunchKernelGGL(k_msm_acc<1>, dim3(acc_grid(ctx, (u32)n_work)), dim3(ACC_WG), 0, stream, ctx->ptable, m.sorted, m.cursor, n_sub, m.bsum, m.psum,
                           split > 1 ? m.tmp : out_dev, (const u32*)nullptr, ctx->health, (u32)n_work, m.ticket);
        BBP_HIP_TRY(ctx, hipGetLastError());
    }
    ScopedEvent evf(ctx, TAG_MSM_FOLD, stream);
    hipLaunchKernelGGL(k_msm_fold<1>, dim3((u32)n_work), dim3(MSM_T), 0, stream, m.cursor, m.bsum, m.psum, split > 1 ? m.tmp : out_dev,
                       (const u32*)nullptr);
    BBP_HIP_TRY(ctx, hipGetLastError());
    if (split > 1) {
        const u32 n_out = (u32)n_msm * FOLD_CLS;
        hipLaunchKernelGGL(k_msm_reduce, dim3((n_out * REDUCE_L + 63) / 64), dim3(64), lds_token(ctx), stream, n_out, split, (u32)FOLD_CLS, m.tmp, out_dev);
        BBP_HIP_TRY(ctx, hipGetLastError());
    }
    return BBP_OK;
}

__global__ __launch_bounds__(64) void k_encode(const ge* __restrict__ pts, u32 n, u32* __restrict__ out_words) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ge p = pts[i];
    u32 w[8];
    ge_encode_words(w, p);
    uint4* o = reinterpret_cast<uint4*>(out_words + (size_t)i * 8);
    o[0] = make_uint4(w[0], w[1], w[2], w[3]);
    o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// test hook (bbp_debug_corrupt_scratch): what a stray write into the engine's scratch would look like to the accumulate kernel
__global__ void k_debug_poke(u32* sorted) { sorted[0] = 0x7ffffff0u; }

int32_t msm_launch(bbp_ctx* ctx, uint32_t n_msm, uint32_t n_terms, const u32* scalars_dev, const u32* base_idx_dev,
                   ge* out_points_dev, hipStream_t stream, uint32_t n_idx_sets, int scratch_slot, const u32* msm_map_dev,
                   const u32* n_active_dev, bool chain_acc) {
    if (n_msm == 0) return BBP_OK;
    if (n_terms == 0 || n_terms > 65535u) {
        ctx->err = "msm_launch: n_terms out of range";
        return BBP_ERR_BAD_ARG;
    }
    DevBuf& scratch = scratch_slot ? ctx->slice_sorted[scratch_slot] : ctx->sorted;  // one scratch area per concurrently running stream
    int32_t rc = dev_reserve(ctx, scratch, msm_scratch_bytes(n_msm, n_terms));
    if (rc) return rc;
    // device-sized launches (n_active_dev: n_msm is only the upper bound of how many MSMs there are) are never split
    const u32 split = n_active_dev ? 1u : msm_split(n_msm, n_terms), n_sub = (n_terms + split - 1) / split;
    const u32 n_work = n_msm * split;
    if (split > 1 && ctx->msm_small) {  // BBP_MSM_SMALL=0: split MSMs keep the 1024-bucket geometry
        // SPLIT MSMs (small batches): width-9 digits into 128 buckets (context.h SMALL_*): same kernels, MODE 2
        const MsmScratch ms = msm_scratch_layout(scratch.p, n_work, n_sub, SMALL_W, SMALL_K);
        {
            ScopedEvent ev(ctx, TAG_MSM_SORT, stream);
            hipLaunchKernelGGL(k_msm_sort<2>, dim3(n_work), dim3(SORT_T), 0, stream, scalars_dev, base_idx_dev, n_terms, n_idx_sets, n_sub, split, ms.sorted, ms.cursor,
                               msm_map_dev, n_active_dev, ms.ticket);
            BBP_HIP_TRY(ctx, hipGetLastError());
        }
        if (ctx->debug_corrupt) {
            ctx->debug_corrupt = 0;
            hipLaunchKernelGGL(k_debug_poke, dim3(1), dim3(1), 0, stream, ms.sorted);
            BBP_HIP_TRY(ctx, hipGetLastError());
        }
        {
            ScopedEvent ev(ctx, TAG_MSM, stream);
            hipLaunchKernelGGL(k_msm_acc<2>, dim3(acc_grid(ctx, n_work)), dim3(ACC_WG), 0, stream, ctx->ptable, ms.sorted, ms.cursor, n_sub, ms.bsum, ms.psum, ms.tmp,
                               n_active_dev, ctx->health, n_work, ms.ticket);
            BBP_HIP_TRY(ctx, hipGetLastError());
        }
        {
            ScopedEvent evf(ctx, TAG_MSM_FOLD, stream);
            hipLaunchKernelGGL(k_msm_fold<2>, dim3(n_work), dim3(MSM_T), 0, stream, ms.cursor, ms.bsum, ms.psum, ms.tmp, n_active_dev);
            BBP_HIP_TRY(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL(k_msm_reduce, dim3((n_msm * REDUCE_L + 63) / 64), dim3(64), lds_token(ctx), stream, n_msm, split, 1u, ms.tmp, out_points_dev);
        BBP_HIP_TRY(ctx, hipGetLastError());
        return BBP_OK;
    }
    const MsmScratch m = msm_scratch_layout(scratch.p, n_work, n_sub, MSM_W, MSM_K);
    {
        ScopedEvent ev(ctx, TAG_MSM_SORT, stream);
        // staged scatter: 84 KB image for the prover's 2049- / 2933-term MSMs (two / three windows); wider MSMs (the
        // verifier's 4098 terms = 81 k entries: six windows, measured 3 % slower than the plain scatter) get a 128 KB image with bit 2 of the knob
        const bool wide = n_sub > SORT_WIDE_FROM;
        if ((ctx->sort_staged & 1) && (!wide || (ctx->sort_staged & 4))) {
            const u32 cap = wide ? SORT_CAP_WIDE : sort_cap<0>::V;
            if (!ctx->sort_lds_attr) {  // per context = per device: a process may hold one context per GPU (bbp-uds-server --devices)
                BBP_HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_msm_sort_staged<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SORT_CAP_WIDE * 4)));
                ctx->sort_lds_attr = true;
            }
            hipLaunchKernelGGL(k_msm_sort_staged<0>, dim3(n_work), dim3(SORT_T), cap * 4, stream, scalars_dev, base_idx_dev, n_terms, n_idx_sets, n_sub, split,
                               m.sorted, m.cursor, msm_map_dev, n_active_dev, cap, m.ticket);
        }
        else
            hipLaunchKernelGGL(k_msm_sort<0>, dim3(n_work), dim3(SORT_T), 0, stream, scalars_dev, base_idx_dev, n_terms, n_idx_sets, n_sub, split,
                               m.sorted, m.cursor, msm_map_dev, n_active_dev, m.ticket);
        BBP_HIP_TRY(ctx, hipGetLastError());
    }
    if (ctx->debug_corrupt) {  // bbp_debug_corrupt_scratch: the next MSM launch finds an out-of-range entry in its sorted scratch
        ctx->debug_corrupt = 0;
        hipLaunchKernelGGL(k_debug_poke, dim3(1), dim3(1), 0, stream, m.sorted);
        BBP_HIP_TRY(ctx, hipGetLastError());
    }
    chain_acc = chain_acc && ctx->verify_serial_acc;
    if (chain_acc && ctx->vacc_valid)  // (the sort above ran beside the previous accumulate; this one starts when that one is through)
        BBP_HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->ev_vacc[ctx->vacc_seq % bbp_ctx::VACC_RING], 0));
    {
        ScopedEvent ev(ctx, TAG_MSM, stream);
        hipLaunchKernelGGL(k_msm_acc<0>, dim3(acc_grid(ctx, n_work)), dim3(ACC_WG), 0, stream, ctx->ptable, m.sorted, m.cursor, n_sub, m.bsum, m.psum,
                           split > 1 ? m.tmp : out_points_dev, n_active_dev, ctx->health, n_work, m.ticket);
        BBP_HIP_TRY(ctx, hipGetLastError());
    }
    static const bool chain_after_fold = getenv("BBP_VERIFY_CHAIN_AFTER_FOLD") && atoi(getenv("BBP_VERIFY_CHAIN_AFTER_FOLD"));
    if (chain_acc && !chain_after_fold) {
        ctx->vacc_seq++;
        BBP_HIP_TRY(ctx, hipEventRecord(ctx->ev_vacc[ctx->vacc_seq % bbp_ctx::VACC_RING], stream));
        ctx->vacc_valid = true;
    }
    ScopedEvent evf(ctx, TAG_MSM_FOLD, stream);
    // many MSMs: the fold on half a wavefront per MSM (fewer wave-instructions); few: the 128-lane fold (shorter chain).  BBP_FOLD_HALF_FROM
    if (n_work >= (u32)ctx->fold_half_from)
        hipLaunchKernelGGL(k_msm_fold_half, dim3((n_work + 1) / 2), dim3(64), 0, stream, m.cursor, m.bsum, m.psum, split > 1 ? m.tmp : out_points_dev,
                           n_work, n_active_dev);
    else
        hipLaunchKernelGGL(k_msm_fold<0>, dim3(n_work), dim3(MSM_T), 0, stream, m.cursor, m.bsum, m.psum, split > 1 ? m.tmp : out_points_dev, n_active_dev);
    BBP_HIP_TRY(ctx, hipGetLastError());
    if (chain_acc && chain_after_fold) {  // experiment: the next lane's accumulate waits for this lane's fold as well (222 registers: it crawls beside two accumulate waves)
        ctx->vacc_seq++;
        BBP_HIP_TRY(ctx, hipEventRecord(ctx->ev_vacc[ctx->vacc_seq % bbp_ctx::VACC_RING], stream));
        ctx->vacc_valid = true;
    }
    if (split > 1) {
        hipLaunchKernelGGL(k_msm_reduce, dim3((n_msm * REDUCE_L + 63) / 64), dim3(64), lds_token(ctx), stream, n_msm, split, 1u, m.tmp, out_points_dev);
        BBP_HIP_TRY(ctx, hipGetLastError());
    }
#ifdef BBP_MSM_PROF
    static int launches = 0;
    if (++launches % 16 == 0) {
        unsigned long long h[8];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_msm_prof), sizeof h);
        fprintf(stderr, "[msm prof, %d launches, 10ns ticks of lane 0 summed over WGs] load %llu D1 %llu D1wait %llu D2 %llu E %llu; D1 shader clock %.0f MHz\n", launches,
                h[1], h[2], h[3], h[4], h[5], 100.0 * (double)h[6] / (double)h[2]);
    }
#endif
    return BBP_OK;
}

int32_t encode_launch(bbp_ctx* ctx, uint32_t n, const ge* pts_dev, uint8_t* out32_dev, hipStream_t stream) {
    if (n == 0) return BBP_OK;
    ScopedEvent ev(ctx, TAG_ENCODE, stream);
    hipLaunchKernelGGL(k_encode, dim3((n + 63) / 64), dim3(64), lds_token(ctx), stream, pts_dev, n, (u32*)out32_dev);
    BBP_HIP_TRY(ctx, hipGetLastError());
    return BBP_OK;
}

}  // namespace bbp
