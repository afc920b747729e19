// Experiment (GPU box): can the mixed-addition loop of the MSM be fed with table rows gathered at random from a table far larger
// than the Infinity Cache?  Decides whether per-generator WINDOW tables (every multiple d * 2^(14 w) * P tabulated: tens of GB, no
// buckets, no sort, 18 additions per scalar) are viable on MI355X, or whether HBM / TLB behaviour of 128-byte random rows is.
//   hipcc --offload-arch=gfx950 -O3 -I dusk_blindbidproof_amd/csrc tools/exp_gather.hip -o gpurun_out/exp_gather
//   gpurun_out/exp_gather <table GiB> [lanes per WG = 256] [workgroups = 4096] [iterations = 152]
// Same row layout, loads, software pipeline and mixed addition as k_msm_acc (msm.hip); rows are picked by a per-lane hash.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "point.h"

using namespace bbp;

struct row_regs {
    fe ypx, ymx, xy2d;
};

__device__ __forceinline__ fe load_fe40(const u8* p) {
    const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 16);
    const uint2 c = *reinterpret_cast<const uint2*>(p + 32);
    return fe{{(i32)a.x, (i32)a.y, (i32)a.z, (i32)a.w, (i32)b.x, (i32)b.y, (i32)b.z, (i32)b.w, (i32)c.x, (i32)c.y}};
}

__device__ __forceinline__ row_regs load_row(const u8* __restrict__ tab, u64 row, u32 neg) {
    const u8* p = tab + row * 128;
    const u32 swap = neg << 6;
    row_regs r;
    r.ypx = load_fe40(p + swap);
    r.ymx = load_fe40(p + (swap ^ 64u));
    const uint2 x0 = *reinterpret_cast<const uint2*>(p + 40);
    const uint4 x1 = *reinterpret_cast<const uint4*>(p + 48);
    const uint2 x2 = *reinterpret_cast<const uint2*>(p + 104), x3 = *reinterpret_cast<const uint2*>(p + 112);
    r.xy2d = fe{{(i32)x0.x, (i32)x0.y, (i32)x1.x, (i32)x1.y, (i32)x1.z, (i32)x1.w, (i32)x2.x, (i32)x2.y, (i32)x3.x, (i32)x3.y}};
    return r;
}

__device__ __forceinline__ ge ge_madd_row(const ge& p, const row_regs& q, bool neg) {
    fe a = fe_mul(fe_sub(p.Y, p.X), q.ymx);
    fe b = fe_mul(fe_add(p.Y, p.X), q.ypx);
    fe c = fe_mul(p.T, q.xy2d);
    fe d = fe_add(p.Z, p.Z);
    fe e = fe_sub(b, a), h = fe_add(b, a);
    fe f0 = fe_sub(d, c), g0 = fe_add(d, c);
    fe f = fe_select(f0, g0, neg), g = fe_select(g0, f0, neg);
    ge r;
    r.X = fe_mul(e, f);
    r.Y = fe_mul(g, h);
    r.Z = fe_mul(g, f);  // (g first: the four products then share 2 x {e, g} and 19 x {f, h})
    r.T = fe_mul(e, h);
    return r;
}

__device__ __forceinline__ u32 mix(u32 x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gather(const u8* __restrict__ tab, u32 n_rows, u32 iters,
                                                                                              u32* __restrict__ sink) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 h = mix(t * 2654435761u + 12345u);
    auto pick = [&](u32 x) -> u64 { return ((u64)x * n_rows) >> 32; };
    ge acc = ge_identity();
    u32 h_cur = h;
    row_regs row = load_row(tab, pick(h_cur), h_cur & 1u);
    for (u32 i = 0; i < iters; i++) {
        const row_regs cur = row;
        const bool neg = h_cur & 1u;
        h_cur = mix(h_cur + 0x9e3779b9u);
        if (i + 1 < iters) row = load_row(tab, pick(h_cur), h_cur & 1u);
        acc = ge_madd_row(acc, cur, neg);
    }
    sink[t] = (u32)(acc.X.v[0] ^ acc.T.v[3] ^ acc.Y.v[5] ^ acc.Z.v[7]);
}


#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

// second mode: `exp_gather cumask` -- the same loop (64 MB table) on streams restricted to 64 / 128 / 192 / 256 CUs: does the rate grow with the
// CU count (issue-bound) or stay put (power / clock bound)?
static int cumask_mode() {
    const size_t bytes = 64u << 20;
    const u32 n_rows = (u32)(bytes / 128);
    u8* tab = nullptr;
    u32* sink = nullptr;
    const int wgs = 8192;
    CK(hipMalloc(&tab, bytes));
    CK(hipMalloc(&sink, (size_t)wgs * 256 * 4));
    CK(hipMemset(tab, 1, bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; rep++)
        for (int cus : {32, 64, 128, 192, 224, 256}) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            // CU i of the mask: spread over the 8 XCDs round-robin the way the runtime numbers them (bit i = CU i)
            for (int i = 0; i < cus; i++) {
                const int cu = (i % 8) * 32 + i / 8;  // XCD-major numbering assumed: take CUs evenly from every XCD
                mask[cu / 32] |= 1u << (cu % 32);
            }
            hipStream_t st;
            CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
            hipLaunchKernelGGL(k_gather, dim3(wgs), dim3(256), 0, st, tab, n_rows, 40u, sink);  // warm
            CK(hipEventRecord(a, st));
            hipLaunchKernelGGL(k_gather, dim3(wgs), dim3(256), 0, st, tab, n_rows, 400u, sink);
            CK(hipEventRecord(b, st));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            const double adds = (double)wgs * 256.0 * 400;
            printf("CUs %3d: %.2f ms  %.3e additions/s  %.3e per CU\n", cus, ms, adds / (ms * 1e-3), adds / (ms * 1e-3) / cus);
            fflush(stdout);
            CK(hipStreamDestroy(st));
        }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "cumask")) return cumask_mode();
    const double gib = argc > 1 ? atof(argv[1]) : 0.25;
    const int wgs = argc > 2 ? atoi(argv[2]) : 4096;
    const u32 iters = argc > 3 ? (u32)atoi(argv[3]) : 152u;
    const size_t bytes = (size_t)(gib * 1024.0 * 1024.0 * 1024.0) / 128 * 128;
    const u32 n_rows = (u32)(bytes / 128);
    u8* tab = nullptr;
    u32* sink = nullptr;
    CK(hipMalloc(&tab, bytes));
    CK(hipMalloc(&sink, (size_t)wgs * 256 * 4));
    CK(hipMemset(tab, 1, bytes));
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_gather, dim3(wgs), dim3(256), 0, 0, tab, n_rows, iters, sink);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        const double adds = (double)wgs * 256.0 * iters;
        printf("table %.2f GiB (%u rows)  wgs %d  iters %u: %.3f ms  %.3e additions/s  %.2f TB/s of rows\n", gib, n_rows, wgs, iters, ms,
               adds / (ms * 1e-3), adds * 128.0 / (ms * 1e-3) / 1e12);
    }
    fflush(stdout);
    CK(hipFree(tab));
    CK(hipFree(sink));
    return 0;
}
