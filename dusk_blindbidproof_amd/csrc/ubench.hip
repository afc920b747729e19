// Integer-ALU roofline microbenchmarks (diagnostic ABI, include/bbp.h bbp_ubench): the blind-bid path is bound by 32-bit
// integer multiply issue (v_mad_i64_i32 / v_mad_u64_u32), not by HBM, so the honest ceiling for K1 is "mixed point additions
// per second the chip can issue", measured here with register-resident dependent chains and no memory traffic.
//   kind 0: raw v_mad_u64_u32 chain (4 independent accumulators per lane)      -> mads/s
//   kind 1: fe_mul chains (2 per lane)                                            -> field multiplications/s
//   kind 2: fe_sq chains                                                          -> field squarings/s
//   kind 3: ge_madd chain (register-resident cached point, no memory)            -> mixed additions/s
//   kind 4: sc_montmul chains                                                     -> Montgomery products mod l /s
//   kind 5: ge_madd chain whose cached point is opaque to the compiler each iteration (as a freshly gathered table row
//           is: 19 * limb precomputations cannot be hoisted)                       -> mixed additions/s
#include <stdio.h>
#include <stdlib.h>

#include "context.h"

namespace bbp {

__device__ unsigned long long g_ubench_clk[2];  // shader-clock and 100 MHz wall-clock ticks of lane 0's loop (BBP_UBENCH_CLK=1 prints MHz)

__global__ __launch_bounds__(256) void k_ubench(int kind, u32 iters, u32* __restrict__ sink) {
    extern __shared__ u32 occupancy_ballast[];  // dynamic LDS only throttles waves/SIMD (BBP_UBENCH_LDS)
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long clk0 = clock64(), wall0 = wall_clock64();
    if (iters == 0xffffffffu) occupancy_ballast[threadIdx.x] = t;
    u32 acc = 0;
    if (kind == 0) {
        u64 a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3;
        u32 x = t * 2654435761u + 1, y = t ^ 0x9e3779b9u;
        for (u32 i = 0; i < iters; i++) {
            a0 = (u64)x * (u32)a0 + a1;
            a1 = (u64)y * (u32)a1 + a2;
            a2 = (u64)x * (u32)a2 + a3;
            a3 = (u64)y * (u32)a3 + a0;
        }
        acc = (u32)(a0 ^ a1 ^ a2 ^ a3);
    } else if (kind == 1 || kind == 2) {
        fe a = fe_d(), b = fe_sqrt_m1();
        a.v[0] ^= (i32)(t & 0xffff);
        b.v[1] ^= (i32)(t & 0xffff);
        for (u32 i = 0; i < iters; i++) {
            if (kind == 1) {
                a = fe_mul(a, b);
                b = fe_mul(b, a);
            } else {
                a = fe_sq(a);
                b = fe_sq(b);
            }
        }
        acc = (u32)(a.v[0] ^ b.v[3]);
    } else if (kind == 3) {
        ge p = ge_basepoint();
        p.X.v[0] ^= (i32)(t & 0xffff);
        ge_niels n;
        n.ypx = fe_d();
        n.ymx = fe_d2();
        n.xy2d = fe_sqrt_m1();
        for (u32 i = 0; i < iters; i++) p = ge_madd(p, n);
        acc = (u32)(p.X.v[0] ^ p.T.v[2]);
    } else if (kind == 5) {
        ge p = ge_basepoint();
        p.X.v[0] ^= (i32)(t & 0xffff);
        ge_niels n;
        n.ypx = fe_d();
        n.ymx = fe_d2();
        n.xy2d = fe_sqrt_m1();
        for (u32 i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 10; k++) {
                asm volatile("" : "+v"(n.ypx.v[k]));
                asm volatile("" : "+v"(n.ymx.v[k]));
                asm volatile("" : "+v"(n.xy2d.v[k]));
            }
            p = ge_madd(p, n);
        }
        acc = (u32)(p.X.v[0] ^ p.T.v[2]);
    } else {
        sc a = sc_rr(), b = sc_r();
        a.v[0] ^= (t & 0xff);
        for (u32 i = 0; i < iters; i++) {
            a = sc_montmul(a, b);
            b = sc_montmul(b, a);
        }
        acc = a.v[0] ^ b.v[1];
    }
    sink[t] = acc;
    if (t == 0) {
        g_ubench_clk[0] = clock64() - clk0;
        g_ubench_clk[1] = wall_clock64() - wall0;
    }
}

}  // namespace bbp

using namespace bbp;

// Runs `blocks` x 256 lanes x `iters` iterations; *ops_per_sec = operations/s (ops per iteration per lane: 4,2,2,1,2,1).
extern "C" int32_t bbp_ubench(bbp_ctx* ctx, int32_t kind, uint32_t blocks, uint32_t iters, double* ops_per_sec) {
    if (!ctx || !ops_per_sec || kind < 0 || kind > 5 || blocks == 0) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) return bbp_ubench(ctx->members[0], kind, blocks, iters, ops_per_sec);
    return api_guard(ctx, [&]() -> int32_t {
    BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    BBP_HIP_TRY(ctx, hipDeviceSynchronize());  // ctx->vl[0].misc is the verifier's scratch: nothing of this context may still be using it
    int32_t rc = dev_reserve(ctx, ctx->vl[0].misc, (size_t)blocks * 256 * 4);
    if (rc) return rc;
    hipEvent_t a, b;
    BBP_HIP_TRY(ctx, hipEventCreate(&a));
    BBP_HIP_TRY(ctx, hipEventCreate(&b));
    const char* lds_env = getenv("BBP_UBENCH_LDS");
    const unsigned lds = lds_env ? (unsigned)atoi(lds_env) : 0u;
    if (lds > 64 * 1024) BBP_HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_ubench, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_ubench, dim3(blocks), dim3(256), lds, ctx->stream, kind, iters / 8 + 1, (u32*)ctx->vl[0].misc.p);  // warm-up
    BBP_HIP_TRY(ctx, hipEventRecord(a, ctx->stream));
    hipLaunchKernelGGL(k_ubench, dim3(blocks), dim3(256), lds, ctx->stream, kind, iters, (u32*)ctx->vl[0].misc.p);
    BBP_HIP_TRY(ctx, hipEventRecord(b, ctx->stream));
    BBP_HIP_TRY(ctx, hipEventSynchronize(b));
    float ms = 0;
    BBP_HIP_TRY(ctx, hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (getenv("BBP_UBENCH_CLK")) {
        unsigned long long c[2] = {0, 1};
        (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_ubench_clk), sizeof c);
        fprintf(stderr, "[ubench kind %d] shader clock during the loop: %.0f MHz\n", (int)kind, 100.0 * (double)c[0] / (double)c[1]);
    }
    const double per_iter[6] = {4, 2, 2, 1, 2, 1};
    *ops_per_sec = per_iter[kind] * (double)blocks * 256.0 * (double)iters / (ms * 1e-3);
    return BBP_OK;
    });
}
