// Experiment / self-check (GPU box): the one-half-per-lane Keccak-f (csrc/keccak_wave.h) against the plain 25-word permutation on the
// host, and its time per permutation on a lone wavefront (what the prover's draw chain pays).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_keccak_wave.hip -o gpurun_out/exp_keccak_wave && gpurun_out/exp_keccak_wave
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../dusk_blindbidproof_amd/csrc/keccak_wave.h"
using bbp::u32;
using bbp::u64;

__global__ void k_chain(u64* st, u32 perms) {
    const u32 L = threadIdx.x & 63u;
    const bbp::kw_lane c = bbp::kw_setup(L);
    const bbp::kw_iota k = bbp::kw_iota_setup(L);
    u32 a = c.live ? bbp::kw_half(st[c.word], c.half) : 0u;
    for (u32 i = 0; i < perms; i++) a = bbp::kw_keccak_f(a, c, k);
    const auto q = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    if (c.live && c.lower) st[c.word] = bbp::kw_join(a, q[1]);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

int main() {
    u64 h[25], ref[25];
    for (int i = 0; i < 25; i++) h[i] = ref[i] = 0x9E3779B97F4A7C15ull * (u64)(i + 1) ^ ((u64)i << 57) ^ 0x0123456789abcdefull;
    // interleave helpers first
    for (int i = 0; i < 25; i++)
        if (bbp::kw_join(bbp::kw_half(h[i], 0), bbp::kw_half(h[i], 1)) != h[i]) { printf("FAIL interleave round trip\n"); return 1; }
    u64* d;
    CK(hipMalloc(&d, sizeof h));
    int bad = 0;
    for (u32 perms : {1u, 2u, 7u}) {
        memcpy(ref, h, sizeof h);
        for (u32 i = 0; i < perms; i++) bbp::keccak_f1600(ref);
        CK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, perms);
        u64 out[25];
        CK(hipMemcpy(out, d, sizeof out, hipMemcpyDeviceToHost));
        int diff = 0;
        for (int i = 0; i < 25; i++) diff += out[i] != ref[i];
        printf("perms %u: %d of 25 words differ%s\n", perms, diff, diff ? "  FAIL" : "");
        if (diff && perms == 1)
            for (int i = 0; i < 25; i++) printf("  w%02d got %016llx want %016llx\n", i, (unsigned long long)out[i], (unsigned long long)ref[i]);
        bad += diff;
    }
    if (bad) return 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const u32 P = 30000;
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, 100u);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, P);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("one wavefront: %.3f us per permutation (%u permutations in %.2f ms); the draw chain of one N = 8 proof = %.2f ms\n", ms * 1e3 / P, P, ms,
           ms / P * 2936);
    return 0;
}
