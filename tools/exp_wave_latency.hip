// Experiment (GPU box): what a LONE wavefront pays per instruction on a dependent chain -- the prices the one-wavefront Keccak
// (csrc/keccak_wave.h) is made of.  Each kernel runs `iters` x 32 copies of one dependent step; printed: ns per step and shader
// cycles at the clock the box reports.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_wave_latency.hip -o /tmp/exp_wave_latency && /tmp/exp_wave_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

#define KERNEL(name, STEP)                                                       \
    __global__ void name(u32* out, u32 iters, u32 seed) {                        \
        u32 a = seed ^ threadIdx.x, b = seed * 3u + threadIdx.x, c = 4u * ((threadIdx.x * 7u + 3u) & 63u), d = b ^ 0x55u; \
        for (u32 i = 0; i < iters; i++) { asm volatile(REP32(STEP) : "+v"(a), "+v"(b), "+v"(d) : "v"(c) : "vcc"); }      \
        if ((a ^ b ^ d) == 0x12345u) out[0] = a;                                 \
    }

KERNEL(k_xor_dep, "v_xor_b32 %0, %0, %1\n")
KERNEL(k_xor_2chains, "v_xor_b32 %0, %0, %3\n v_xor_b32 %1, %1, %3\n")
KERNEL(k_xor_3chains, "v_xor_b32 %0, %0, %3\n v_xor_b32 %1, %1, %3\n v_xor_b32 %2, %2, %3\n")
KERNEL(k_alignbit_dep, "v_alignbit_b32 %0, %0, %0, %3\n")
KERNEL(k_bitop3_dep, "v_bitop3_b32 %0, %0, %1, %3 bitop3:0x96\n")
KERNEL(k_nop1, "s_nop 1\n")
KERNEL(k_nop0, "s_nop 0\n")
KERNEL(k_xor_nop1, "v_xor_b32 %0, %0, %1\n s_nop 1\n")
KERNEL(k_dpp_dep_nop1, "s_nop 1\n v_xor_b32_dpp %0, %0, %0 row_shl:5 row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
KERNEL(k_dpp_indep, "v_xor_b32_dpp %0, %1, %0 row_shl:5 row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
KERNEL(k_dpp_src_other, "v_mov_b32_dpp %0, %1 row_shl:5 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %2, %1 row_shr:5 row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
KERNEL(k_swap32_dep, "v_mov_b32 %1, %0\n s_nop 1\n v_permlane32_swap_b32 %0, %1\n v_xor_b32 %0, %0, %1\n")
KERNEL(k_swap16_dep, "v_mov_b32 %1, %0\n s_nop 1\n v_permlane16_swap_b32 %0, %1\n v_xor_b32 %0, %0, %1\n")
KERNEL(k_bperm_dep, "ds_bpermute_b32 %0, %3, %0\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_bperm3_dep, "ds_bpermute_b32 %1, %3, %0\n ds_bpermute_b32 %2, %3, %0\n ds_bpermute_b32 %0, %3, %0\n s_waitcnt lgkmcnt(0)\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n")
KERNEL(k_readlane_dep, "v_readlane_b32 s20, %0, 5\n s_nop 3\n v_xor_b32 %0, s20, %0\n")

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

template <typename K> static int run(const char* name, K k, int instr_per_step, u32* d, double ghz) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const u32 iters = 20000;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 100u, 7u);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, iters, 7u);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / (32.0 * iters);
    printf("%-18s %7.2f ns per step (%d instructions) = %6.1f cycles\n", name, ns, instr_per_step, ns * ghz);
    return 0;
}

int main() {
    u32* d;
    CK(hipMalloc(&d, 64));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate / 1e6;
    printf("device clock %.2f GHz\n", ghz);
#define RUN(k, n) if (run(#k, k, n, d, ghz)) return 2;
    RUN(k_xor_dep, 1) RUN(k_xor_2chains, 2) RUN(k_xor_3chains, 3) RUN(k_alignbit_dep, 1) RUN(k_bitop3_dep, 1) RUN(k_nop1, 1) RUN(k_nop0, 1)
    RUN(k_xor_nop1, 2) RUN(k_dpp_dep_nop1, 2) RUN(k_dpp_indep, 1) RUN(k_dpp_src_other, 2) RUN(k_swap32_dep, 4) RUN(k_swap16_dep, 4)
    RUN(k_bperm_dep, 2) RUN(k_bperm3_dep, 5) RUN(k_readlane_dep, 3)
    return 0;
}
