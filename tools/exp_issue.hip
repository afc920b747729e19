// Experiment (GPU box): what does ONE wave-instruction of each kind the field arithmetic is made of cost a SIMD, in shader cycles?
// The accumulate loop is VALU-bound by the counters (profiles/r04_msm_acc_issue_breakdown.*): the way to make it faster is a cheaper
// instruction mix, and that needs prices.  For every candidate instruction: a loop of UNROLL copies on 8 independent register chains
// (so neither dependency latency nor the loop overhead shows), timed with s_memtime, one wavefront per SIMD and two (what k_msm_acc
// runs at).  Printed: cycles per wave-instruction per SIMD (= wave cycles / instructions / waves on the SIMD ... the second column is
// what two co-resident waves pay together).
//   hipcc --offload-arch=gfx950 -O3 tools/exp_issue.hip -o gpurun_out/exp_issue && gpurun_out/exp_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "../dusk_blindbidproof_amd/csrc/point.h"
using bbp::u32;
typedef unsigned long long u64;

#define REP8(x) x x x x x x x x
// 8 chains x 8 = 64 instructions per macro use
#define CHAINS(OP)                                                                                                         \
    REP8(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7))

#define K_BEGIN(name)                                                                                                      \
    __global__ __launch_bounds__(1024) void name(u32 iters, u64* out, u32 seed) {                                           \
        u32 a[8], b[8];                                                                                                    \
        u64 w[8];                                                                                                          \
        for (int i = 0; i < 8; i++) {                                                                                      \
            a[i] = seed * (threadIdx.x + 1) + i * 977u;                                                                    \
            b[i] = seed ^ (threadIdx.x * 2654435761u + i);                                                                 \
            w[i] = ((u64)a[i] << 32) | b[i];                                                                               \
        }                                                                                                                  \
        const u64 t0 = __builtin_amdgcn_s_memtime();                                                                       \
        for (u32 it = 0; it < iters; it++) {
#define K_END                                                                                                              \
        }                                                                                                                  \
        __builtin_amdgcn_s_waitcnt(0);                                                                                     \
        const u64 t1 = __builtin_amdgcn_s_memtime();                                                                       \
        u32 s = 0;                                                                                                         \
        for (int i = 0; i < 8; i++) s ^= a[i] ^ b[i] ^ (u32)w[i] ^ (u32)(w[i] >> 32);                                      \
        if (s == 0x12345678u) out[1] = s;                                                                                  \
        if ((threadIdx.x & 63) == 0) out[2 + (threadIdx.x >> 6)] = t1 - t0;                                                \
    }

#define OP_MAD_I64_I32(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
#define OP_MAD_U64_U32(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
#define OP_MUL_LO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP_MUL_HI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP_MUL_HI_I(i) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP_MUL_U24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP_MAD_U24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define OP_MAD_I32_I24(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define OP_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP_LSHL_ADD(i) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP_BFE(i) asm volatile("v_bfe_i32 %0, %0, 3, 26" : "+v"(a[i]));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 26" : "+v"(a[i]) : "v"(b[i]));
#define OP_LSHL_ADD_U64(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(w[i]));
#define OP_ASHR_I64(i) asm volatile("v_ashrrev_i64 %0, 26, %0" : "+v"(w[i]));
#define OP_LSHR_B64(i) asm volatile("v_lshrrev_b64 %0, 26, %0" : "+v"(w[i]));
#define OP_MOV_B64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(w[i]) : "v"(w[(i + 1) & 7]));
#define OP_ADDC_PAIR(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(a[(i + 1) & 7]) : "vcc");
#define OP_PK_ADD_U16(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP_PK_MUL_LO_U16(i) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP_PK_MAD_U16(i) asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define OP_DOT2_U32_U16(i) asm volatile("v_dot2_u32_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define OP_DOT4_U32_U8(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define OP_FMA_F64(i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(w[i]));
#define OP_FMA_F32(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define OP_PK_FMA_F32(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[i]));
#define OP_MUL_F64(i) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(w[i]));
#define OP_CVT_F64_U32(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(w[i]) : "v"(a[i]));
#define OP_MAD_MIX(i) OP_MAD_I64_I32(i) OP_ADD(i)

#define DEF(name, OP) K_BEGIN(name) CHAINS(OP) K_END
DEF(k_mad_i64_i32, OP_MAD_I64_I32)
DEF(k_mad_u64_u32, OP_MAD_U64_U32)
DEF(k_mul_lo, OP_MUL_LO)
DEF(k_mul_hi, OP_MUL_HI)
DEF(k_mul_hi_i, OP_MUL_HI_I)
DEF(k_mul_u24, OP_MUL_U24)
DEF(k_mad_u24, OP_MAD_U24)
DEF(k_mad_i32_i24, OP_MAD_I32_I24)
DEF(k_add, OP_ADD)
DEF(k_add3, OP_ADD3)
DEF(k_lshl_add, OP_LSHL_ADD)
DEF(k_and, OP_AND)
DEF(k_bfe, OP_BFE)
DEF(k_cndmask, OP_CNDMASK)
DEF(k_mov, OP_MOV)
DEF(k_alignbit, OP_ALIGNBIT)
DEF(k_lshl_add_u64, OP_LSHL_ADD_U64)
DEF(k_ashr_i64, OP_ASHR_I64)
DEF(k_lshr_b64, OP_LSHR_B64)
DEF(k_mov_b64, OP_MOV_B64)
DEF(k_addc_pair, OP_ADDC_PAIR)
DEF(k_pk_add_u16, OP_PK_ADD_U16)
DEF(k_pk_mul_lo_u16, OP_PK_MUL_LO_U16)
DEF(k_pk_mad_u16, OP_PK_MAD_U16)
DEF(k_dot2_u32_u16, OP_DOT2_U32_U16)
DEF(k_dot4_u32_u8, OP_DOT4_U32_U8)
DEF(k_fma_f64, OP_FMA_F64)
DEF(k_fma_f32, OP_FMA_F32)
DEF(k_pk_fma_f32, OP_PK_FMA_F32)
DEF(k_mul_f64, OP_MUL_F64)
DEF(k_cvt_f64_u32, OP_CVT_F64_U32)
DEF(k_mad_then_add, OP_MAD_MIX)

// the real thing: field.h's multiplication (two dependent chains per lane, as in a mixed addition's independent products) and the
// real mixed addition against a row the compiler cannot see through -- cycles per OPERATION per SIMD against waves per SIMD
__global__ __launch_bounds__(1024) void k_fe_mul(u32 iters, u64* out, u32 seed) {
    using bbp::fe; using bbp::fe_mul; using bbp::i32;
    fe a = bbp::fe_d(), b = bbp::fe_sqrt_m1(), c = bbp::fe_d2(), d = bbp::fe_sqrt_ad_minus_one();
    a.v[0] ^= (i32)(threadIdx.x & 0xffff) ^ (i32)(seed & 0xff);
    c.v[1] ^= (i32)(threadIdx.x & 0xffff);
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (u32 it = 0; it < iters; it++) {
        a = fe_mul(a, b);
        c = fe_mul(c, d);
        b = fe_mul(b, a);
        d = fe_mul(d, c);
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if ((u32)(a.v[0] ^ b.v[3] ^ c.v[5] ^ d.v[7]) == 0x12345678u) out[1] = 1;
    if ((threadIdx.x & 63) == 0) out[2 + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ __launch_bounds__(1024) void k_ge_madd(u32 iters, u64* out, u32 seed) {
    using bbp::ge; using bbp::ge_niels; using bbp::i32;
    ge p = bbp::ge_basepoint();
    p.X.v[0] ^= (i32)(threadIdx.x & 0xffff) ^ (i32)(seed & 0xff);
    ge_niels n;
    n.ypx = bbp::fe_d();
    n.ymx = bbp::fe_d2();
    n.xy2d = bbp::fe_sqrt_m1();
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (u32 it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 10; k++) {
            asm volatile("" : "+v"(n.ypx.v[k]));
            asm volatile("" : "+v"(n.ymx.v[k]));
            asm volatile("" : "+v"(n.xy2d.v[k]));
        }
        p = bbp::ge_madd(p, n);
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if ((u32)(p.X.v[0] ^ p.T.v[2]) == 0x12345678u) out[1] = 1;
    if ((threadIdx.x & 63) == 0) out[2 + (threadIdx.x >> 6)] = t1 - t0;
}

struct Entry {
    const char* name;
    void (*fn)(u32, u64*, u32);
    int per_macro;  // instructions per OP
};

int main() {
    const Entry tab[] = {
        {"v_mad_i64_i32", k_mad_i64_i32, 1}, {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_u32", k_mul_hi, 1},
        {"v_mul_hi_i32", k_mul_hi_i, 1}, {"v_mul_u32_u24", k_mul_u24, 1}, {"v_mad_u32_u24", k_mad_u24, 1}, {"v_mad_i32_i24", k_mad_i32_i24, 1},
        {"v_add_u32", k_add, 1}, {"v_add3_u32", k_add3, 1}, {"v_lshl_add_u32", k_lshl_add, 1}, {"v_and_b32", k_and, 1}, {"v_bfe_i32", k_bfe, 1},
        {"v_cndmask_b32", k_cndmask, 1}, {"v_mov_b32", k_mov, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_lshl_add_u64", k_lshl_add_u64, 1},
        {"v_ashrrev_i64", k_ashr_i64, 1}, {"v_lshrrev_b64", k_lshr_b64, 1}, {"v_mov_b64", k_mov_b64, 1}, {"v_add_co+v_addc_co (pair)", k_addc_pair, 2},
        {"v_pk_add_u16", k_pk_add_u16, 1}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, 1}, {"v_pk_mad_u16", k_pk_mad_u16, 1}, {"v_dot2_u32_u16", k_dot2_u32_u16, 1},
        {"v_dot4_u32_u8", k_dot4_u32_u8, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_fma_f32", k_fma_f32, 1}, {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_mul_f64", k_mul_f64, 1},
        {"v_cvt_f64_u32", k_cvt_f64_u32, 1}, {"v_mad_i64_i32 ; v_add_u32 (alternating)", k_mad_then_add, 2},
        {"fe_mul (cycles per multiplication / 256)", k_fe_mul, -4}, {"ge_madd, fresh row (cycles per addition / 256)", k_ge_madd, -1},
    };
    u64* d;
    if (hipMalloc(&d, 64 * sizeof(u64)) != hipSuccess) return 1;
    const u32 iters = 2000;
    printf("%-44s %9s %9s %9s %9s   (cycles per wave-instruction per SIMD at 1 / 2 / 3 / 4 waves per SIMD)\n", "instruction", "1", "2", "3", "4");
    for (const Entry& e : tab) {
        double res[4];
        for (int wps = 1; wps <= 4; wps++) {
            const int threads = 256 * wps;  // one workgroup on one CU: 4 wavefronts per wave-per-SIMD
            u64 h[64] = {0};
            (void)hipMemset(d, 0, sizeof h);
            hipLaunchKernelGGL(e.fn, dim3(1), dim3(threads), 0, 0, iters / 10, d, 12345u);  // warm-up (clocks, instruction cache)
            hipLaunchKernelGGL(e.fn, dim3(1), dim3(threads), 0, 0, iters, d, 12345u);
            if (hipDeviceSynchronize() != hipSuccess) {
                printf("%s: launch failed\n", e.name);
                return 1;
            }
            (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            double sum = 0;
            const int waves = threads / 64;
            for (int w = 0; w < waves; w++) sum += (double)h[2 + w];
            const double per_wave = sum / waves;                       // cycles of one wave's loop
            // wave-instructions one wave issued; the two real-code kernels (per_macro < 0) run |per_macro| operations per loop trip and are
            // reported per operation, scaled by 1 / 256 to sit in the same column format
            const double insts = e.per_macro > 0 ? (double)iters * 64.0 * e.per_macro : (double)iters * (double)(-e.per_macro) * 256.0;
            res[wps - 1] = per_wave / insts / wps;                     // per SIMD: the co-resident waves share it
        }
        printf("%-44s %9.2f %9.2f %9.2f %9.2f\n", e.name, res[0], res[1], res[2], res[3]);
    }
    (void)hipFree(d);
    return 0;
}
