// Batched R1CS prover for the blind-bid circuit on gfx950: replaces Proof::prove (src/blindbid/proof.rs:36-91) and the
// bulletproofs machinery under it (Prover::{new,commit,multiply,constrain,prove}, InnerProductProof::create;
// SURVEY.md App. A.4-A.6) for a whole batch of independent bids.
//
// Pipeline (every stage is a device kernel over the batch; the host only sequences launches):
//   k_witness_head / k_open_serial: constants + committed values; one lane per proof interprets the gadget program   (K3)
//                  (batches on the cooperative draw chain: k_open_bulk50 -- one wavefront per sponge, keccak_wave.h -- with the
//                  gates written from the gadget wiring beside it, witness_gates_native_lane)
//   k_commit       V_i = v_i B + vb_i B~ through the radix-16 comb of the two Pedersen bases                (K2)
//   k_open_serial  Merlin: "V" x m, "m"; TranscriptRng keyed with the blindings; draws i~,o~,s~,s_L,s_R     (K7)
//   MSM x3         A_I1, A_O1, S1 (msm.hip: k_msm_sort + k_msm_acc)                                            (K1)
//   k_tr_yz        Merlin: commitments, 1-phase dom-sep, identity A_I2/A_O2/S2, challenges y, z
//   k_powers       z^k, y^k, y^-k by chunked square-and-multiply
//   k_flatten      wL, wR, wO, wV = sparse gather of z powers (circuit structure is shared by the batch)   (K4)
//   k_poly         l1, r0, r1, r3 and the six t coefficients, block reduction in LDS                        (K5)
//   k_tr_tblind / k_commit / k_tr_ux   T_1,3,4,5,6, challenges u, x, t_x, t_x~, e~, challenge w
//   k_lrvec        l(x), r(x) and the per-generator factor vectors g[k], h[k]
//   rounds 1-6:  { k_ipa_challenge, k_ipa_round, MSM (2 per proof), k_encode }                               (K6)
//   round 7:     generator fold (MSM kernels <1>), k_tail_tables; rounds 7-11: { k_tail_lr, k_encode }
//   k_ipa_final, k_assemble
//
// The inner-product argument never folds the generator vectors point by point.  Round j's L and R (j <= 6) are multiscalar multiplications
// over the ORIGINAL resident generators with scalars a[i] * g[k] / b[i] * h[k], where g[k], h[k] accumulate the
// challenge products u_r^(+-1) that the reference applies by folding G and H (A.6).  The group elements are equal,
// encodings are canonical, hence identical bytes -- and every MSM of the prover is a fixed-base MSM over one table.
// the transcript kernels of small launches run one transcript per WAVEFRONT and spread its permutations over the lanes (keccak.h strobe_run_f)
#define BBP_KECCAK_WAVE 1
#include <string.h>

#include "batch.h"
#include "hosthash.h"
#include "keccak_wave.h"
#include "witness.h"

namespace bbp {

// Heavy-stage kernels other than the MSM accumulation are short and mostly latency-bound; when they share a SIMD with another
// slice's two fat (older, issue-bound) accumulation waves they would crawl.  Raised wave priority lets them take the issue
// slots they need -- a few percent of the SIMD -- and get out of the way.
#define BBP_THIN_PRIO() __builtin_amdgcn_s_setprio(3)
// one-lane-per-item kernels launched with 64-thread workgroups: say so, or the compiler assumes 1024-thread workgroups, caps
// them at 128 VGPRs and spills (k_tr_ux 74 registers, k_vtranscript 64, k_powers 34)
#ifndef BBP_LANE_KERNEL
#define BBP_LANE_KERNEL __launch_bounds__(64)
#endif


// ---------------------------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sc_to_bytes32(uint8_t* o, const sc& s) { sc_tobytes(o, s); }
__device__ __forceinline__ void words_to_bytes32(uint8_t* o, const u32* w) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        o[4 * i] = (uint8_t)w[i];
        o[4 * i + 1] = (uint8_t)(w[i] >> 8);
        o[4 * i + 2] = (uint8_t)(w[i] >> 16);
        o[4 * i + 3] = (uint8_t)(w[i] >> 24);
    }
}
__device__ __forceinline__ void bytes_to_words(u32* w, const uint8_t* b, int nwords) {
    for (int i = 0; i < nwords; i++) w[i] = (u32)b[4 * i] | ((u32)b[4 * i + 1] << 8) | ((u32)b[4 * i + 2] << 16) | ((u32)b[4 * i + 3] << 24);
}

#define LBL(s) reinterpret_cast<const uint8_t*>(s), (u32)(sizeof(s) - 1)

__device__ void tr_append_words(merlin_transcript& t, const uint8_t* label, u32 llen, const u32* w8) {
    uint8_t b[32];
    words_to_bytes32(b, w8);
    merlin_append(t, label, llen, b, 32);
}
__device__ void tr_append_sc(merlin_transcript& t, const uint8_t* label, u32 llen, const sc& s) { tr_append_words(t, label, llen, s.v); }
__device__ sc tr_challenge_sc(merlin_transcript& t, const uint8_t* label, u32 llen) {
    uint8_t b[64];
    merlin_challenge(t, label, llen, b, 64);
    u32 w[16];
    bytes_to_words(w, b, 16);
    return sc_from_wide(w);
}
__device__ sc rng_scalar(merlin_transcript& r) {
    uint8_t b[64];
    merlin_rng_fill(r, b, 64);
    u32 w[16];
    bytes_to_words(w, b, 16);
    return sc_from_wide(w);
}

// x^e by square-and-multiply (e < 2^16)
__device__ sc sc_pow_small(const sc& x, u32 e) {
    sc xm = sc_to_mont(x), acc = sc_r();
    for (int i = 15; i >= 0; i--) {
        acc = sc_montmul(acc, acc);
        if ((e >> i) & 1u) acc = sc_montmul(acc, xm);
    }
    return sc_from_mont(acc);
}

// ---------------------------------------------------------------------------------------------------------------
// K3: witness
// ---------------------------------------------------------------------------------------------------------------
// in_raw per proof: 7 scalars (d,k,y,y_inv,q,z_img,seed) || N items || toggle(u64)
// head: constants and committed values of every proof (cheap, what the V commitments need)
__global__ void k_witness_head(u32 B, u32 n_items, u32 n_cst, const u8* __restrict__ in_raw, sc* __restrict__ cst_all, sc* __restrict__ v_all) {
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    const size_t in_stride = 7 * 32 + (size_t)n_items * 32 + 8;
    const u32* in = reinterpret_cast<const u32*>(in_raw + in_stride * p);
    sc* cst = cst_all + (size_t)p * n_cst;
    const u32 m = 4 + n_items;
    sc s7[7];
    for (int i = 0; i < 7; i++) s7[i] = sc_reduce256(in + 8 * i);
    st_sc(&cst[circuit::CST_ONE], sc_one());
    st_sc(&cst[circuit::CST_ZERO], sc_zero());
    st_sc(&cst[circuit::CST_SEED], s7[6]);
    st_sc(&cst[circuit::CST_ZIMG], s7[5]);
    st_sc(&cst[circuit::CST_Q], s7[4]);
    for (u32 i = 0; i < n_items; i++) st_sc(&cst[circuit::CST_ITEM0 + i], sc_from_bits(in + 56 + 8 * i));  // bid.rs:27
    sc* v = v_all + (size_t)p * m;
    // `x as u64 == toggle` (proof.rs:63): the whole u64 is compared -- toggle = 2^32 + 3 sets no bit
    const u64 toggle = (u64)in[56 + 8 * n_items] | ((u64)in[56 + 8 * n_items + 1] << 32);
    st_sc(&v[0], s7[0]);
    st_sc(&v[1], s7[1]);
    st_sc(&v[2], s7[2]);  // y: committed but never wired into the gadget (proof.rs:55, 76-78)
    st_sc(&v[3], s7[3]);
    for (u32 i = 0; i < n_items; i++) st_sc(&v[4 + i], (u64)i == toggle ? sc_one() : sc_zero());
}

// gates: one lane per proof, interpreting the compiled gadget program or writing the gadget wiring out (witness.h: both forms, host + device)
__device__ void witness_gates_lane(u32 p, u32 m, u32 n_mul, u32 n_cst, const u32* w_terms, const u32* w_loff,
                                   const u32* w_roff, const sc* __restrict__ cst_all, const sc* __restrict__ v_all,
                                   sc* __restrict__ ai1_all, sc* __restrict__ ao1_all) {
    sc* aL = ai1_all + (size_t)p * (1 + 2 * n_mul) + 1;
    witness_gates_interpret(n_mul, w_terms, w_loff, w_roff, cst_all + (size_t)p * n_cst, v_all + (size_t)p * m, aL, aL + n_mul,
                            ao1_all + (size_t)p * (1 + n_mul) + 1);
}
// The interpreter fetches every operand it stored one multiplier earlier back from memory, which is its whole running time (5.9 ms
// per proof against 1.1 for the written-out wiring); BBP_WITNESS_NATIVE=0 keeps it, and the launches with one lane per proof
// for the draw chain always use it.
__device__ void witness_gates_native_lane(u32 p, u32 m, u32 n_mul, u32 n_cst, const sc* __restrict__ cst_all, const sc* __restrict__ v_all,
                                          sc* __restrict__ ai1_all, sc* __restrict__ ao1_all) {
    sc* aL = ai1_all + (size_t)p * (1 + 2 * n_mul) + 1;
    witness_gates_native(m - 4, cst_all + (size_t)p * n_cst, v_all + (size_t)p * m, aL, aL + n_mul, ao1_all + (size_t)p * (1 + n_mul) + 1);
}

// The witness blocks of the cooperative opening launches (one lane per proof).  The interpreter's time is its dependent loads --
// offsets -> term word -> value, 5200 terms, and every value a MiMC round needs was stored by the multiplier before it: 6.6 ms per
// proof, MORE than the draw chain beside it since keccak_wave.h.  The program (offsets + term words, the same for every proof of
// the launch) is therefore staged ONCE per workgroup in the dynamic LDS these launches reserve anyway (the CU is theirs alone):
// 6.6 -> 5.9 ms, level with the chain.  (Fetching a multiplier's eight term words and values together instead: 7.5 ms -- the
// store -> load round trips of the values are the chain, not the number of loads; a value ring in LDS is what would cut it.)
// lds_bytes: what the launch was given; a program that does not fit (or BBP_SERIAL_LDS=0) is read from global memory as before.
__device__ void witness_gates_block(u32 first_proof, u32 B, u32 m, u32 n_mul, u32 n_cst, const u32* __restrict__ w_terms, const u32* __restrict__ w_loff,
                                    const u32* __restrict__ w_roff, const sc* __restrict__ cst_all, const sc* __restrict__ v_all,
                                    sc* __restrict__ ai1_all, sc* __restrict__ ao1_all, u32 lds_bytes, u32 native) {
    if (native) {
        const u32 p = first_proof + threadIdx.x;
        if (p < B) witness_gates_native_lane(p, m, n_mul, n_cst, cst_all, v_all, ai1_all, ao1_all);
        return;
    }
    extern __shared__ u32 prog[];
    const u32 n_terms = w_loff[n_mul];
    const bool staged = (size_t)(2 * n_mul + 1 + n_terms) * 4 <= lds_bytes;  // uniform over the launch
    const u32 *lo = w_loff, *ro = w_roff, *tw = w_terms;
    if (staged) {
        for (u32 i = threadIdx.x; i <= n_mul; i += blockDim.x) prog[i] = w_loff[i];
        for (u32 i = threadIdx.x; i < n_mul; i += blockDim.x) prog[n_mul + 1 + i] = w_roff[i];
        for (u32 i = threadIdx.x; i < n_terms; i += blockDim.x) prog[2 * n_mul + 1 + i] = w_terms[i];
        __syncthreads();
        lo = prog;
        ro = prog + n_mul + 1;
        tw = prog + 2 * n_mul + 1;
    }
    const u32 p = first_proof + threadIdx.x;
    if (p < B) witness_gates_lane(p, m, n_mul, n_cst, tw, lo, ro, cst_all, v_all, ai1_all, ao1_all);
}

// MiMC constants into every proof's constant table (one lane per (proof, round))
__global__ void k_fill_mimc(u32 B, u32 n_cst, const sc* __restrict__ mimc, sc* __restrict__ cst_all) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * BBP_MIMC_ROUNDS) return;
    u32 p = t / BBP_MIMC_ROUNDS, i = t % BBP_MIMC_ROUNDS;
    st_sc(&cst_all[(size_t)p * n_cst + circuit::CST_MIMC0 + i], ld_sc(&mimc[i]));
}

// ---------------------------------------------------------------------------------------------------------------
// K2: Pedersen commitments through the radix-16 comb (64 signed digits per scalar, 8 cached multiples each)
// ---------------------------------------------------------------------------------------------------------------
__device__ ge comb_mul_add(ge acc, const niels_packed* __restrict__ comb_base, const sc& s) {
    u32 carry = 0;
    for (int j = 0; j < 64; j++) {
        u32 d = ((s.v[j >> 3] >> (4 * (j & 7))) & 15u) + carry;
        carry = d > 8u;
        u32 mag = carry ? 16u - d : d;
        if (mag) {
            const uint4* q = reinterpret_cast<const uint4*>(comb_base + (size_t)j * 8 + (mag - 1));
            uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5];
            ge_niels n;
            n.ypx = BBP_FE_LIT(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w);
            n.ymx = BBP_FE_LIT(q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w);
            n.xy2d = BBP_FE_LIT(q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w);
            if (carry) {
                fe t = n.ypx;
                n.ypx = n.ymx;
                n.ymx = t;
                n.xy2d = fe_neg(n.xy2d);
            }
            acc = ge_madd(acc, n);
        }
    }
    return acc;  // canonical scalars are < 2^253: the top digit never carries out
}

// commitment c of proof p: values[p*stride_v + c], blindings[p*stride_b + c] -> out[p*out_stride + c]
__global__ BBP_LANE_KERNEL void k_commit(u32 count, u32 per_proof, const sc* __restrict__ values, const sc* __restrict__ blindings, u32 stride_v,
                         u32 stride_b, const niels_packed* __restrict__ comb, ge* __restrict__ out, u32 out_stride) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    u32 p = t / per_proof, c = t % per_proof;
    sc v = ld_sc(&values[(size_t)p * stride_v + c]);
    sc b = ld_sc(&blindings[(size_t)p * stride_b + c]);
    ge acc = comb_mul_add(ge_identity(), comb, v);              // v * B
    acc = comb_mul_add(acc, comb + 64 * 8, b);                  // + vb * B_blinding
    out[(size_t)p * out_stride + c] = acc;
}

// Small launches: a commitment on COMMIT_L lanes.  One lane walks 2 x 64 comb digits (~120 dependent mixed additions: 380 us for
// ONE proof's twelve commitments, and again for its five T commitments); here lane q of a group takes the digit positions
// j = q (mod COMMIT_L) of both scalars (every lane recodes the whole scalar -- the carries -- which is cheap) and three shuffle
// steps add the partial sums.  The sum is the same group element, its encoding the same bytes.
constexpr int COMMIT_L = 8;
__device__ ge comb_mul_add_part(ge acc, const niels_packed* __restrict__ comb_base, const sc& s, u32 q) {
    // the signed radix-16 digits of the whole scalar first (the carries are a chain): magnitudes as nibbles, signs as a bit mask;
    // then the lane's own eight positions q, q + 8, ... -- every lane of the wavefront adds at the same time
    u32 mags[8];
    u64 negs = 0;
    u32 carry = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        u32 mw = 0;
#pragma unroll
        for (int n = 0; n < 8; n++) {
            const u32 d = ((s.v[w] >> (4 * n)) & 15u) + carry;
            carry = d > 8u;
            mw |= (carry ? 16u - d : d) << (4 * n);
            negs |= (u64)carry << (8 * w + n);
        }
        mags[w] = mw;
    }
#pragma unroll 1
    for (int i = 0; i < 64 / COMMIT_L; i++) {
        u32 mw = mags[0];
#pragma unroll
        for (int w = 1; w < 8; w++) mw = i == w ? mags[w] : mw;
        const u32 j = q + (u32)COMMIT_L * (u32)i, mag = (mw >> (4 * q)) & 15u, neg = (u32)(negs >> j) & 1u;
        if (mag) {
            const uint4* qq = reinterpret_cast<const uint4*>(comb_base + (size_t)j * 8 + (mag - 1));
            uint4 q0 = qq[0], q1 = qq[1], q2 = qq[2], q3 = qq[3], q4 = qq[4], q5 = qq[5];
            ge_niels n;
            n.ypx = BBP_FE_LIT(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w);
            n.ymx = BBP_FE_LIT(q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w);
            n.xy2d = BBP_FE_LIT(q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w);
            if (neg) {
                fe t = n.ypx;
                n.ypx = n.ymx;
                n.ymx = t;
                n.xy2d = fe_neg(n.xy2d);
            }
            acc = ge_madd(acc, n);
        }
    }
    return acc;
}
__device__ __forceinline__ ge commit_group_sum(ge acc) {  // lane 0 of every COMMIT_L-lane group ends up with the group's sum
#pragma unroll 1
    for (int d = COMMIT_L / 2; d >= 1; d >>= 1) {
        ge other;
        const u32* w = reinterpret_cast<const u32*>(&acc);
        u32* o = reinterpret_cast<u32*>(&other);
#pragma unroll
        for (int i = 0; i < GE_WORDS; i++) o[i] = (u32)__shfl_down((int)w[i], d, 64);
        acc = ge_add(acc, other);  // (lanes whose partner lies in the next group add something nobody reads)
    }
    return acc;
}
__global__ BBP_LANE_KERNEL void k_commit_split(u32 count, u32 per_proof, const sc* __restrict__ values, const sc* __restrict__ blindings, u32 stride_v,
                                               u32 stride_b, const niels_packed* __restrict__ comb, ge* __restrict__ out, u32 out_stride) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x, g = t / COMMIT_L, q = t % COMMIT_L;
    const u32 gi = g < count ? g : count - 1;  // whole wavefronts reach the shuffles
    const u32 p = gi / per_proof, c = gi % per_proof;
    const sc v = ld_sc(&values[(size_t)p * stride_v + c]);
    const sc b = ld_sc(&blindings[(size_t)p * stride_b + c]);
    ge acc = comb_mul_add_part(ge_identity(), comb, v, q);
    acc = comb_mul_add_part(acc, comb + 64 * 8, b, q);
    acc = commit_group_sum(acc);
    if (q == 0 && g < count) out[(size_t)p * out_stride + c] = acc;
}

int32_t commit_launch(bbp_ctx* ctx, u32 count, const sc* values, const sc* blindings, u32 stride_v, u32 stride_b, u32 per_proof,
                      ge* out, u32 out_stride, hipStream_t s) {
    if (!count) return BBP_OK;
    ScopedEvent ev(ctx, TAG_COMMIT, s);
    if (count <= (u32)ctx->commit_split_below) {
        hipLaunchKernelGGL(k_commit_split, dim3((count * COMMIT_L + 63) / 64), dim3(64), 0, s, count, per_proof, values, blindings, stride_v, stride_b,
                           ctx->comb, out, out_stride);
        BBP_HIP_TRY(ctx, hipGetLastError());
        return BBP_OK;
    }
    hipLaunchKernelGGL(k_commit, dim3((count + 63) / 64), dim3(64), 0, s, count, per_proof, values, blindings, stride_v, stride_b,
                       ctx->comb, out, out_stride);
    BBP_HIP_TRY(ctx, hipGetLastError());
    return BBP_OK;
}

// encode `per_proof` points per proof from a strided point array into a strided encoding array
// -DBBP_ENCODE_WAVES=3 (experiment): 179 registers + 384 B of scratch instead of 248 + 32 AGPRs + 144 B -- it then fits on a SIMD beside
// two accumulate waves (<= 192).  Measured: see DESIGN.md section 9 "round 3".
#ifdef BBP_ENCODE_WAVES
#define BBP_ENCODE_ATTR __attribute__((amdgpu_waves_per_eu(BBP_ENCODE_WAVES, BBP_ENCODE_WAVES)))
#else
#define BBP_ENCODE_ATTR
#endif
__global__ BBP_LANE_KERNEL BBP_ENCODE_ATTR void k_encode_strided(u32 count, u32 per_proof, const ge* __restrict__ pts, u32 pts_stride, u32* __restrict__ enc,
                                 u32 enc_stride_words, u32 enc_off_words, u32 pts_cstride) {
    BBP_THIN_PRIO();
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    u32 p = t / per_proof, c = t % per_proof;
    u32 w[8];
    ge_encode_words(w, pts[(size_t)p * pts_stride + (size_t)c * pts_cstride]);  // point c of proof p (pts_cstride = 1: a proof's points are contiguous)
    u32* o = enc + (size_t)p * enc_stride_words + enc_off_words + 8 * c;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = w[i];
}

// ---------------------------------------------------------------------------------------------------------------
// K7: transcript stages (one lane per proof)
// ---------------------------------------------------------------------------------------------------------------
// enc layout per proof (words): V[m] | A_I1 A_O1 S1 | T_1 T_3 T_4 T_5 T_6 | (L_j R_j) x 11
__device__ __forceinline__ u32 enc_stride_words(u32 m) { return (m + 8 + 22) * 8; }

__device__ void tr_open_lane(u32 p, u32 m, u32 n1, const merlin_transcript& prefix, const u32* __restrict__ enc, const u8* __restrict__ entropy,
                             const sc* __restrict__ vb_all, u32* __restrict__ raw, merlin_transcript* __restrict__ tr_out,
                             merlin_transcript* __restrict__ rng_out, bool coop, u32 wave) {
    merlin_transcript t = prefix;  // Transcript::new(b"BlindBidProofGadget") + r1cs_domain_sep (A.4)
    t.wave = wave;
    const u32* e = enc + (size_t)p * enc_stride_words(m);
    for (u32 i = 0; i < m; i++) tr_append_words(t, LBL("V"), e + 8 * i);
    merlin_append_u64(t, LBL("m"), (u64)m);
    // TranscriptRng: rekey with every v_blinding in commit order, then 32 bytes of external entropy (A.5 step 2)
    merlin_transcript r = t;
    const sc* vb = vb_all + (size_t)p * m;
    for (u32 i = 0; i < m; i++) {
        uint8_t b[32];
        sc_to_bytes32(b, ld_sc(&vb[i]));
        merlin_rng_rekey(r, LBL("v_blinding"), b, 32);
    }
    const u8* ent = entropy + (size_t)p * (32 * (size_t)m + 32) + 32 * (size_t)m;
    uint8_t seed[32];
    for (int i = 0; i < 32; i++) seed[i] = ent[i];
    merlin_rng_finalize(r, seed);
    // draws, in order: i_blinding1, o_blinding1, s_blinding1, s_L1[n1], s_R1[n1] (A.5 step 3).  Raw 64-byte outputs go
    // to `raw`; the wide reductions mod l run afterwards in k_reduce_draws, off this strictly sequential chain.
    u32* rw = raw + (size_t)p * (3 + 2 * (size_t)n1) * 16;
    {
        uint8_t b[64];
        merlin_rng_fill(r, b, 64);  // first draw leaves the sponge in the steady state the bulk path needs
        bytes_to_words(rw, b, 16);
    }
    // the remaining 2 + 2 n1 draws: BBP_RNG_COOP (default) leaves them to the cooperative kernel (k_open_bulk, 25 lanes per sponge)
    if (!coop) merlin_rng_fill64_bulk(r, 2 + 2 * n1, rw + 16);
    t.wave = r.wave = 0;  // (memory never holds the flag: every kernel sets it for itself)
    tr_out[p] = t;
    rng_out[p] = r;
}

// The two strictly serial jobs of the opening stage in ONE launch: the first `rng_blocks` workgroups run the transcript opening +
// the 2935 sequential rng draws of their proofs, the others interpret the gadget program of theirs.  Neither needs the other
// (the V commitments only need k_witness_head), so the opening stage lasts max(36 ms, 6 ms) instead of their sum.
// (BBP_RNG_COOP=0 path: one lane per proof does everything.  Default path: this kernel is launched with rng lanes only for the
// transcript prefix -- V x m, "m", rng rekeys, first draw -- and k_open_bulk does the bulk of the draws and the witness.)
__global__ void k_open_serial(u32 B, u32 rng_blocks, u32 m, u32 n1, merlin_transcript prefix, const u32* __restrict__ enc,
                              const u8* __restrict__ entropy, const sc* __restrict__ vb_all, u32* __restrict__ raw,
                              merlin_transcript* __restrict__ tr_out, merlin_transcript* __restrict__ rng_out, u32 n_cst,
                              const u32* __restrict__ w_terms, const u32* __restrict__ w_loff, const u32* __restrict__ w_roff,
                              const sc* __restrict__ cst_all, const sc* __restrict__ v_all, sc* __restrict__ ai1_all, sc* __restrict__ ao1_all,
                              u32 coop) {
    if (blockIdx.x < rng_blocks) {
        // coop: 0 = the whole draw chain here; 1 = the transcript prefix and the first draw only (the chain follows in k_open_bulk*);
        // 2 = the same with one proof per WAVEFRONT, every lane in lockstep, the permutations spread over the lanes
        const u32 t = blockIdx.x * blockDim.x + threadIdx.x, p = coop == 2 ? t >> 6 : t;
        if (p < B) tr_open_lane(p, m, n1, prefix, enc, entropy, vb_all, raw, tr_out, rng_out, coop != 0, coop == 2);
    } else {
        const u32 p = (blockIdx.x - rng_blocks) * blockDim.x + threadIdx.x;
        if (p < B) witness_gates_lane(p, m, n1, n_cst, w_terms, w_loff, w_roff, cst_all, v_all, ai1_all, ao1_all);
    }
}

// ---- cooperative Keccak: the TranscriptRng draw chain with 25 lanes per sponge --------------------------------------------------
// A proof's 2 + 2 n1 (= 2934 at N = 8) steady-state draws are strictly sequential -- each is one Keccak-f[1600] of the previous
// state (keccak.h merlin_rng_fill64_bulk) -- so one lane per proof runs 2934 x 24 rounds x ~270 dependent-ish instructions: 36 ms
// whatever the batch size, the floor under every small batch and under a single proof's latency.  Here a sponge is spread over a
// 32-lane group (lane i < 25 holds state word i = x + 5y as a register pair; two groups per wavefront): theta's column parities,
// the neighbours for D, and the fused rho-pi-chi gather are cross-lane reads (ds_bpermute: 18 per round), everything else is
// per-lane 64-bit arithmetic with lane-constant rotation counts -- ~50 instructions per round per wavefront instead of ~270,
// in three short dependent phases.  The bytes are the same (tests compare every record with the oracle).
__device__ __forceinline__ u64 coop_gather64(int addr, u64 v) {
    const u32 lo = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)(u32)v);
    const u32 hi = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)(u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}

struct coop_lane {  // per-lane constants of the cooperative permutation
    int col[4], dm, dp, s0, s1, s2;  // byte addresses (4 * source lane) for ds_bpermute
    u32 rot;                         // rho offset of THIS lane's word
    u64 iota_mask;                   // all ones in lane 0 of the group
};

__device__ __forceinline__ coop_lane coop_setup(u32 lane_in_wave) {
    const u32 l = lane_in_wave & 31u, base = lane_in_wave & 32u;
    const u32 li = l < 25 ? l : 0;  // idle lanes mirror lane 0's wiring; their values are never read by live lanes
    const u32 x = li % 5, y = li / 5;
    const u32 RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    coop_lane c;
#pragma unroll
    for (int k = 0; k < 4; k++) c.col[k] = 4 * (int)(base + x + 5 * ((y + 1 + k) % 5));
    c.dm = 4 * (int)(base + (x + 4) % 5 + 5 * y);
    c.dp = 4 * (int)(base + (x + 1) % 5 + 5 * y);
    // rho-pi-chi fused: destination (X, Y) needs B[X][Y], B[X+1][Y], B[X+2][Y]; B[X][Y] is the rotated word of source lane
    // ((X + 3Y) mod 5) + 5 X   (pi: B[y][2x+3y] = rot(A[x][y]))
    auto src = [&](u32 X) { return 4 * (int)(base + (X + 3 * y) % 5 + 5 * X); };
    c.s0 = src(x);
    c.s1 = src((x + 1) % 5);
    c.s2 = src((x + 2) % 5);
    c.rot = RHO[li];
    c.iota_mask = l == 0 ? ~0ull : 0ull;
    return c;
}

__device__ __forceinline__ u64 coop_keccak_f(u64 a, const coop_lane& c) {
    const u64 RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                        0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                        0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                        0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                        0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
        // theta: C[x] in every lane of column x, then D[x] = C[x-1] ^ rotl(C[x+1], 1)
        const u64 C = a ^ coop_gather64(c.col[0], a) ^ coop_gather64(c.col[1], a) ^ coop_gather64(c.col[2], a) ^ coop_gather64(c.col[3], a);
        const u64 Cm = coop_gather64(c.dm, C), Cp = coop_gather64(c.dp, C);
        a ^= Cm ^ ((Cp << 1) | (Cp >> 63));
        // rho at the source, then pi and chi's two neighbours in one gather phase
        const u64 ro = (a << c.rot) | (a >> ((64u - c.rot) & 63u));
        const u64 b0 = coop_gather64(c.s0, ro), b1 = coop_gather64(c.s1, ro), b2 = coop_gather64(c.s2, ro);
        a = b0 ^ (~b1 & b2);
        a ^= RC[r] & c.iota_mask;  // iota
    }
    return a;
}

// rng lanes: (B proofs) x 32 lanes, two proofs per wavefront; witness lanes (blocks >= rng_blocks): one lane per proof as before
__global__ void k_open_bulk(u32 B, u32 rng_blocks, u32 n1, u32 count, merlin_transcript* __restrict__ rng, u32* __restrict__ raw, u32 m,
                            u32 n_cst, const u32* __restrict__ w_terms, const u32* __restrict__ w_loff, const u32* __restrict__ w_roff,
                            const sc* __restrict__ cst_all, const sc* __restrict__ v_all, sc* __restrict__ ai1_all, sc* __restrict__ ao1_all, u32 lds_bytes, u32 wit_native) {
    if (blockIdx.x >= rng_blocks) {
        witness_gates_block((blockIdx.x - rng_blocks) * blockDim.x, B, m, n1, n_cst, w_terms, w_loff, w_roff, cst_all, v_all, ai1_all, ao1_all, lds_bytes, wit_native);
        return;
    }
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 g = t >> 5, l = t & 31u;
    const bool live = g < B && l < 25;
    const u32 p = g < B ? g : B - 1;
    const coop_lane c = coop_setup(threadIdx.x & 63u);
    merlin_transcript* T = rng + p;
    u64 a = live ? T->st[l] : 0;
    // per-draw constants (keccak.h merlin_rng_fill64_bulk): st[8] ^= .., st[9] ^= .., st[20] ^= ..
    const u64 konst = l == 8 ? 0x0741000000401200ull : l == 9 ? 0x0000000000000447ull : l == 20 ? 0x8000000000000000ull : 0ull;
    u32* out = raw + (size_t)p * (3 + 2 * (size_t)n1) * 16 + 16 + 2 * l;  // draw 0 was written by the prefix kernel
    for (u32 d = 0; d < count; d++) {
        a ^= konst;
        a = coop_keccak_f(a, c);
        if (l < 8) {
            if (live) {
                out[(size_t)d * 16] = (u32)a;
                out[(size_t)d * 16 + 1] = (u32)(a >> 32);
            }
            a = 0;
        }
    }
    if (live) T->st[l] = a;
    if (live && l == 0) T->cur_flags = BBP_FLAG_I | BBP_FLAG_A | BBP_FLAG_C;
}

// ---- cooperative Keccak, second form (round 3): ONE sponge per wavefront, theta without the LDS crossbar ---------------------------
// State word (x, y) lives in lane 8 y + x (x, y < 5: lanes 0..36 with gaps; every other lane holds zero and stays zero), so a
// DPP row (16 lanes) holds two planes.  Column parity: one row_ror:8 XOR joins the two planes of a row, v_permlane16_swap /
// v_permlane32_swap (gfx950) join the four rows -- seven VALU instructions per 32-bit half instead of eight ds_bpermute and a
// wait; theta's two neighbours are DPP row shifts inside the 8-lane group (the wrap from x = 4 to x = 0 through a second,
// bank-masked move).  Only the rho-pi-chi gather still crosses the wavefront through ds_bpermute (6 per round instead of 18):
// one LDS round trip per round instead of three.  Rho is two v_alignbit_b32 with a lane-constant shift after a lane-constant
// word swap.  The bytes are the same as the 25-lane form's and the single-lane chain's (every parity test goes through it).
#ifndef BBP_COOP8_UNROLL
#define BBP_COOP8_UNROLL 24
#endif
struct coop8_lane {
    int s0, s1, s2;      // ds_bpermute byte addresses of B[x][y], B[x+1][y], B[x+2][y]
    u32 sh;              // v_alignbit shift of the rho rotation
    bool swap, x0, first;  // rho: swap the halves first; lane holds column 0; lane holds word (0, 0)
    u32 live;            // all ones in the 25 state lanes
};

__device__ __forceinline__ coop8_lane coop8_setup(u32 L) {
    const u32 x = L & 7u, y = L >> 3;
    const bool live = x < 5 && y < 5;
    const u32 RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    coop8_lane c;
    // pi: B[X][Y] is the rotated word of source (x, y) = ((X + 3 Y) mod 5, X); idle lanes read themselves (zero)
    auto src = [&](u32 X) { return live ? 4 * (int)(8 * X + (X + 3 * y) % 5) : 4 * (int)L; };
    c.s0 = src(x % 5);
    c.s1 = src((x + 1) % 5);
    c.s2 = src((x + 2) % 5);
    const u32 rot = live ? RHO[(x + 5 * y) % 25] : 0, r = rot & 31u;
    c.swap = (rot >= 32) != (r == 0);  // alignbit by 0 is itself a swap of the halves
    c.sh = (32u - r) & 31u;
    c.x0 = x == 0;
    c.first = L == 0;
    c.live = live ? ~0u : 0u;
    return c;
}

#define BBP_DPP_ROW_SHL(n) (0x100 + (n))
#define BBP_DPP_ROW_SHR(n) (0x110 + (n))
#define BBP_DPP_ROW_ROR(n) (0x120 + (n))
// XOR over the five planes of both halves of the state words, in every lane of the column (lane & 7).  The two halves share the
// permlane swaps: after the first one a single register carries the low half's row pairs in even rows and the high half's in odd rows.
__device__ __forceinline__ void coop8_colpar(u32 lo, u32 hi, u32& cl, u32& ch) {
    const u32 tl = lo ^ (u32)__builtin_amdgcn_update_dpp(0, (int)lo, BBP_DPP_ROW_ROR(8), 0xf, 0xf, true);  // the row's two planes
    const u32 th = hi ^ (u32)__builtin_amdgcn_update_dpp(0, (int)hi, BBP_DPP_ROW_ROR(8), 0xf, 0xf, true);
    const auto p = __builtin_amdgcn_permlane16_swap(tl, th, false, false);  // (tl.r0, th.r0, tl.r2, th.r2), (tl.r1, th.r1, tl.r3, th.r3)
    const u32 x = p[0] ^ p[1];                                              // rows: A_l, A_h, B_l, B_h  (A = r0 ^ r1, B = r2 ^ r3)
    const auto q = __builtin_amdgcn_permlane32_swap(x, x, false, false);    // (A_l, A_h, A_l, A_h), (B_l, B_h, B_l, B_h)
    const u32 y = q[0] ^ q[1];                                              // rows: L, H, L, H
    const auto r = __builtin_amdgcn_permlane16_swap(y, y, false, false);    // (L, L, L, L), (H, H, H, H)
    cl = r[0];
    ch = r[1];
}
__device__ __forceinline__ u32 coop8_next(u32 v) {  // column x + 1 (mod 5) of the same plane
    const int a = __builtin_amdgcn_update_dpp(0, (int)v, BBP_DPP_ROW_SHL(1), 0xf, 0xf, true);
    return (u32)__builtin_amdgcn_update_dpp(a, (int)v, BBP_DPP_ROW_SHR(4), 0xf, 0xa, false);  // lanes 4..7 / 12..15 of a row: x = 4 reads x = 0
}
__device__ __forceinline__ u32 coop8_prev(u32 v, bool x0) {  // column x - 1 (mod 5) of the same plane
    const u32 w = (u32)__builtin_amdgcn_update_dpp(0, (int)v, BBP_DPP_ROW_SHL(4), 0xf, 0xf, true);  // x = 0 reads x = 4
    const u32 a = (u32)__builtin_amdgcn_update_dpp(0, (int)v, BBP_DPP_ROW_SHR(1), 0xf, 0xf, true);
    return x0 ? w : a;
}

__device__ __forceinline__ void coop8_keccak_f(u32& lo, u32& hi, const coop8_lane& c) {
    const u64 RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                        0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                        0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                        0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                        0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
#pragma unroll BBP_COOP8_UNROLL
    for (int r = 0; r < 24; r++) {
        // theta
        u32 cl, ch;
        coop8_colpar(lo, hi, cl, ch);
        const u32 ml = coop8_prev(cl, c.x0), mh = coop8_prev(ch, c.x0), pl = coop8_next(cl), ph = coop8_next(ch);
        lo ^= (ml ^ __builtin_amdgcn_alignbit(pl, ph, 31)) & c.live;  // D = C[x-1] ^ rotl(C[x+1], 1); the idle planes must stay zero
        hi ^= (mh ^ __builtin_amdgcn_alignbit(ph, pl, 31)) & c.live;
        // rho at the source
        const u32 a = c.swap ? hi : lo, b = c.swap ? lo : hi;  // (low, high) after the optional rotation by 32
        const u32 rl = __builtin_amdgcn_alignbit(a, b, c.sh), rh = __builtin_amdgcn_alignbit(b, a, c.sh);
        // pi and chi's two neighbours in one gather phase
        const u32 b0l = (u32)__builtin_amdgcn_ds_bpermute(c.s0, (int)rl), b0h = (u32)__builtin_amdgcn_ds_bpermute(c.s0, (int)rh);
        const u32 b1l = (u32)__builtin_amdgcn_ds_bpermute(c.s1, (int)rl), b1h = (u32)__builtin_amdgcn_ds_bpermute(c.s1, (int)rh);
        const u32 b2l = (u32)__builtin_amdgcn_ds_bpermute(c.s2, (int)rl), b2h = (u32)__builtin_amdgcn_ds_bpermute(c.s2, (int)rh);
        // chi as one bit-select: b1 ? b0 : b0 ^ b2; iota in lane 0
        lo = ((b1l & b0l) | (~b1l & (b0l ^ b2l))) ^ (c.first ? (u32)RC[r] : 0u);
        hi = ((b1h & b0h) | (~b1h & (b0h ^ b2h))) ^ (c.first ? (u32)(RC[r] >> 32) : 0u);
    }
}

// rng waves: one proof per wavefront (state lanes 8 y + x); witness blocks as in k_open_bulk
__global__ void k_open_bulk8(u32 B, u32 rng_blocks, u32 n1, u32 count, merlin_transcript* __restrict__ rng, u32* __restrict__ raw, u32 m,
                             u32 n_cst, const u32* __restrict__ w_terms, const u32* __restrict__ w_loff, const u32* __restrict__ w_roff,
                             const sc* __restrict__ cst_all, const sc* __restrict__ v_all, sc* __restrict__ ai1_all, sc* __restrict__ ao1_all, u32 lds_bytes, u32 wit_native) {
    if (blockIdx.x >= rng_blocks) {
        witness_gates_block((blockIdx.x - rng_blocks) * blockDim.x, B, m, n1, n_cst, w_terms, w_loff, w_roff, cst_all, v_all, ai1_all, ao1_all, lds_bytes, wit_native);
        return;
    }
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 g = t >> 6, L = t & 63u;
    if (g >= B) return;  // spare wavefront of the last workgroup (see k_open_bulk50)
    const coop8_lane c = coop8_setup(L);
    const bool live = g < B && c.live != 0;
    const u32 p = g < B ? g : B - 1;
    const u32 word = (L & 7u) + 5 * (L >> 3);  // index of this lane's state word (state lanes only)
    merlin_transcript* T = rng + p;
    const u64 a0 = c.live ? T->st[word % 25] : 0;
    u32 lo = (u32)a0, hi = (u32)(a0 >> 32);
    // per-draw constants (keccak.h merlin_rng_fill64_bulk): st[8] ^= .., st[9] ^= .., st[20] ^= ..
    const u64 konst = !c.live ? 0ull : word == 8 ? 0x0741000000401200ull : word == 9 ? 0x0000000000000447ull : word == 20 ? 0x8000000000000000ull : 0ull;
    const bool rate = c.live && word < 8;
    u32* out = raw + (size_t)p * (3 + 2 * (size_t)n1) * 16 + 16 + 2 * (word & 7u);  // draw 0 was written by the prefix kernel
    for (u32 d = 0; d < count; d++) {
        lo ^= (u32)konst;
        hi ^= (u32)(konst >> 32);
        coop8_keccak_f(lo, hi, c);
        if (rate) {
            if (live) {
                out[(size_t)d * 16] = lo;
                out[(size_t)d * 16 + 1] = hi;
            }
            lo = hi = 0;
        }
    }
    if (live) T->st[word] = ((u64)hi << 32) | lo;
    if (live && c.first) T->cur_flags = BBP_FLAG_I | BBP_FLAG_A | BBP_FLAG_C;
}

// rng waves, third form (keccak_wave.h): one proof per wavefront, one 32-bit half of a state word per lane, bit-interleaved.
// The draws go out as (even bits, odd bits) pairs; k_reduce_draws joins them (interleaved = 1).  Witness blocks as in k_open_bulk.
__global__ void k_open_bulk50(u32 B, u32 rng_blocks, u32 n1, u32 count, merlin_transcript* __restrict__ rng, u32* __restrict__ raw, u32 m,
                              u32 n_cst, const u32* __restrict__ w_terms, const u32* __restrict__ w_loff, const u32* __restrict__ w_roff,
                              const sc* __restrict__ cst_all, const sc* __restrict__ v_all, sc* __restrict__ ai1_all, sc* __restrict__ ao1_all, u32 lds_bytes, u32 wit_native) {
    if (blockIdx.x >= rng_blocks) {
        witness_gates_block((blockIdx.x - rng_blocks) * blockDim.x, B, m, n1, n_cst, w_terms, w_loff, w_roff, cst_all, v_all, ai1_all, ao1_all, lds_bytes, wit_native);
        return;
    }
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 g = t >> 6, L = t & 63u;
    if (g >= B) return;  // a spare wavefront of the last workgroup (no barrier in this kernel): it would run a whole chain beside a live one, on the same LDS crossbar
    const kw_lane c = kw_setup(L);
    const kw_iota k = kw_iota_setup(L);
    const bool live = g < B && c.live != 0;
    const u32 p = g < B ? g : B - 1;
    merlin_transcript* T = rng + p;
    u32 a = c.live ? kw_half(T->st[c.word], c.half) : 0u;
    // per-draw constants (keccak.h merlin_rng_fill64_bulk): st[8] ^= .., st[9] ^= .., st[20] ^= ..
    const u64 konst64 = !c.live ? 0ull : c.word == 8 ? 0x0741000000401200ull : c.word == 9 ? 0x0000000000000447ull : c.word == 20 ? 0x8000000000000000ull : 0ull;
    const u32 konst = kw_half(konst64, c.half);
    const bool rate = c.live && c.word < 8;
    u32* out = raw + (size_t)p * (3 + 2 * (size_t)n1) * 16 + 16 + 2 * (c.word & 7u) + c.half;  // draw 0 was written by the prefix kernel
#ifdef BBP_KO_RNG_CHAIN  // timing experiment (wrong results): no draws, the launch lasts as long as its witness blocks
    count = 0;
#endif
    for (u32 d = 0; d < count; d++) {
        a ^= konst;
        a = kw_keccak_f(a, c, k);
        if (rate) {
            if (live) out[(size_t)d * 16] = a;
            a = 0;
        }
    }
    const auto q = __builtin_amdgcn_permlane32_swap(a, a, false, false);  // the odd halves, for the lanes of the even ones
    if (live && c.lower) T->st[c.word] = kw_join(a, q[1]);
    if (live && L == 0) T->cur_flags = BBP_FLAG_I | BBP_FLAG_A | BBP_FLAG_C;
}

// draw j of proof p -> its scalar slot: 0 -> ai1[0], 1 -> ao1[0], 2 -> s1[0], j >= 3 -> s1[1 + (j - 3)]
// interleaved: draws 1.. were written by k_open_bulk50 as (even bits, odd bits) pairs per 64-bit word
__global__ void k_reduce_draws(u32 B, u32 n1, const u32* __restrict__ raw, sc* __restrict__ ai1, sc* __restrict__ ao1, sc* __restrict__ s1, u32 interleaved) {
    const u32 per = 3 + 2 * n1;
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * per) return;
    u32 p = t / per, j = t % per;
    const uint4* q = reinterpret_cast<const uint4*>(raw + (size_t)t * 16);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    u32 w[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    if (interleaved && j > 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u64 v = kw_join(w[2 * i], w[2 * i + 1]);
            w[2 * i] = (u32)v;
            w[2 * i + 1] = (u32)(v >> 32);
        }
    }
    sc v = sc_from_wide(w);
    sc* dst = j == 0 ? &ai1[(size_t)p * (1 + 2 * n1)] : j == 1 ? &ao1[(size_t)p * (1 + n1)] : &s1[(size_t)p * (1 + 2 * n1) + (j - 2)];
    st_sc(dst, v);
}

// blindings from the entropy block: (4+N) 32-byte values, reduced
__global__ void k_load_blindings(u32 B, u32 m, const u8* __restrict__ entropy, sc* __restrict__ vb_all) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * m) return;
    u32 p = t / m, i = t % m;
    const u32* w = reinterpret_cast<const u32*>(entropy + (size_t)p * (32 * (size_t)m + 32) + 32 * (size_t)i);
    st_sc(&vb_all[(size_t)p * m + i], sc_reduce256(w));
}

__global__ BBP_LANE_KERNEL void k_tr_yz(u32 B, u32 m, const u32* __restrict__ enc, merlin_transcript* __restrict__ tr, sc* __restrict__ misc, u32 wave) {
    BBP_THIN_PRIO();
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (wave) p >>= 6;  // one transcript per wavefront, every lane in lockstep (keccak_wave.h keccak_f1600_wave)
    if (p >= B) return;
    merlin_transcript t = tr[p];
    t.wave = wave;
    const u32* e = enc + (size_t)p * enc_stride_words(m) + 8 * m;
    tr_append_words(t, LBL("A_I1"), e);
    tr_append_words(t, LBL("A_O1"), e + 8);
    tr_append_words(t, LBL("S1"), e + 16);
    merlin_append(t, LBL("dom-sep"), LBL("r1cs-1phase"));  // no randomized constraints in this circuit (A.5 step 5)
    const u32 ident[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tr_append_words(t, LBL("A_I2"), ident);
    tr_append_words(t, LBL("A_O2"), ident);
    tr_append_words(t, LBL("S2"), ident);
    sc y = tr_challenge_sc(t, LBL("y"));
    sc z = tr_challenge_sc(t, LBL("z"));
    sc* ms = misc + (size_t)p * MS_COUNT;
    st_sc(&ms[MS_Y], y);
    st_sc(&ms[MS_Z], z);
    st_sc(&ms[MS_YINV], sc_invert(y));
    t.wave = 0;  // (memory never holds the flag: every kernel sets it for itself)
    tr[p] = t;
}

// out[p][e] = base[p]^e for e in [0, count): one lane per chunk of 32 exponents
// mont_out = 0: out[e] = x^e;  1: out[e] = x^e R (Montgomery form, for consumers that multiply by it once).  Either way one
// Montgomery multiplication per power: cur * (x R) * R^-1 = cur * x keeps cur in whichever domain it started in.
__device__ __forceinline__ void powers_lane(u32 B, u32 count, const sc* __restrict__ misc, int slot, sc* __restrict__ out, u32 out_stride, u32 mont_out) {
    const u32 chunks = (count + 31) / 32;
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * chunks) return;
    u32 p = t / chunks, c = t % chunks;
    sc x = ld_sc(&misc[(size_t)p * MS_COUNT + slot]);
    sc xm = sc_to_mont(x);
    sc cur = sc_pow_small(x, c * 32);
    if (mont_out) cur = sc_to_mont(cur);
    sc* o = out + (size_t)p * out_stride;
    u32 end = min(count, c * 32 + 32);
    for (u32 e = c * 32; e < end; e++) {
        st_sc(&o[e], cur);
        cur = sc_montmul(cur, xm);
    }
}
__global__ BBP_LANE_KERNEL void k_powers(u32 B, u32 count, const sc* __restrict__ misc, int slot, sc* __restrict__ out, u32 out_stride, u32 mont_out) {
    BBP_THIN_PRIO();
    powers_lane(B, count, misc, slot, out, out_stride, mont_out);
}
// the prover's three power tables (z^k, y^k, y^-k) in one launch: blockIdx.y picks the table (three launches of ~64 us each were
// three dependent-chain latencies of a small call)
struct PowersJob {
    u32 count;
    int slot;
    sc* out;
    u32 out_stride, mont_out;
};
__global__ BBP_LANE_KERNEL void k_powers3(u32 B, const sc* __restrict__ misc, PowersJob a, PowersJob b, PowersJob c) {
    BBP_THIN_PRIO();
    const PowersJob j = blockIdx.y == 0 ? a : blockIdx.y == 1 ? b : c;
    powers_lane(B, j.count, misc, j.slot, j.out, j.out_stride, j.mont_out);
}

// K4: flattened constraint weights.  target t of proof p = sum over its entries of +-z^(q+1)
__global__ void k_flatten(u32 B, u32 n_tgt, u32 n_mul, u32 m, const u32* __restrict__ f_off, const u32* __restrict__ f_ent,
                          const sc* __restrict__ zpow, u32 zstride, sc* __restrict__ wl, sc* __restrict__ wr, sc* __restrict__ wo,
                          sc* __restrict__ wv, u32 wstride) {
    BBP_THIN_PRIO();
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n_tgt) return;
    u32 p = t / n_tgt, k = t % n_tgt;
    const sc* zp = zpow + (size_t)p * zstride;
    sc acc = sc_zero();
    // eight entries at a time: their powers are fetched together (independent loads).  A handful of targets -- a MiMC key, the
    // hash input x -- sit in several hundred constraints, and one load round trip per entry made those lanes the whole kernel
    // (334 us for ONE proof)
    const u32 e0 = f_off[k], e1 = f_off[k + 1];
    for (u32 e = e0; e < e1; e += 8) {
        u32 w[8];
        sc zq[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = f_ent[min(e + (u32)i, e1 - 1)];
#pragma unroll
        for (int i = 0; i < 8; i++) zq[i] = ld_sc(&zp[(w[i] & 0x7fffffffu) + 1]);
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (e + (u32)i < e1) acc = (w[i] >> 31) ? sc_sub(acc, zq[i]) : sc_add(acc, zq[i]);
    }
    sc* dst = k < n_mul ? &wl[(size_t)p * wstride + k]
              : k < 2 * n_mul ? &wr[(size_t)p * wstride + (k - n_mul)]
              : k < 3 * n_mul ? &wo[(size_t)p * wstride + (k - 2 * n_mul)]
                              : &wv[(size_t)p * m + (k - 3 * n_mul)];
    st_sc(dst, acc);
}
// k_flatten for launches of a few proofs: a target on FLAT_L lanes (lane q takes the entries e0 + q, e0 + q + FLAT_L, ...; three shuffle
// steps add the partial sums).  A MiMC key or the hash input x sits in several hundred constraints: one lane per target is a 200 us
// chain for ONE proof however its loads are batched.
constexpr int FLAT_L = 8;
__global__ void k_flatten_split(u32 B, u32 n_tgt, u32 n_mul, u32 m, const u32* __restrict__ f_off, const u32* __restrict__ f_ent,
                                const sc* __restrict__ zpow, u32 zstride, sc* __restrict__ wl, sc* __restrict__ wr, sc* __restrict__ wo,
                                sc* __restrict__ wv, u32 wstride) {
    BBP_THIN_PRIO();
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x, g = t / FLAT_L, q = t % FLAT_L;
    const u32 gi = g < B * n_tgt ? g : B * n_tgt - 1;  // whole wavefronts reach the shuffles
    const u32 p = gi / n_tgt, k = gi % n_tgt;
    const sc* zp = zpow + (size_t)p * zstride;
    sc acc = sc_zero();
    const u32 e1 = f_off[k + 1];
    for (u32 e = f_off[k] + q; e < e1; e += FLAT_L) {
        const u32 w = f_ent[e];
        const sc zq = ld_sc(&zp[(w & 0x7fffffffu) + 1]);
        acc = (w >> 31) ? sc_sub(acc, zq) : sc_add(acc, zq);
    }
#pragma unroll 1
    for (int d = FLAT_L / 2; d >= 1; d >>= 1) {
        sc o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = (u32)__shfl_down((int)acc.v[i], d, 64);
        acc = sc_add(acc, o);  // (lanes whose partner lies in the next group add something nobody reads)
    }
    if (q != 0 || g >= B * n_tgt) return;
    sc* dst = k < n_mul ? &wl[(size_t)p * wstride + k]
              : k < 2 * n_mul ? &wr[(size_t)p * wstride + (k - n_mul)]
              : k < 3 * n_mul ? &wo[(size_t)p * wstride + (k - 2 * n_mul)]
                              : &wv[(size_t)p * m + (k - 3 * n_mul)];
    st_sc(dst, acc);
}

// block-wide sum of NV scalars per lane through LDS (BLK lanes); result valid in lane 0
template <int NV, int BLK>
__device__ void block_sum_sc(sc (&vals)[NV], u32* lds /* NV*8*BLK words */) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NV; k++)
#pragma unroll
        for (int w = 0; w < 8; w++) lds[(k * 8 + w) * BLK + tid] = vals[k].v[w];
    __syncthreads();
    for (int d = BLK / 2; d >= 1; d >>= 1) {
        if (tid < d) {
#pragma unroll
            for (int k = 0; k < NV; k++) {
                sc o;
#pragma unroll
                for (int w = 0; w < 8; w++) o.v[w] = lds[(k * 8 + w) * BLK + tid + d];
                vals[k] = sc_add(vals[k], o);
#pragma unroll
                for (int w = 0; w < 8; w++) lds[(k * 8 + w) * BLK + tid] = vals[k].v[w];
            }
        }
        __syncthreads();
    }
}

// K5: l/r polynomial coefficient sweep and the t coefficients (A.5 steps 8-9); one 256-lane block per proof
constexpr int POLY_BLK = 256;
__global__ __launch_bounds__(POLY_BLK) void k_poly(u32 n1, const sc* __restrict__ ai1, const sc* __restrict__ ao1, const sc* __restrict__ s1,
                                                    const sc* __restrict__ wl, const sc* __restrict__ wr, const sc* __restrict__ wo,
                                                    u32 wstride, const sc* __restrict__ ypow, const sc* __restrict__ yipow,
                                                    sc* __restrict__ l1o, sc* __restrict__ r0o, sc* __restrict__ r1o, sc* __restrict__ r3o,
                                                    sc* __restrict__ misc) {
    BBP_THIN_PRIO();
    __shared__ u32 lds[6 * 8 * POLY_BLK];
    const u32 p = blockIdx.x;
    const sc* aL = ai1 + (size_t)p * (1 + 2 * n1) + 1;
    const sc* aR = aL + n1;
    const sc* aO = ao1 + (size_t)p * (1 + n1) + 1;
    const sc* sL = s1 + (size_t)p * (1 + 2 * n1) + 1;
    const sc* sR = sL + n1;
    const sc *WL = wl + (size_t)p * wstride, *WR = wr + (size_t)p * wstride, *WO = wo + (size_t)p * wstride;
    const sc *Y = ypow + (size_t)p * 2049, *YI = yipow + (size_t)p * 2048;
    sc t[6];
#pragma unroll
    for (int k = 0; k < 6; k++) t[k] = sc_zero();
    for (u32 i = threadIdx.x; i < n1; i += POLY_BLK) {
        sc y = ld_sc(&Y[i]);
        sc l1 = sc_add(ld_sc(&aL[i]), sc_mul(ld_sc(&YI[i]), ld_sc(&WR[i])));
        sc l2 = ld_sc(&aO[i]), l3 = ld_sc(&sL[i]);
        sc r0 = sc_sub(ld_sc(&WO[i]), y);
        sc r1 = sc_add(sc_mul(y, ld_sc(&aR[i])), ld_sc(&WL[i]));
        sc r3 = sc_mul(y, ld_sc(&sR[i]));
        st_sc(&l1o[(size_t)p * n1 + i], l1);
        st_sc(&r0o[(size_t)p * n1 + i], r0);
        st_sc(&r1o[(size_t)p * n1 + i], r1);
        st_sc(&r3o[(size_t)p * n1 + i], r3);
        // the six sums collect x y R^-1 (single Montgomery multiplications); one conversion each after the block sum
        t[0] = sc_add(t[0], sc_montmul(l1, r0));
        t[1] = sc_add(t[1], sc_add(sc_montmul(l1, r1), sc_montmul(l2, r0)));
        t[2] = sc_add(t[2], sc_add(sc_montmul(l2, r1), sc_montmul(l3, r0)));
        t[3] = sc_add(t[3], sc_add(sc_montmul(l1, r3), sc_montmul(l3, r1)));
        t[4] = sc_add(t[4], sc_montmul(l2, r3));
        t[5] = sc_add(t[5], sc_montmul(l3, r3));
    }
    block_sum_sc<6, POLY_BLK>(t, lds);
    if (threadIdx.x == 0) {
        sc* ms = misc + (size_t)p * MS_COUNT;
#pragma unroll
        for (int k = 0; k < 6; k++) st_sc(&ms[MS_T1 + k], sc_to_mont(t[k]));  // (t R^-1) R^2 R^-1 = t
    }
}

__global__ BBP_LANE_KERNEL void k_tr_tblind(u32 B, merlin_transcript* __restrict__ rng, sc* __restrict__ misc, u32 wave) {
    BBP_THIN_PRIO();
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (wave) p >>= 6;  // one transcript per wavefront, every lane in lockstep (keccak_wave.h keccak_f1600_wave)
    if (p >= B) return;
    merlin_transcript r = rng[p];
    r.wave = wave;
    sc* ms = misc + (size_t)p * MS_COUNT;
    st_sc(&ms[MS_TB1], rng_scalar(r));  // t_1, t_3, t_4, t_5, t_6 blindings, in this order (A.5 step 10)
    st_sc(&ms[MS_TB3], rng_scalar(r));
    st_sc(&ms[MS_TB4], rng_scalar(r));
    st_sc(&ms[MS_TB5], rng_scalar(r));
    st_sc(&ms[MS_TB6], rng_scalar(r));
    r.wave = 0;  // (memory never holds the flag: every kernel sets it for itself)
    rng[p] = r;
}

// T_k = t_k B + tb_k B~ for k in {1,3,4,5,6}: one lane per (proof, k)
__global__ BBP_LANE_KERNEL void k_commit_T_split(u32 B, const sc* __restrict__ misc, const niels_packed* __restrict__ comb, ge* __restrict__ pts, u32 pts_stride, u32 m) {
    BBP_THIN_PRIO();
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x, g = t / COMMIT_L, q = t % COMMIT_L;
    const u32 gi = g < B * 5 ? g : B * 5 - 1;
    const u32 p = gi / 5, k = gi % 5;
    const int slot[5] = {0, 2, 3, 4, 5};
    const sc* ms = misc + (size_t)p * MS_COUNT;
    ge acc = comb_mul_add_part(ge_identity(), comb, ld_sc(&ms[MS_T1 + slot[k]]), q);
    acc = comb_mul_add_part(acc, comb + 64 * 8, ld_sc(&ms[MS_TB1 + slot[k]]), q);
    acc = commit_group_sum(acc);
    if (q == 0 && g < B * 5) pts[(size_t)p * pts_stride + m + 3 + k] = acc;
}
__global__ BBP_LANE_KERNEL void k_commit_T(u32 B, const sc* __restrict__ misc, const niels_packed* __restrict__ comb, ge* __restrict__ pts, u32 pts_stride, u32 m) {
    BBP_THIN_PRIO();
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 5) return;
    u32 p = t / 5, k = t % 5;
    const int slot[5] = {0, 2, 3, 4, 5};
    const sc* ms = misc + (size_t)p * MS_COUNT;
    ge acc = comb_mul_add(ge_identity(), comb, ld_sc(&ms[MS_T1 + slot[k]]));
    acc = comb_mul_add(acc, comb + 64 * 8, ld_sc(&ms[MS_TB1 + slot[k]]));
    pts[(size_t)p * pts_stride + m + 3 + k] = acc;
}

__global__ BBP_LANE_KERNEL void k_tr_ux(u32 B, u32 m, u32 n1, const u32* __restrict__ enc, const sc* __restrict__ wv, const sc* __restrict__ vb,
                        const sc* __restrict__ ai1, const sc* __restrict__ ao1, const sc* __restrict__ s1,
                        merlin_transcript* __restrict__ tr, sc* __restrict__ misc, u32 wave) {
    BBP_THIN_PRIO();
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (wave) p >>= 6;  // one transcript per wavefront, every lane in lockstep (keccak_wave.h keccak_f1600_wave)
    if (p >= B) return;
    merlin_transcript t = tr[p];
    t.wave = wave;
    const u32* e = enc + (size_t)p * enc_stride_words(m) + 8 * (m + 3);
    tr_append_words(t, LBL("T_1"), e);
    tr_append_words(t, LBL("T_3"), e + 8);
    tr_append_words(t, LBL("T_4"), e + 16);
    tr_append_words(t, LBL("T_5"), e + 24);
    tr_append_words(t, LBL("T_6"), e + 32);
    sc u = tr_challenge_sc(t, LBL("u"));
    sc x = tr_challenge_sc(t, LBL("x"));
    sc* ms = misc + (size_t)p * MS_COUNT;
    sc tb2 = sc_zero();  // t_2_blinding = <wV, v_blinding> (A.5 step 12)
    for (u32 i = 0; i < m; i++) tb2 = sc_add(tb2, sc_mul(ld_sc(&wv[(size_t)p * m + i]), ld_sc(&vb[(size_t)p * m + i])));
    st_sc(&ms[MS_TB2], tb2);
    sc t_x = sc_zero(), t_xb = sc_zero(), xp = x;
    for (int k = 0; k < 6; k++) {
        t_x = sc_add(t_x, sc_mul(ld_sc(&ms[MS_T1 + k]), xp));
        t_xb = sc_add(t_xb, sc_mul(ld_sc(&ms[MS_TB1 + k]), xp));
        xp = sc_mul(xp, x);
    }
    sc ib = ld_sc(&ai1[(size_t)p * (1 + 2 * n1)]), ob = ld_sc(&ao1[(size_t)p * (1 + n1)]), sb = ld_sc(&s1[(size_t)p * (1 + 2 * n1)]);
    sc e_bl = sc_mul(x, sc_add(ib, sc_mul(x, sc_add(ob, sc_mul(x, sb)))));  // phase-2 blindings are zero (A.5 step 14)
    tr_append_sc(t, LBL("t_x"), t_x);
    tr_append_sc(t, LBL("t_x_blinding"), t_xb);
    tr_append_sc(t, LBL("e_blinding"), e_bl);
    sc w = tr_challenge_sc(t, LBL("w"));
    // InnerProductProof::create opens with its domain separator (A.6)
    merlin_append(t, LBL("dom-sep"), LBL("ipp v1"));
    merlin_append_u64(t, LBL("n"), 2048);
    st_sc(&ms[MS_U], u);
    st_sc(&ms[MS_X], x);
    st_sc(&ms[MS_W], w);
    st_sc(&ms[MS_TX], t_x);
    st_sc(&ms[MS_TXB], t_xb);
    st_sc(&ms[MS_EBL], e_bl);
    t.wave = 0;  // (memory never holds the flag: every kernel sets it for itself)
    tr[p] = t;
}

// l(x), r(x) (A.5 step 13) and the generator factor vectors g[k] = G_factors[k], h[k] = y^-k G_factors[k] (step 16)
__global__ void k_lrvec(u32 B, u32 n1, const sc* __restrict__ l1, const sc* __restrict__ r0, const sc* __restrict__ r1, const sc* __restrict__ r3,
                        const sc* __restrict__ ao1, const sc* __restrict__ s1, const sc* __restrict__ ypow, const sc* __restrict__ yipow,
                        const sc* __restrict__ misc, sc* __restrict__ a, sc* __restrict__ b, sc* __restrict__ g, sc* __restrict__ h) {
    BBP_THIN_PRIO();
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 2048) return;
    u32 p = t >> 11, i = t & 2047;
    const sc* ms = misc + (size_t)p * MS_COUNT;
    sc x = ld_sc(&ms[MS_X]);
    sc av, bv, gv;
    if (i < n1) {
        // x R, x^2 R, x^3 R: each product with a coefficient is then one Montgomery multiplication
        const sc xr = sc_to_mont(x), x2r = sc_montmul(xr, xr), x3r = sc_montmul(x2r, xr);
        sc l2 = ld_sc(&ao1[(size_t)p * (1 + n1) + 1 + i]), l3 = ld_sc(&s1[(size_t)p * (1 + 2 * n1) + 1 + i]);
        av = sc_add(sc_add(sc_montmul(ld_sc(&l1[(size_t)p * n1 + i]), xr), sc_montmul(l2, x2r)), sc_montmul(l3, x3r));
        bv = sc_add(sc_add(ld_sc(&r0[(size_t)p * n1 + i]), sc_montmul(ld_sc(&r1[(size_t)p * n1 + i]), xr)),
                    sc_montmul(ld_sc(&r3[(size_t)p * n1 + i]), x3r));
        gv = sc_one();
    } else {
        av = sc_zero();
        bv = sc_neg(ld_sc(&ypow[(size_t)p * 2049 + i]));
        gv = ld_sc(&ms[MS_U]);
    }
    st_sc(&a[(size_t)p * 2048 + i], av);
    st_sc(&b[(size_t)p * 2048 + i], bv);
    // the factor vectors live in Montgomery form (x R mod l) through the MSM rounds: every use is a product with a plain
    // scalar, which then costs ONE Montgomery multiplication instead of two (k_ipa_round)
    st_sc(&g[(size_t)p * 2048 + i], sc_to_mont(gv));
    st_sc(&h[(size_t)p * 2048 + i], sc_montmul(sc_to_mont(ld_sc(&yipow[(size_t)p * 2048 + i])), sc_to_mont(gv)));
}

// Fiat-Shamir step between IPA rounds: absorb L_j, R_j (j = prev_round), draw u_j, invert it.  One lane per proof.
__global__ BBP_LANE_KERNEL void k_ipa_challenge(u32 B, u32 prev_round, u32 m, const u32* __restrict__ enc, merlin_transcript* __restrict__ tr,
                                sc* __restrict__ misc, u32 wave) {
    BBP_THIN_PRIO();
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (wave) p >>= 6;  // one transcript per wavefront, every lane in lockstep (keccak_wave.h keccak_f1600_wave)
    if (p >= B) return;
    merlin_transcript t = tr[p];
    t.wave = wave;
    const u32* e = enc + (size_t)p * enc_stride_words(m) + 8 * (m + 8) + 16 * (prev_round - 1);
    tr_append_words(t, LBL("L"), e);
    tr_append_words(t, LBL("R"), e + 8);
    sc u = tr_challenge_sc(t, LBL("u"));
    sc* ms = misc + (size_t)p * MS_COUNT;
    st_sc(&ms[MS_UJ], u);
    st_sc(&ms[MS_UJI], sc_invert(u));
    t.wave = 0;  // (memory never holds the flag: every kernel sets it for itself)
    tr[p] = t;
}

// K6: one inner-product round.  n = half length of THIS round (1024 >> (round-1)).
//   round > 1: with u, 1/u of the previous round (k_ipa_challenge) fold a, b and update the factor vectors g, h.
//   then: c_L, c_R and the scalars of this round's L and R over the ORIGINAL generators.
constexpr int IPA_BLK = 256, IPA_BLK_WIDE = 1024;  // lanes per proof: 256, or 1024 for launches of a few proofs (eight elements per lane are a 75 us chain, eleven times per proof)
template <int BLK>
__global__ __launch_bounds__(BLK) void k_ipa_round(u32 round, u32 n1, sc* __restrict__ misc, sc* __restrict__ a_all, sc* __restrict__ b_all,
                                                        sc* __restrict__ g_all, sc* __restrict__ h_all, sc* __restrict__ lr_all) {
    BBP_THIN_PRIO();
    __shared__ u32 lds[2 * 8 * BLK];
    const u32 p = blockIdx.x, tid = threadIdx.x;
    sc* ms = misc + (size_t)p * MS_COUNT;
    sc *a = a_all + (size_t)p * 2048, *b = b_all + (size_t)p * 2048, *g = g_all + (size_t)p * 2048, *h = h_all + (size_t)p * 2048;
    const u32 n = 1024u >> (round - 1);
    if (round > 1) {
        // u, u^-1 of the previous round were produced by k_ipa_challenge (one lane per proof, its own tiny launch, so that
        // this 256-lane block is never resident while a single lane hashes and inverts)
        // u R, u^-1 R: products with them are single Montgomery multiplications, x * (u R) * R^-1 = x u
        const sc u = sc_to_mont(ld_sc(&ms[MS_UJ])), ui = sc_to_mont(ld_sc(&ms[MS_UJI]));
        const u32 n2 = 2 * n;  // half length of the previous round = current full length
        // fold a, b: a'[i] = a[i] u + u^-1 a[n2+i] ; b'[i] = b[i] u^-1 + u b[n2+i]
        for (u32 i = tid; i < n2; i += BLK) {
            sc alo = ld_sc(&a[i]), ahi = ld_sc(&a[n2 + i]), blo = ld_sc(&b[i]), bhi = ld_sc(&b[n2 + i]);
            st_sc(&a[i], sc_add(sc_montmul(alo, u), sc_montmul(ui, ahi)));
            st_sc(&b[i], sc_add(sc_montmul(blo, ui), sc_montmul(u, bhi)));
        }
        // generator factors (Montgomery form in, Montgomery form out): low half of each 2*n2 block took u^-1 (G) / u (H); high
        // half the opposite.  The tail preparation call hands them to the generator-fold MSM, which needs plain scalars.
        const bool to_plain = lr_all == nullptr;
        for (u32 k = tid; k < 2048; k += BLK) {
            bool hi = (k & (2 * n2 - 1)) >= n2;
            sc gk = sc_montmul(ld_sc(&g[k]), hi ? u : ui), hk = sc_montmul(ld_sc(&h[k]), hi ? ui : u);
            st_sc(&g[k], to_plain ? sc_from_mont(gk) : gk);
            st_sc(&h[k], to_plain ? sc_from_mont(hk) : hk);
        }
        __syncthreads();
    }
    if (lr_all == nullptr) return;  // tail preparation: only the fold of a, b and the factor update were wanted
    // scalars of L and R; c_L = <a_lo, b_hi>, c_R = <a_hi, b_lo>
    sc* L = lr_all + (size_t)p * 2 * 2049;
    sc* R = L + 2049;
    sc c[2] = {sc_zero(), sc_zero()};
    for (u32 k = tid; k < 2048; k += BLK) {
        const u32 i = k & (2 * n - 1), blk = k / (2 * n);
        const bool hi = i >= n;
        const u32 io = hi ? i - n : i, rank = blk * n + io;
        sc gk = ld_sc(&g[k]), hk = ld_sc(&h[k]);  // Montgomery form: a * (g R) * R^-1 = a g
        if (hi) {
            st_sc(&L[rank], sc_montmul(ld_sc(&a[io]), gk));            // a_L[io] * G_R
            st_sc(&R[1024 + rank], sc_montmul(ld_sc(&b[io]), hk));     // b_L[io] * H_R
        } else {
            st_sc(&R[rank], sc_montmul(ld_sc(&a[n + io]), gk));        // a_R[io] * G_L
            // b_R[io] * H_L.  Round 1: for the zero-padded multipliers i = n + io >= n1 this product is -y^1024 whatever io is;
            // only the first of them keeps it, on the precomputed sum of those H's (PAD_BASE0, circuit_get), the rest are zero
            const bool dup = round == 1 && n + io > n1;
            st_sc(&L[1024 + rank], dup ? sc_zero() : sc_montmul(ld_sc(&b[n + io]), hk));
        }
    }
    for (u32 i = tid; i < n; i += BLK) {
        sc alo = ld_sc(&a[i]), ahi = ld_sc(&a[n + i]), blo = ld_sc(&b[i]), bhi = ld_sc(&b[n + i]);
        c[0] = sc_add(c[0], sc_montmul(alo, bhi));  // sums of x y R^-1: one conversion after the block sum
        c[1] = sc_add(c[1], sc_montmul(ahi, blo));
    }
    block_sum_sc<2, BLK>(c, lds);
    if (tid == 0) {
        const sc wr = sc_montmul(sc_to_mont(ld_sc(&ms[MS_W])), sc_rr());  // w R^2: (c R^-1) * (w R^2) * R^-1 = c w.  Q = w B, c * Q = (c w) B
        st_sc(&L[2048], sc_montmul(c[0], wr));
        st_sc(&R[2048], sc_montmul(c[1], wr));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// IPA tail on explicit folded generators (rounds FOLD_ROUND..11, vectors of length <= 32)
// ---------------------------------------------------------------------------------------------------------------
// signed radix-16 digits of a canonical scalar: d[0..63] in [-8, 8], d[64] = final carry
__device__ void sc_radix16(const sc& s, int8_t* d) {
    u32 carry = 0;
    for (int j = 0; j < 64; j++) {
        u32 v = ((s.v[j >> 3] >> (4 * (j & 7))) & 15u) + carry;
        carry = v > 8u;
        d[j] = (int8_t)(carry ? (int)v - 16 : (int)v);
    }
    d[64] = (int8_t)carry;
}

// s1*P1 + s2*P2 with shared doublings; tab: 16 points of scratch owned by the calling lane (1..8 multiples of each point)
__device__ ge ge_double_scalarmul(const sc& s1, const ge& P1, const sc& s2, const ge& P2, ge* tab, bool two) {
    tab[0] = P1;
    ge cur = P1;
    for (int i = 1; i < 8; i++) {
        cur = ge_add(cur, P1);
        tab[i] = cur;
    }
    if (two) {
        tab[8] = P2;
        cur = P2;
        for (int i = 1; i < 8; i++) {
            cur = ge_add(cur, P2);
            tab[8 + i] = cur;
        }
    }
    int8_t d1[65], d2[65];
    sc_radix16(s1, d1);
    if (two) sc_radix16(s2, d2);
    ge acc = ge_identity();
    for (int j = 64; j >= 0; j--) {
        if (j != 64) {
            acc = ge_dbl(acc);
            acc = ge_dbl(acc);
            acc = ge_dbl(acc);
            acc = ge_dbl(acc);
        }
        int a = d1[j];
        if (a != 0) {
            ge q = tab[(a > 0 ? a : -a) - 1];
            if (a < 0) q = ge_neg(q);
            acc = ge_add(acc, q);
        }
        if (two) {
            int b = d2[j];
            if (b != 0) {
                ge q = tab[8 + (b > 0 ? b : -b) - 1];
                if (b < 0) q = ge_neg(q);
                acc = ge_add(acc, q);
            }
        }
    }
    return acc;
}

// Tail tables: for a point P the multiples m * 2^(w k) * P, m = 1..8, k = 0..TAIL_PIECES-1, w = 256 / TAIL_PIECES.  A scalar
// multiplication over such a table is (w - 4) doublings + 64 additions (signed radix-16 digits, the pieces share the doublings)
// instead of 252 + 64 + 7, and the tables of the 64 materialised generators are built once and used by all five tail rounds.
// Four 64-bit pieces: 60 doublings per multiplication, 192 to build a table; eight 32-bit pieces: 28 and 224 (+ 28 additions): the
// tail launches are chain-bound, a table serves five rounds.
#ifndef BBP_TAIL_PIECES
#define BBP_TAIL_PIECES 8
#endif
constexpr int TAIL_PIECES = BBP_TAIL_PIECES, TAIL_PIECE_BITS = 256 / TAIL_PIECES, TAIL_DIGITS = TAIL_PIECE_BITS / 4;
constexpr int TAIL_TAB = 8 * TAIL_PIECES;
static_assert(TAIL_PIECES == 4 || TAIL_PIECES == 8 || TAIL_PIECES == 16, "tail table geometry");
// (64-lane workgroups and a 2-3 waves/SIMD register budget: with the default 128-VGPR cap this kernel spilled 216 registers)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 3))) void k_tail_tables(u32 count, const ge* __restrict__ pts, ge* __restrict__ tab) {
    BBP_THIN_PRIO();
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    ge P = pts[t];
    ge* T = tab + (size_t)t * TAIL_TAB;
#pragma unroll 1
    for (int k = 0; k < TAIL_PIECES; k++) {
        ge cur = P;
        T[8 * k] = P;
#pragma unroll 1
        for (int i = 1; i < 8; i++) {
            cur = ge_add(cur, P);
            T[8 * k + i] = cur;
        }
        if (k < TAIL_PIECES - 1) {
#pragma unroll 1
            for (int i = 0; i < TAIL_PIECE_BITS; i++) P = ge_dbl(P);
        }
    }
}

// pieces k_lo <= k < k_hi only: the partial product sum_k 2^(w k) * (piece k of s) * P
__device__ ge ge_scalarmul_pieces(const sc& s, const ge* __restrict__ T, int k_lo = 0, int k_hi = TAIL_PIECES) {
    // carry mask of the signed radix-16 recoding: bit j = carry INTO digit j (a canonical scalar never carries out of digit 63)
    u64 cm = 0;
    u32 c = 0;
    for (int j = 0; j < 64; j++) {
        u32 v = ((s.v[j >> 3] >> (4 * (j & 7))) & 15u) + c;
        c = v > 8u;
        if (j < 63) cm |= (u64)c << (j + 1);
    }
    ge acc = ge_identity();
    for (int r = TAIL_DIGITS - 1; r >= 0; r--) {
        if (r != TAIL_DIGITS - 1) {
            acc = ge_dbl(acc);
            acc = ge_dbl(acc);
            acc = ge_dbl(acc);
            acc = ge_dbl(acc);
        }
        for (int k = k_lo; k < k_hi; k++) {
            const int j = TAIL_DIGITS * k + r;
            const int d = (int)((s.v[j >> 3] >> (4 * (j & 7))) & 15u) + (int)((cm >> j) & 1u) - 16 * (int)((j < 63) ? ((cm >> (j + 1)) & 1u) : 0u);
            if (d != 0) {
                ge q = T[8 * k + (d > 0 ? d : -d) - 1];
                if (d < 0) q = ge_neg(q);
                acc = ge_add(acc, q);
            }
        }
    }
    return acc;
}

// s1 * P1 + s2 * P2 over tail tables T1, T2 with ONE doubling chain (the merged form of k_tail_lr: two terms per lane)
__device__ ge ge_scalarmul_pieces_pair(const sc& s1, const ge* __restrict__ T1, const sc& s2, const ge* __restrict__ T2) {
    u64 cm1 = 0, cm2 = 0;
    u32 c1 = 0, c2 = 0;
    for (int j = 0; j < 64; j++) {
        const u32 v1 = ((s1.v[j >> 3] >> (4 * (j & 7))) & 15u) + c1, v2 = ((s2.v[j >> 3] >> (4 * (j & 7))) & 15u) + c2;
        c1 = v1 > 8u;
        c2 = v2 > 8u;
        if (j < 63) {
            cm1 |= (u64)c1 << (j + 1);
            cm2 |= (u64)c2 << (j + 1);
        }
    }
    ge acc = ge_identity();
    for (int r = TAIL_DIGITS - 1; r >= 0; r--) {
        if (r != TAIL_DIGITS - 1) {
            acc = ge_dbl(acc);
            acc = ge_dbl(acc);
            acc = ge_dbl(acc);
            acc = ge_dbl(acc);
        }
#pragma unroll 1
        for (int kk = 0; kk < 2 * TAIL_PIECES; kk++) {  // piece k of scalar 1, then piece k of scalar 2: one addition site
            const int k = kk >> 1;
            const bool second = kk & 1;
            const int j = TAIL_DIGITS * k + r;
            const u32 word = second ? s2.v[j >> 3] : s1.v[j >> 3];
            const u64 cm = second ? cm2 : cm1;
            const int d = (int)((word >> (4 * (j & 7))) & 15u) + (int)((cm >> j) & 1u) - 16 * (int)((j < 63) ? ((cm >> (j + 1)) & 1u) : 0u);
            if (d != 0) {
                ge q = (second ? T2 : T1)[8 * k + (d > 0 ? d : -d) - 1];
                if (d < 0) q = ge_neg(q);
                acc = ge_add(acc, q);
            }
        }
    }
    return acc;
}

// The tail never folds points either: like the main rounds it keeps per-generator factor scalars (gg, hh: 32 each, stored in
// the first 32 slots of bd.g / bd.h once the big factor vectors have been consumed by the generator-fold MSM) and multiplies
// them into the term scalars.  Every tail round is then ONE launch of 2 x 33 independent scalar multiplications per proof.
__global__ void k_tail_init(u32 B, sc* __restrict__ g_all, sc* __restrict__ h_all) {
    BBP_THIN_PRIO();
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * FOLD_CLS) return;
    const u32 p = t / FOLD_CLS, k = t % FOLD_CLS;
    st_sc(&g_all[(size_t)p * 2048 + k], sc_one());
    st_sc(&h_all[(size_t)p * 2048 + k], sc_one());
}

// L and R of a tail round (half length n <= 16) over the 32 + 32 materialised generators: one lane per (side, term), 33 terms a side.
//   L = sum_{hi k} a[io] gg[k] F_G[k] + sum_{lo k} b[n+io] hh[k] F_H[k] + c_L w B      (k = blk 2n + {0, n} + io, rank = blk n + io)
//   R = sum_{lo k} a[n+io] gg[k] F_G[k] + sum_{hi k} b[io] hh[k] F_H[k] + c_R w B
// With `prev_round` != 0 the block first finishes the previous round: the first lane of each proof absorbs L, R of that round,
// draws u and inverts it (the other lanes wait), then the proof's lanes fold a, b to length 2n and update gg, hh in parallel.
// Three proofs share a 256-lane block (66 lanes each: 33 terms a side), 77 % of the lanes doing scalar multiplications instead of
// the 52 % of one proof per 128 lanes.  A term can be SPLIT over TAIL_SPLIT lanes, each taking a share of the table's pieces (a shorter
// chain per lane, the doublings paid once per share): with two shares measured 52.2 instead of 50.2 ms per batch -- the tail is bound
// by its work like everything else, not by its chain -- and TWO terms per lane on one doubling chain (-DBBP_TAIL_MERGE=2: 13 % less work,
// a 74 % longer chain) 53.8 instead of 49.3 ms: one lane per term is the optimum from both sides.
#ifndef BBP_TAIL_SPLIT
#define BBP_TAIL_SPLIT 1
#endif
#ifndef BBP_TAIL_MERGE  // 2: a lane takes TWO terms of its side and runs one doubling chain for both (17 lanes a side, seven proofs per block)
#define BBP_TAIL_MERGE 1
#endif
#ifndef BBP_TAIL_BLK
#define BBP_TAIL_BLK (BBP_TAIL_SPLIT == 1 ? 256 : 320)
#endif
constexpr int TAIL_SPLIT = BBP_TAIL_SPLIT, TAIL_MERGE = BBP_TAIL_MERGE, TAIL_LS = TAIL_MERGE == 2 ? 17 : 33 * TAIL_SPLIT, TAIL_BLK = BBP_TAIL_BLK, TAIL_LP = 2 * TAIL_LS, TAIL_PPB = TAIL_BLK / TAIL_LP;
static_assert(TAIL_PIECES % TAIL_SPLIT == 0 && TAIL_PPB >= 1 && (TAIL_MERGE == 1 || (TAIL_MERGE == 2 && TAIL_SPLIT == 1)), "tail kernel geometry");
__global__ __launch_bounds__(TAIL_BLK) void k_tail_lr(u32 B, u32 n, u32 prev_round, u32 m, const u32* __restrict__ enc, merlin_transcript* __restrict__ tr,
                                                       const sc* __restrict__ misc, sc* __restrict__ a_all, sc* __restrict__ b_all,
                                                       sc* __restrict__ g_all, sc* __restrict__ h_all, const ge* __restrict__ ftab,
                                                       const ge* __restrict__ btab, ge* __restrict__ lrpts) {
    BBP_THIN_PRIO();
    __shared__ u32 stage[GE_WORDS * TAIL_BLK];
    __shared__ u32 uu_w[TAIL_PPB * 2 * 8];
    const u32 tid = threadIdx.x;
    const u32 pj = tid / TAIL_LP, l = tid % TAIL_LP;        // proof slot in the block, lane within the proof
    const u32 p = blockIdx.x * TAIL_PPB + pj;
    const bool live = pj < TAIL_PPB && p < B;
    const u32 side = l / TAIL_LS, rem = l % TAIL_LS;        // lane `rem` of its side: share rem % TAIL_SPLIT of term j
    const u32 j = TAIL_MERGE == 2 ? 2 * rem : rem / TAIL_SPLIT, share = rem % TAIL_SPLIT;  // j: 0..15 G terms, 16..31 H terms, 32 the B term (merged: terms j and j + 1)
    const size_t po = live ? (size_t)p : 0;
    sc *a = a_all + po * 2048, *b = b_all + po * 2048, *gg = g_all + po * 2048, *hh = h_all + po * 2048;
    if (prev_round) {
        if (live && l == 0) {
            merlin_transcript t = tr[p];
            const u32* e = enc + (size_t)p * enc_stride_words(m) + 8 * (m + 8) + 16 * (prev_round - 1);
            tr_append_words(t, LBL("L"), e);
            tr_append_words(t, LBL("R"), e + 8);
            const sc u = tr_challenge_sc(t, LBL("u")), ui = sc_invert(u);
#pragma unroll
            for (int w = 0; w < 8; w++) {
                uu_w[(pj * 2) * 8 + w] = u.v[w];
                uu_w[(pj * 2 + 1) * 8 + w] = ui.v[w];
            }
            tr[p] = t;
        }
        __syncthreads();
        if (live) {
            sc u, ui;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                u.v[w] = uu_w[(pj * 2) * 8 + w];
                ui.v[w] = uu_w[(pj * 2 + 1) * 8 + w];
            }
            const u32 n2 = 2 * n;
            // work items of one proof: n2 folds of a, n2 folds of b, 32 factors gg, 32 factors hh
            for (u32 w = l; w < 2 * n2 + 2 * FOLD_CLS; w += TAIL_LP) {
                if (w < n2) {
                    st_sc(&a[w], sc_add(sc_mul(ld_sc(&a[w]), u), sc_mul(ui, ld_sc(&a[n2 + w]))));
                } else if (w < 2 * n2) {
                    const u32 i = w - n2;
                    st_sc(&b[i], sc_add(sc_mul(ld_sc(&b[i]), ui), sc_mul(u, ld_sc(&b[n2 + i]))));
                } else {
                    const u32 k = (w - 2 * n2) % FOLD_CLS;
                    const bool hi = (k & (2 * n2 - 1)) >= n2;
                    if (w - 2 * n2 < FOLD_CLS) st_sc(&gg[k], sc_mul(ld_sc(&gg[k]), hi ? u : ui));
                    else st_sc(&hh[k], sc_mul(ld_sc(&hh[k]), hi ? ui : u));
                }
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    const ge *FG = ftab + po * 2 * FOLD_CLS * TAIL_TAB, *FH = FG + (size_t)FOLD_CLS * TAIL_TAB;  // tables of F_G[32], F_H[32]
    constexpr u32 HALF = FOLD_CLS / 2;  // 16 G-terms and 16 H-terms per side
    // scalar and table of term jt of this lane's side
    auto term = [&](u32 jt, sc& s, const ge*& T) {
        if (jt < 2 * HALF) {
            const u32 rank = jt % HALF, blk = rank / n, io = rank % n;
            const u32 k_lo = blk * 2 * n + io, k_hi = k_lo + n;
            if (jt < HALF) {  // G term
                const u32 k = side == 0 ? k_hi : k_lo;
                s = sc_mul(ld_sc(&a[side == 0 ? io : n + io]), ld_sc(&gg[k]));
                T = FG + (size_t)k * TAIL_TAB;
            } else {  // H term
                const u32 k = side == 0 ? k_lo : k_hi;
                s = sc_mul(ld_sc(&b[side == 0 ? n + io : io]), ld_sc(&hh[k]));
                T = FH + (size_t)k * TAIL_TAB;
            }
        } else {
            sc c = sc_zero();
            for (u32 i = 0; i < n; i++)
                c = sc_add(c, side == 0 ? sc_mul(ld_sc(&a[i]), ld_sc(&b[n + i])) : sc_mul(ld_sc(&a[n + i]), ld_sc(&b[i])));
            s = sc_mul(c, ld_sc(&misc[(size_t)p * MS_COUNT + MS_W]));  // Q = w B
            T = btab;
        }
    };
    ge q = ge_identity();
    if (live) {
        sc s;
        const ge* T;
        term(j, s, T);
#ifndef BBP_KO_TAIL  // (timing experiment, wrong results, when defined)
        if (TAIL_MERGE == 2 && j + 1 < 2 * HALF + 1) {
            sc s2;
            const ge* T2;
            term(j + 1, s2, T2);
            q = ge_scalarmul_pieces_pair(s, T, s2, T2);
        } else {
            constexpr int PPS = TAIL_PIECES / TAIL_SPLIT;
            q = ge_scalarmul_pieces(s, T, (int)share * PPS, (int)share * PPS + PPS);
        }
#else
        q = T[0];
        q.X.v[0] += (i32)(s.v[0] & 1u);
#endif
    }
    const u32* w = reinterpret_cast<const u32*>(&q);
    for (int k = 0; k < GE_WORDS; k++) stage[k * TAIL_BLK + tid] = w[k];
    __syncthreads();
    // tree sum inside each group of TAIL_LS consecutive lanes (one group per proof and side)
    for (int d = 64; d >= 1; d >>= 1) {
        if (live && rem < (u32)d && rem + d < (u32)TAIL_LS) {
            ge o;
            u32* ow = reinterpret_cast<u32*>(&o);
            for (int k = 0; k < GE_WORDS; k++) ow[k] = stage[k * TAIL_BLK + tid + d];
            q = ge_add(q, o);
            for (int k = 0; k < GE_WORDS; k++) stage[k * TAIL_BLK + tid] = reinterpret_cast<const u32*>(&q)[k];
        }
        __syncthreads();
    }
    if (live && rem == 0) lrpts[(size_t)p * 2 + side] = q;
}

__global__ BBP_LANE_KERNEL void k_ipa_final(u32 B, u32 m, const u32* __restrict__ enc, merlin_transcript* __restrict__ tr, sc* __restrict__ misc,
                            const sc* __restrict__ a_all, const sc* __restrict__ b_all, u32 wave) {
    BBP_THIN_PRIO();
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (wave) p >>= 6;  // one transcript per wavefront, every lane in lockstep (keccak_wave.h keccak_f1600_wave)
    if (p >= B) return;
    merlin_transcript t = tr[p];
    t.wave = wave;
    const u32* e = enc + (size_t)p * enc_stride_words(m) + 8 * (m + 8) + 16 * 10;
    tr_append_words(t, LBL("L"), e);
    tr_append_words(t, LBL("R"), e + 8);
    sc u = tr_challenge_sc(t, LBL("u"));
    sc ui = sc_invert(u);
    const sc *a = a_all + (size_t)p * 2048, *b = b_all + (size_t)p * 2048;
    sc* ms = misc + (size_t)p * MS_COUNT;
    st_sc(&ms[MS_A0], sc_add(sc_mul(ld_sc(&a[0]), u), sc_mul(ui, ld_sc(&a[1]))));
    st_sc(&ms[MS_B0], sc_add(sc_mul(ld_sc(&b[0]), ui), sc_mul(u, ld_sc(&b[1]))));
    t.wave = 0;  // (memory never holds the flag: every kernel sets it for itself)
    tr[p] = t;
}

// record = R1CSProof::to_bytes (1-phase compact form, A.8) || V[0..4) || V[4..m)
__global__ BBP_LANE_KERNEL void k_assemble(u32 B, u32 m, const u32* __restrict__ enc, const sc* __restrict__ misc, u8* __restrict__ out) {
    BBP_THIN_PRIO();
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    const u32* e = enc + (size_t)p * enc_stride_words(m);
    const sc* ms = misc + (size_t)p * MS_COUNT;
    u8* o = out + (size_t)p * (BBP_R1CS_PROOF_BYTES + 32 * (size_t)m);
    *o++ = 0;  // ONE_PHASE_COMMITMENTS
    auto put = [&](const u32* w) {
        words_to_bytes32(o, w);
        o += 32;
    };
    for (u32 i = 0; i < 8; i++) put(e + 8 * (m + i));  // A_I1 A_O1 S1 T_1 T_3 T_4 T_5 T_6
    sc s;
    s = ld_sc(&ms[MS_TX]); put(s.v);
    s = ld_sc(&ms[MS_TXB]); put(s.v);
    s = ld_sc(&ms[MS_EBL]); put(s.v);
    for (u32 j = 0; j < 22; j++) put(e + 8 * (m + 8 + j));  // L_1 R_1 ... L_11 R_11
    s = ld_sc(&ms[MS_A0]); put(s.v);
    s = ld_sc(&ms[MS_B0]); put(s.v);
    for (u32 i = 0; i < m; i++) put(e + 8 * i);
}

// ---------------------------------------------------------------------------------------------------------------
// host side: circuit cache, batch buffers, orchestration
// ---------------------------------------------------------------------------------------------------------------
template <class T>
static int32_t upload(bbp_ctx* ctx, const std::vector<T>& v, T** out) {
    *out = nullptr;
    if (v.empty()) return BBP_OK;
    BBP_HIP_TRY(ctx, hipMalloc(out, v.size() * sizeof(T)));
    BBP_HIP_TRY(ctx, hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return BBP_OK;
}

int32_t circuit_get(bbp_ctx* ctx, uint32_t n_items, const CircuitDev** out) {
    auto it = ctx->circuits.find(n_items);
    if (it != ctx->circuits.end()) {
        *out = static_cast<const CircuitDev*>(it->second);
        return BBP_OK;
    }
    fault_injected("compile");
    circuit::Compiled c = circuit::compile(n_items);
    if (c.padded != 2048) {
        ctx->err = "circuit does not pad to 2048 multipliers";
        return BBP_ERR_GENS_LEN;
    }
    if (c.n_mul != 4u * 4u * (u32)BBP_MIMC_ROUNDS + 3u * n_items + 2u || c.m != 4u + n_items) {  // what witness_gates_native_lane writes
        ctx->err = "circuit synthesis and the native witness disagree about the multiplier count";
        return BBP_ERR_INTERNAL;
    }
    CircuitDev* d = new CircuitDev();
    d->n_items = n_items;
    d->m = c.m;
    d->n_mul = c.n_mul;
    d->n_cons = c.n_cons;
    d->padded = c.padded;
    d->n_cst = circuit::cst_count(n_items);
    d->n_cterms = (u32)c.c_q.size();
    int32_t rc;
    if ((rc = upload(ctx, c.w_terms, &d->w_terms)) || (rc = upload(ctx, c.w_loff, &d->w_loff)) || (rc = upload(ctx, c.w_roff, &d->w_roff)) ||
        (rc = upload(ctx, c.f_off, &d->f_off)) || (rc = upload(ctx, c.f_ent, &d->f_ent)) || (rc = upload(ctx, c.c_q, &d->c_q)) ||
        (rc = upload(ctx, c.c_cst, &d->c_cst)))
        return rc;
    std::vector<u32> ai, ao, ipa, ver;
    ai.push_back(BBP_BASE_BBLIND);
    ao.push_back(BBP_BASE_BBLIND);
    for (u32 i = 0; i < c.n_mul; i++) ai.push_back(BBP_BASE_G0 + i), ao.push_back(BBP_BASE_G0 + i);
    for (u32 i = 0; i < c.n_mul; i++) ai.push_back(BBP_BASE_H0 + i);
    const std::vector<u32> s1 = ai;
    // Multiplier inputs wired to the SAME linear combination always carry the same scalar: where three of them sit as
    // {L_i, R_i, R_i+1} or {L_i, L_i+1, R_i+1} (every MiMC round has one of each: a and a^2) the first keeps the scalar on the
    // merged base G/H sum and the other two are skipped by the sort kernel -- 1440 of A_I1's 2933 terms.
    {
        const u32 n = c.n_mul;
        auto side = [&](u32 s) {  // s < n: left input of multiplier s, else right input of multiplier s - n
            const bool right = s >= n;
            const u32 i = right ? s - n : s;
            const u32 b = right ? c.w_roff[i] : c.w_loff[i], e = right ? c.w_loff[i + 1] : c.w_roff[i];
            return std::vector<u32>(c.w_terms.begin() + b, c.w_terms.begin() + e);
        };
        auto same = [&](u32 a, u32 b, u32 d) { return side(a) == side(b) && side(a) == side(d); };
        std::vector<char> used(2 * n, 0);
        for (u32 i = 0; i + 1 < n; i++) {
            const u32 Li = i, Ri = n + i, Li1 = i + 1, Ri1 = n + i + 1;
            if (!used[Li] && !used[Ri] && !used[Ri1] && same(Li, Ri, Ri1)) {
                ai[1 + Li] = MRG_BASE0 + i;
                ai[1 + Ri] = ai[1 + Ri1] = MSM_SKIP_BASE;
                used[Li] = used[Ri] = used[Ri1] = 1;
            } else if (!used[Li] && !used[Li1] && !used[Ri1] && same(Li, Li1, Ri1)) {
                ai[1 + Li] = MRG_BASE0 + 2048 + i;
                ai[1 + Li1] = ai[1 + Ri1] = MSM_SKIP_BASE;
                used[Li] = used[Li1] = used[Ri1] = 1;
            }
        }
        d->n_ai_terms = 0;
        for (u32 v : ai) d->n_ai_terms += v != MSM_SKIP_BASE;
    }
    // IPA round r (1-based), n = 1024 >> (r-1): term rank = blk*n + io of block blk = k / 2n
    for (u32 r = 1; r <= 11; r++) {
        const u32 n = 1024u >> (r - 1);
        std::vector<u32> L(2049), R(2049);
        for (u32 rank = 0; rank < 1024; rank++) {
            u32 blk = rank / n, io = rank % n;
            u32 k_lo = blk * 2 * n + io, k_hi = k_lo + n;
            L[rank] = BBP_BASE_G0 + k_hi;
            L[1024 + rank] = BBP_BASE_H0 + k_lo;
            R[rank] = BBP_BASE_G0 + k_lo;
            R[1024 + rank] = BBP_BASE_H0 + k_hi;
        }
        L[2048] = R[2048] = BBP_BASE_B;
        // round 1: the padded multipliers' terms of L share one scalar; their first slot carries it on the range-sum base, the
        // others carry zero (k_ipa_round)
        if (r == 1 && c.n_mul < 2048) L[1024 + (c.n_mul - 1024)] = PAD_BASE0 + n_items - 1;
        ipa.insert(ipa.end(), L.begin(), L.end());
        ipa.insert(ipa.end(), R.begin(), R.end());
    }
    for (u32 i = 0; i < 2048; i++) ver.push_back(BBP_BASE_G0 + i);
    for (u32 i = 0; i < 2048; i++) ver.push_back(BBP_BASE_H0 + i);
    ver.push_back(BBP_BASE_B);
    ver.push_back(BBP_BASE_BBLIND);
    if ((rc = upload(ctx, ai, &d->idx_ai)) || (rc = upload(ctx, s1, &d->idx_s1)) || (rc = upload(ctx, ao, &d->idx_ao)) || (rc = upload(ctx, ipa, &d->idx_ipa)) ||
        (rc = upload(ctx, ver, &d->idx_ver)))
        return rc;
    ctx->circuits[n_items] = d;
    *out = d;
    return BBP_OK;
}

int32_t batch_reserve(bbp_ctx* ctx, uint32_t B, const CircuitDev& c, BatchDev& bd, int parity) {
    const size_t n1 = c.n_mul, m = c.m;
    DevBuf& buf = ctx->batch[parity];  // 0..2: the prover's buffers in rotation, 3 (VERIFY_BUF): the verifier's
    size_t off = 0;
    auto take = [&](size_t bytes_per_proof) {
        size_t o = off;
        off += ((bytes_per_proof * B + 255) / 256) * 256;
        return o;
    };
    const size_t S = sizeof(sc);
    size_t o_cst = take(c.n_cst * S), o_v = take(m * S), o_vb = take(m * S), o_ai1 = take((1 + 2 * n1) * S), o_ao1 = take((1 + n1) * S),
           o_s1 = take((1 + 2 * n1) * S), o_tr = take(sizeof(merlin_transcript)), o_rng = take(sizeof(merlin_transcript)),
           o_misc = take(MS_COUNT * S), o_zpow = take(((size_t)c.n_cons + 1) * S), o_ypow = take(2049 * S), o_yipow = take(2048 * S),
           o_wl = take(2048 * S), o_wr = take(2048 * S), o_wo = take(2048 * S), o_wv = take(m * S), o_l1 = take(n1 * S), o_r0 = take(n1 * S),
           o_r1 = take(n1 * S), o_r3 = take(n1 * S), o_a = take(2048 * S), o_b = take(2048 * S), o_g = take(2048 * S), o_h = take(2048 * S),
           o_lr = take(2 * 2049 * S), o_pts = take((m + 8) * sizeof(ge)), o_lrpts = take(2 * sizeof(ge)), o_fpts = take(2 * FOLD_CLS * sizeof(ge)),
           o_enc = take((m + 8 + 22) * 32), o_ent = take(32 * m + 32);
    int32_t rc = dev_reserve(ctx, buf, off);
    if (rc) return rc;
    u8* base = static_cast<u8*>(buf.p);
    bd.B = B;
    bd.n_items = c.n_items;
    bd.m = c.m;
    bd.n1 = c.n_mul;
    bd.n_cons = c.n_cons;
    bd.cst = (sc*)(base + o_cst); bd.v = (sc*)(base + o_v); bd.vb = (sc*)(base + o_vb); bd.ai1 = (sc*)(base + o_ai1);
    bd.ao1 = (sc*)(base + o_ao1); bd.s1 = (sc*)(base + o_s1); bd.tr = (merlin_transcript*)(base + o_tr);
    bd.rng = (merlin_transcript*)(base + o_rng); bd.misc = (sc*)(base + o_misc); bd.zpow = (sc*)(base + o_zpow);
    bd.ypow = (sc*)(base + o_ypow); bd.yipow = (sc*)(base + o_yipow); bd.wl = (sc*)(base + o_wl); bd.wr = (sc*)(base + o_wr);
    bd.wo = (sc*)(base + o_wo); bd.wv = (sc*)(base + o_wv); bd.l1 = (sc*)(base + o_l1); bd.r0 = (sc*)(base + o_r0);
    bd.r1 = (sc*)(base + o_r1); bd.r3 = (sc*)(base + o_r3); bd.a = (sc*)(base + o_a); bd.b = (sc*)(base + o_b); bd.g = (sc*)(base + o_g);
    bd.h = (sc*)(base + o_h); bd.lr = (sc*)(base + o_lr); bd.pts = (ge*)(base + o_pts); bd.lrpts = (ge*)(base + o_lrpts); bd.fpts = (ge*)(base + o_fpts);
    bd.enc = (u32*)(base + o_enc); bd.entropy = base + o_ent;
    return BBP_OK;
}

static merlin_transcript prover_prefix() {
    // Transcript::new(b"BlindBidProofGadget") (src/blindbid/mod.rs:37) then Prover::new / Verifier::new's r1cs_domain_sep (A.4)
    merlin_transcript t;
    merlin_init(t, reinterpret_cast<const uint8_t*>("BlindBidProofGadget"), 19);
    merlin_append(t, reinterpret_cast<const uint8_t*>("dom-sep"), 7, reinterpret_cast<const uint8_t*>("r1cs v1"), 7);
    return t;
}

#define LAUNCH_LDS(ctx, tag, kern, grid, block, lds, stream, ...)                        \
    do {                                                                                 \
        ScopedEvent _ev(ctx, tag, stream);                                               \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, __VA_ARGS__);     \
        BBP_HIP_TRY(ctx, hipGetLastError());                                             \
    } while (0)
// every other launch carries a token 64 bytes of LDS so that it, too, stays off the fully reserved CUs
#define LAUNCH(ctx, tag, kern, grid, block, stream, ...) LAUNCH_LDS(ctx, tag, kern, grid, block, lds_token(ctx), stream, __VA_ARGS__)

// The one-lane-per-proof kernels of the opening stage (k_open_serial: 32 wavefronts that run for tens of
// milliseconds) overlap the previous batch's heavy stage.  Sharing a SIMD with them is poison for that stage: the long-lived
// wave is the oldest on its SIMD and wins issue arbitration, the MSM workgroup next to it runs several times slower, and
// since a 1024-workgroup launch is placed in one round, every kernel then lasts as long as its slowest workgroup (measured:
// the MSM kernel 3.3 ms -> 19.5 ms while the rng kernel is resident; 107 ms -> 78 ms per batch without it).  So these launches ask for
// nearly a whole CU's LDS, which they never touch: no LDS-using workgroup (every heavy kernel that matters) can be placed
// beside them, the dispatcher routes those to the other ~240 CUs, and the serial wave has its CU to itself.
static inline u32 cdiv(u32 a, u32 b) { return (a + b - 1) / b; }

int32_t tail_btab_build(bbp_ctx* ctx) {  // called once from bbp_init: the table of B for the c w B term of the tail rounds
    BBP_HIP_TRY(ctx, hipMalloc(&ctx->btab, sizeof(ge) * TAIL_TAB));
    hipLaunchKernelGGL(k_tail_tables, dim3(1), dim3(64), 0, ctx->stream, 1u, ctx->gens + BBP_BASE_B, ctx->btab);
    BBP_HIP_TRY(ctx, hipGetLastError());
    return BBP_OK;
}

static BatchDev batch_view(const BatchDev& bd, const CircuitDev& c, u32 first);
static int32_t prove_heavy(bbp_ctx* ctx, const CircuitDev& c, const BatchDev& bd, u32 B, u8* out_dev, hipStream_t s, int slot,
                           hipEvent_t stagger, hipEvent_t out_guard);

// in_dev: B * (7*32 + N*32 + 8) ; ent_dev: B * (32 m + 32) ; out_dev: B * (1121 + 32 m).  All device pointers.
int32_t prove_batch_dev(bbp_ctx* ctx, u32 B, u32 N, const u8* in_dev, const u8* ent_dev, u8* out_dev, hipStream_t s) {
    const CircuitDev* cp;
    int32_t rc = circuit_get(ctx, N, &cp);
    if (rc) return rc;
    const CircuitDev& c = *cp;
    // Two-stage software pipeline across consecutive calls.  The OPENING stage (witness, V commitments, transcript open and
    // the 2935 strictly sequential TranscriptRng draws) is latency-bound -- one lane per proof, 16 wavefronts for 1024 proofs
    // -- so it runs on the context's side stream into the batch buffer of this call's parity, while the caller's stream is
    // still busy with the MSM-heavy stage of the PREVIOUS call (other parity).  Events order: inputs (caller stream) ->
    // opening (side) -> heavy stage (caller stream); a buffer is reused only after its previous heavy stage has finished.
    StreamGuard guard(ctx, s);
    if ((rc = guard.enter())) return rc;
    const u32 call = ctx->seq++;
    // Batches too small to fill three heavy slices are bound by the opening stage (its rng chain lasts ~40 ms whatever the
    // batch size), so their openings alternate between two streams and two of them are in flight at once; the heavy stage
    // is then cut in at most two slices so that no more than four queues are active (a fifth costs ~7 %).
    // ... unless a SLICED heavy stage (a batch of 1024 or more) is still on the device and this batch is not small: its unsliced
    // chain would run beside that batch's three slices -- a fourth and fifth busy heavy stream -- and both lose.  Two host threads
    // alternating 870- and 2202-proof batches (what the combiner's two batch threads settle into under closed-loop load): 18.8 k
    // proofs/s that way, 21.6 k with the 870 sliced and stream-ordered behind the 2202 like any large batch (tools/host_pairs.py).
    // A caller that keeps TWO OR MORE earlier prove calls in flight (device API, calls enqueued without host synchronisation) is
    // better served by whole calls in rotation than by slices of one call, whatever the batch size: three unsliced chains of three
    // calls, each with its opening stage behind it, out of phase by construction (five buffers: context.h).  1024-proof calls back
    // to back: 48.8 ms per call sliced, 45.9-46.7 in rotation (21.0 k -> 21.9-22.3 k proofs/s); 2048: 90.4 -> 87.2 ms.  With at most
    // one earlier call in flight (the host-pointer path under the combiner's two batch threads) slices win: 21.4 k against 17.9 k.
    int inflight = 0;  // earlier prove calls still on the device: a ring of per-call completion events (the buffers' own events say
                       // nothing beyond two calls while the calls are sliced: those alternate between two buffers)
    for (int k = 0; k < bbp_ctx::CALL_RING; k++)
        if (ctx->ev_call_valid[k] && hipEventQuery(ctx->ev_call[k]) == hipErrorNotReady) inflight++;
    // (sticky: entered with two earlier calls in flight, left after six calls in a row that found fewer -- a caller that
    // synchronises now and then, bench.py's barriers around its timed loop, does not fall back to slices for the calls that refill
    // its pipeline; the combiner's two batch threads never enter, and leave within six batches if a third caller once made them)
    if (inflight >= ctx->deep_from) {
        ctx->deep_mode = true;
        ctx->deep_idle_seen = 0;
    } else if (++ctx->deep_idle_seen >= 6) {
        ctx->deep_mode = false;
    }
    const bool deep = (ctx->deep_mode || ctx->force_deep) && ctx->slices > 1 && ctx->rotate_deep_max > 0 && B <= (u32)ctx->rotate_deep_max;
    bool behind_sliced = false;
    if (!deep && ctx->mixed_from > 0 && B >= (u32)ctx->mixed_from && ctx->last_sliced && ctx->last_prove_par >= 0 && ctx->ev_done_valid[ctx->last_prove_par])
        behind_sliced = hipEventQuery(ctx->ev_done[ctx->last_prove_par]) == hipErrorNotReady;
    const bool dual = !behind_sliced && (deep || B < (u32)(ctx->dual_open_below > 0 ? ctx->dual_open_below : 0));
    const int sidx = dual ? (int)(call & 1u) : 0;
    const int par = dual ? (int)(call % (u32)bbp_ctx::PROVE_BUFS) : (int)(call & 1u);  // three heavy-stage chains + two openings in flight: five buffers.  (With three, call k's opening had to wait for call k-3's heavy stage -- the chain that runs on the very stream call k's heavy stage is queued on -- and every chain stream idled for an opening stage per call: 24 % at 256 proofs per call.)
    if (dual && !ctx->side2) BBP_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->side2, hipStreamNonBlocking));
    BatchDev bd;
    if ((rc = batch_reserve(ctx, B, c, bd, par))) return rc;
    const u32 m = c.m, n1 = c.n_mul, encw = (m + 8 + 22) * 8;
    const merlin_transcript prefix = prover_prefix();
    const size_t n_draws = 3 + 2 * (size_t)n1;
    if ((rc = dev_reserve(ctx, ctx->raw[sidx], (size_t)B * n_draws * 64))) return rc;
    hipStream_t main_s = s;
    bool traced_coop = false;  // (BBP_TRACE_PROVE) the draw chain ran on a wavefront per proof
    BBP_HIP_TRY(ctx, hipEventRecord(ctx->ev_entry[par], main_s));
    {
        hipStream_t s = sidx ? ctx->side2 : ctx->side;  // opening stage
        // NOTE the opening stage does NOT wait for the caller's stream: in_dev / ent_dev must be complete when the call is
        // made (include/bbp.h).  Waiting on the caller's stream tail would serialise it behind the previous call's heavy stage.
        if (ctx->ev_done_valid[par]) BBP_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_done[par], 0));
        if (ctx->ev_prep_valid) {  // rows written by bbp_prepare_bids_dev since the last prove call
            BBP_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_prep, 0));
            ctx->ev_prep_valid = false;
        }
        LAUNCH(ctx, TAG_WITNESS, k_fill_mimc, cdiv(B * BBP_MIMC_ROUNDS, 64), 64, s, B, c.n_cst, ctx->mimc_c, bd.cst);
        // every launch asks for what serial_lds_bytes set the kernel's limit to (reservation minus the kernel's static LDS): the two
        // cannot diverge when a kernel gains a __shared__ array
        u32 hog = 0;
        if ((rc = serial_lds_bytes(ctx, (const void*)k_open_serial, &hog))) return rc;
        const u32 sblk = ctx->serial_lds > 0 ? (u32)ctx->serial_block : 64u;
        LAUNCH(ctx, TAG_WITNESS, k_witness_head, cdiv(B, 64), 64, s, B, N, c.n_cst, in_dev, bd.cst, bd.v);
        LAUNCH(ctx, TAG_TRANSCRIPT, k_load_blindings, cdiv(B * m, 64), 64, s, B, m, ent_dev, bd.vb);
        if ((rc = commit_launch(ctx, B * m, bd.v, bd.vb, m, m, m, bd.pts, m + 8, s))) return rc;
        LAUNCH(ctx, TAG_ENCODE, k_encode_strided, cdiv(B * m, 64), 64, s, B * m, m, bd.pts, m + 8, bd.enc, encw, 0u, 1u);
        // one wavefront per proof for the draw chain: always for batches up to rng_coop_below proofs; for larger ones (up to
        // rng_coop_idle_below) only when NO earlier prove call is on the device -- a lone 1024-proof call then returns after 58
        // instead of 81 ms, while a caller that keeps the device busy keeps the single-lane chain (a wavefront per proof costs a
        // full pipeline 1-2 % of its throughput: two host threads 22.2 k -> 21.8 k proofs/s)
        const bool coop = ctx->rng_coop < 0 ? B <= (u32)ctx->rng_coop_below || (inflight == 0 && B <= (u32)ctx->rng_coop_idle_below) : ctx->rng_coop != 0;
        traced_coop = coop;
        u32 interleaved = 0;  // the draws' layout in `raw` (k_open_bulk50 writes bit-interleaved halves)
        if (!coop) {
            LAUNCH_LDS(ctx, TAG_RNG, k_open_serial, 2 * cdiv(B, sblk), sblk, hog, s, B, cdiv(B, sblk), m, n1, prefix, bd.enc, ent_dev, bd.vb,
                       (u32*)ctx->raw[sidx].p, bd.tr, bd.rng, c.n_cst, c.w_terms, c.w_loff, c.w_roff, bd.cst, bd.v, bd.ai1, bd.ao1, 0u);
        } else {
            // prefix: one lane per proof, ~20 permutations (V x m, "m", the rng's rekeys, the first draw): 0.3 ms, no witness blocks
            if (B <= (u32)ctx->tr_wave_below)
                LAUNCH(ctx, TAG_RNG, k_open_serial, B, 64, s, B, B, m, n1, prefix, bd.enc, ent_dev, bd.vb, (u32*)ctx->raw[sidx].p,
                       bd.tr, bd.rng, c.n_cst, c.w_terms, c.w_loff, c.w_roff, bd.cst, bd.v, bd.ai1, bd.ao1, 2u);
            else
                LAUNCH(ctx, TAG_RNG, k_open_serial, cdiv(B, 64), 64, s, B, cdiv(B, 64), m, n1, prefix, bd.enc, ent_dev, bd.vb, (u32*)ctx->raw[sidx].p,
                       bd.tr, bd.rng, c.n_cst, c.w_terms, c.w_loff, c.w_roff, bd.cst, bd.v, bd.ai1, bd.ao1, 1u);
            // bulk: 32 lanes per proof for the draw chain + one lane per proof for the witness, in one launch of `cblk`-thread
            // workgroups that keep their CU to themselves (LDS hog): cblk / 32 proofs per rng workgroup
            // workgroup size: one wavefront (two proofs) per reserved CU is the fastest chain, but at 256 proofs that reserves 128 CUs
            // under the other in-flight calls' heavy stages; two wavefronts per CU there (measured 11.3 k -> 12.5 k proofs/s back to back),
            // four above 256 proofs (384: 12.8 k -> 14.8 k, 512: 14.6 k -> 16.5 k against the single-lane chain)
            const u32 cblk = ctx->rng_block > 0 ? (u32)ctx->rng_block : (B <= 128 ? 64u : B <= 256 ? 128u : 256u);
            if (ctx->rng_dpp) {
                // one proof per wavefront (k_open_bulk50, or BBP_RNG_DPP=1: k_open_bulk8): the same number of proofs per reserved CU needs twice the lanes
                const void* kfn = ctx->rng_dpp >= 2 ? (const void*)k_open_bulk50 : (const void*)k_open_bulk8;
                u32 hog8 = 0;
                if ((rc = serial_lds_bytes(ctx, kfn, &hog8))) return rc;
                const u32 cblk8 = 2 * cblk > 1024u ? 1024u : 2 * cblk, nb_rng = cdiv(B * 64, cblk8), nb_wit = cdiv(B, cblk8);
                if (ctx->rng_dpp >= 2) {
                    interleaved = 1;
                    LAUNCH_LDS(ctx, TAG_RNG, k_open_bulk50, nb_rng + nb_wit, cblk8, hog8, s, B, nb_rng, n1, 2 + 2 * n1, bd.rng, (u32*)ctx->raw[sidx].p, m,
                               c.n_cst, c.w_terms, c.w_loff, c.w_roff, bd.cst, bd.v, bd.ai1, bd.ao1, hog8, (u32)ctx->witness_native);
                } else {
                    LAUNCH_LDS(ctx, TAG_RNG, k_open_bulk8, nb_rng + nb_wit, cblk8, hog8, s, B, nb_rng, n1, 2 + 2 * n1, bd.rng, (u32*)ctx->raw[sidx].p, m,
                               c.n_cst, c.w_terms, c.w_loff, c.w_roff, bd.cst, bd.v, bd.ai1, bd.ao1, hog8, (u32)ctx->witness_native);
                }
            } else {
                u32 hogb = 0;
                if ((rc = serial_lds_bytes(ctx, (const void*)k_open_bulk, &hogb))) return rc;
                const u32 nb_rng = cdiv(B * 32, cblk), nb_wit = cdiv(B, cblk);
                LAUNCH_LDS(ctx, TAG_RNG, k_open_bulk, nb_rng + nb_wit, cblk, hogb, s, B, nb_rng, n1, 2 + 2 * n1, bd.rng, (u32*)ctx->raw[sidx].p, m,
                           c.n_cst, c.w_terms, c.w_loff, c.w_roff, bd.cst, bd.v, bd.ai1, bd.ao1, hogb, (u32)ctx->witness_native);
            }
        }
        LAUNCH(ctx, TAG_RNG, k_reduce_draws, cdiv((u32)(B * n_draws), 128), 128, s, B, n1, (const u32*)ctx->raw[sidx].p, bd.ai1, bd.ao1, bd.s1, interleaved);
        BBP_HIP_TRY(ctx, hipEventRecord(ctx->ev_open[par], s));
        ctx->ev_open_valid[par] = true;
    }
    // HEAVY stage.  The batch is cut into slices (default 3, BBP_SLICES) that run the same kernel sequence on separate
    // streams (slice 0 on the caller's -- a fifth concurrently active queue was measured 7 % slower): while one slice sits in a latency-bound step (the per-round transcript + scalar
    // inversion in k_ipa_round, the small encode / commit kernels) the other half's MSM keeps the CUs busy.
    u32 slices = B >= 64u * (u32)ctx->slices ? (u32)ctx->slices : (B >= 128 ? 2u : 1u);
    if (dual && slices > 2) slices = 2;
    const size_t rec = BBP_R1CS_PROOF_BYTES + 32 * (size_t)m;
    // Small batches: a heavy stage of a few hundred proofs is a chain of ~110 mostly latency-bound launches that leaves most of
    // the GPU idle (256 MSM workgroups for 1024 slots), and on the caller's stream the chains of consecutive calls run one
    // after the other: 31 ms per 256-proof call whatever else is tuned.  Such calls therefore run their heavy stage UNSLICED on
    // one of the three internal slice streams in rotation (own scratch slot each, like slices): up to three calls' chains in
    // flight; the caller's stream only waits for the result.
    const bool rotate = dual && ((ctx->rotate_below > 0 && B <= (u32)ctx->rotate_below) || (deep && B > (u32)ctx->rotate_below));
    if (ctx->trace_prove) fprintf(stderr, "prove call %u: B %u inflight %d deep %d behind_sliced %d dual %d rotate %d par %d coop %d\n", call, B, inflight, (int)deep, (int)behind_sliced, (int)dual, (int)rotate, par, (int)traced_coop);
    if (rotate) {
        const int hs = 1 + (int)(call % (u32)(bbp_ctx::MAX_SLICES - 1));
        hipStream_t ls = ctx->lane[hs];
        BBP_HIP_TRY(ctx, hipStreamWaitEvent(ls, ctx->ev_open[par], 0));
        if ((rc = prove_heavy(ctx, c, bd, B, out_dev, ls, hs, nullptr, ctx->ev_entry[par]))) return rc;
        BBP_HIP_TRY(ctx, hipEventRecord(ctx->ev_join[hs], ls));
        BBP_HIP_TRY(ctx, hipStreamWaitEvent(main_s, ctx->ev_join[hs], 0));
        slices = 0;
    }
    for (u32 i = 0; i < slices; i++) {
        const u32 first = (u32)(((u64)B * i) / slices), last = (u32)(((u64)B * (i + 1)) / slices);
        hipStream_t ls = i == 0 ? main_s : ctx->lane[i];
        BBP_HIP_TRY(ctx, hipStreamWaitEvent(ls, ctx->ev_open[par], 0));
        // stagger: slices run the same kernel sequence, so started together their latency-bound steps would coincide; each
        // slice waits for the previous slice's first MSM, which puts its serial steps under the neighbour's MSMs
        if (i && ctx->stagger_mode) BBP_HIP_TRY(ctx, hipStreamWaitEvent(ls, ctx->ev_stagger[i - 1], 0));
        if ((rc = prove_heavy(ctx, c, batch_view(bd, c, first), last - first, out_dev + rec * first, ls, (int)i,
                              i + 1 < slices ? ctx->ev_stagger[i] : nullptr, i ? ctx->ev_entry[par] : nullptr)))
            return rc;
        if (i) {
            BBP_HIP_TRY(ctx, hipEventRecord(ctx->ev_join[i], ls));
            BBP_HIP_TRY(ctx, hipStreamWaitEvent(main_s, ctx->ev_join[i], 0));
        }
    }
    BBP_HIP_TRY(ctx, hipEventRecord(ctx->ev_done[par], main_s));
    ctx->ev_done_valid[par] = true;
    ctx->last_par = par;
    ctx->last_sliced = !rotate;
    ctx->last_prove_par = par;
    BBP_HIP_TRY(ctx, hipEventRecord(ctx->ev_call[call % (u32)bbp_ctx::CALL_RING], main_s));
    ctx->ev_call_valid[call % (u32)bbp_ctx::CALL_RING] = true;
    return BBP_OK;  // ~StreamGuard records ev_last on the caller's stream
}

// pointers of `bd` advanced to proof `first` (every array is [proof][stride])
static BatchDev batch_view(const BatchDev& bd, const CircuitDev& c, u32 first) {
    BatchDev v = bd;
    const size_t f = first, n1 = c.n_mul, m = c.m;
    v.cst += f * c.n_cst; v.v += f * m; v.vb += f * m; v.ai1 += f * (1 + 2 * n1); v.ao1 += f * (1 + n1); v.s1 += f * (1 + 2 * n1);
    v.tr += f; v.rng += f; v.misc += f * MS_COUNT; v.zpow += f * ((size_t)c.n_cons + 1); v.ypow += f * 2049; v.yipow += f * 2048;
    v.wl += f * 2048; v.wr += f * 2048; v.wo += f * 2048; v.wv += f * m; v.l1 += f * n1; v.r0 += f * n1; v.r1 += f * n1; v.r3 += f * n1;
    v.a += f * 2048; v.b += f * 2048; v.g += f * 2048; v.h += f * 2048; v.lr += f * 2 * 2049; v.pts += f * (m + 8); v.lrpts += f * 2; v.fpts += f * 2 * FOLD_CLS;
    v.enc += f * (m + 8 + 22) * 8; v.entropy += f * (32 * m + 32);
    return v;
}

// everything after the opening stage for `B` proofs of the view `bd`, on stream `s`, with MSM scratch slot `slot`
static int32_t prove_heavy(bbp_ctx* ctx, const CircuitDev& c, const BatchDev& bd, u32 B, u8* out_dev, hipStream_t s, int slot,
                           hipEvent_t stagger, hipEvent_t out_guard) {
    int32_t rc;
    const u32 m = c.m, n1 = c.n_mul, encw = (m + 8 + 22) * 8;
    DevBuf& ptsbuf = slot ? ctx->slice_pts[slot] : ctx->pts;
    // A_I1, A_O1, S1 -> pts[m + 0..2] (strided output: launch per commitment with an output view)
    if ((rc = dev_reserve(ctx, ptsbuf, sizeof(ge) * (size_t)B * 3))) return rc;
    ge* tmp = static_cast<ge*>(ptsbuf.p);
    if ((rc = msm_launch(ctx, B, 1 + 2 * n1, (const u32*)bd.ai1, c.idx_ai, tmp, s, 1, slot))) return rc;
    if (stagger && ctx->stagger_mode == 1) BBP_HIP_TRY(ctx, hipEventRecord(stagger, s));
    if ((rc = msm_launch(ctx, B, 1 + n1, (const u32*)bd.ao1, c.idx_ao, tmp + B, s, 1, slot))) return rc;
    if ((rc = msm_launch(ctx, B, 1 + 2 * n1, (const u32*)bd.s1, c.idx_s1, tmp + 2 * (size_t)B, s, 1, slot))) return rc;
    // the next slice starts once this one has issued its three commitment MSMs: its own commitment MSMs then run under this
    // slice's long MSM-free stretch (encode, transcript, powers, flatten, poly, T commitments, l/r vectors)
    if (stagger && ctx->stagger_mode == 3) BBP_HIP_TRY(ctx, hipEventRecord(stagger, s));
    // the three commitments of a proof in ONE launch (they sit B points apart: tmp[k * B + p]); three launches were three
    // dependent-chain latencies (3 x 82 us for a small call)
    LAUNCH(ctx, TAG_ENCODE, k_encode_strided, cdiv(3 * B, 64), 64, s, 3 * B, 3u, tmp, 1u, bd.enc, encw, 8 * m, B);
    const u32 tw = B <= (u32)ctx->tr_wave_below ? 1u : 0u, tgrid = tw ? B : cdiv(B, 64);  // transcript kernels: one proof per wavefront for small launches
    LAUNCH(ctx, TAG_TRANSCRIPT, k_tr_yz, tgrid, 64, s, B, m, bd.enc, bd.tr, bd.misc, tw);
    {
        const u32 widest = c.n_cons + 1 > 2049u ? c.n_cons + 1 : 2049u;
        const PowersJob jz{c.n_cons + 1, (int)MS_Z, bd.zpow, c.n_cons + 1, 0u}, jy{2049u, (int)MS_Y, bd.ypow, 2049u, 0u}, jyi{2048u, (int)MS_YINV, bd.yipow, 2048u, 0u};
        LAUNCH(ctx, TAG_POLY, k_powers3, dim3(cdiv(B * cdiv(widest, 32), 64), 3), 64, s, B, bd.misc, jz, jy, jyi);
    }
    const u32 n_tgt = 3 * n1 + m;
    if (B <= (u32)ctx->ipa_wide_below)
        LAUNCH(ctx, TAG_POLY, k_flatten_split, cdiv(B * n_tgt * FLAT_L, 128), 128, s, B, n_tgt, n1, m, c.f_off, c.f_ent, bd.zpow, c.n_cons + 1, bd.wl, bd.wr,
           bd.wo, bd.wv, 2048u);
    else
        LAUNCH(ctx, TAG_POLY, k_flatten, cdiv(B * n_tgt, 128), 128, s, B, n_tgt, n1, m, c.f_off, c.f_ent, bd.zpow, c.n_cons + 1, bd.wl, bd.wr,
           bd.wo, bd.wv, 2048u);
    LAUNCH(ctx, TAG_POLY, k_poly, B, POLY_BLK, s, n1, bd.ai1, bd.ao1, bd.s1, bd.wl, bd.wr, bd.wo, 2048u, bd.ypow, bd.yipow, bd.l1, bd.r0,
           bd.r1, bd.r3, bd.misc);
    LAUNCH(ctx, TAG_TRANSCRIPT, k_tr_tblind, tgrid, 64, s, B, bd.rng, bd.misc, tw);
    if (B * 5 <= (u32)ctx->commit_split_below)
        LAUNCH(ctx, TAG_COMMIT, k_commit_T_split, cdiv(B * 5 * COMMIT_L, 64), 64, s, B, bd.misc, ctx->comb, bd.pts, m + 8, m);
    else
        LAUNCH(ctx, TAG_COMMIT, k_commit_T, cdiv(B * 5, 64), 64, s, B, bd.misc, ctx->comb, bd.pts, m + 8, m);
    LAUNCH(ctx, TAG_ENCODE, k_encode_strided, cdiv(B * 5, 64), 64, s, B * 5, 5u, bd.pts + (m + 3), m + 8, bd.enc, encw, 8 * (m + 3), 1u);
    LAUNCH(ctx, TAG_TRANSCRIPT, k_tr_ux, tgrid, 64, s, B, m, n1, bd.enc, bd.wv, bd.vb, bd.ai1, bd.ao1, bd.s1, bd.tr, bd.misc, tw);
    LAUNCH(ctx, TAG_POLY, k_lrvec, cdiv(B * 2048, 128), 128, s, B, n1, bd.l1, bd.r0, bd.r1, bd.r3, bd.ao1, bd.s1, bd.ypow, bd.yipow, bd.misc,
           bd.a, bd.b, bd.g, bd.h);
    // FOLD_ROUND (7), or 12 = never leave the fixed-base formulation (small heavy stages: context.h tail_small_below)
    const u32 tail_from = B < (u32)(ctx->tail_small_below > 0 ? ctx->tail_small_below : 0) ? 12u : (u32)ctx->tail_round;
    for (u32 r = 1; r <= 11 && r < tail_from; r++) {
        if (r > 1) LAUNCH(ctx, TAG_TRANSCRIPT, k_ipa_challenge, tgrid, 64, s, B, r - 1, m, bd.enc, bd.tr, bd.misc, tw);
        if (B <= (u32)ctx->ipa_wide_below)
            LAUNCH(ctx, TAG_IPA_SCALARS, k_ipa_round<IPA_BLK_WIDE>, B, IPA_BLK_WIDE, s, r, n1, bd.misc, bd.a, bd.b, bd.g, bd.h, bd.lr);
        else
            LAUNCH(ctx, TAG_IPA_SCALARS, k_ipa_round<IPA_BLK>, B, IPA_BLK, s, r, n1, bd.misc, bd.a, bd.b, bd.g, bd.h, bd.lr);
        if ((rc = msm_launch(ctx, 2 * B, 2049, (const u32*)bd.lr, c.idx_ipa + (size_t)(r - 1) * 2 * 2049, bd.lrpts, s, 2, slot))) return rc;
        LAUNCH(ctx, TAG_ENCODE, k_encode_strided, cdiv(2 * B, 64), 64, s, 2 * B, 2u, bd.lrpts, 2u, bd.enc, encw, 8 * (m + 8 + 2 * (r - 1)), 1u);
    }
    if (tail_from <= 11) {
        // switch to explicit folded generators: absorb round FOLD_ROUND-1, fold a, b, update g, h (no scalar rows), then one
        // composite-bucket pass materialises F_G[32], F_H[32]; the remaining rounds are small variable-base kernels
        DevBuf& vt = ctx->slice_vtab[slot];
        if ((rc = dev_reserve(ctx, vt, (size_t)B * 2 * FOLD_CLS * TAIL_TAB * sizeof(ge)))) return rc;
        ge* ftab = static_cast<ge*>(vt.p);
        LAUNCH(ctx, TAG_TRANSCRIPT, k_ipa_challenge, tgrid, 64, s, B, tail_from - 1, m, bd.enc, bd.tr, bd.misc, tw);
        LAUNCH(ctx, TAG_IPA_SCALARS, k_ipa_round<IPA_BLK>, B, IPA_BLK, s, tail_from, n1, bd.misc, bd.a, bd.b, bd.g, bd.h, (sc*)nullptr);
        if ((rc = fold_generators_launch(ctx, B, bd.g, bd.h, bd.fpts, s, slot))) return rc;
        LAUNCH(ctx, TAG_VARBASE, k_tail_init, cdiv(B * FOLD_CLS, 64), 64, s, B, bd.g, bd.h);
        LAUNCH(ctx, TAG_VARBASE, k_tail_tables, cdiv(B * 2 * FOLD_CLS, 64), 64, s, B * 2 * FOLD_CLS, bd.fpts, ftab);
        for (u32 r = tail_from; r <= 11; r++) {
            const u32 n = 1024u >> (r - 1);
            LAUNCH(ctx, TAG_VARBASE, k_tail_lr, cdiv(B, TAIL_PPB), TAIL_BLK, s, B, n, r > tail_from ? r - 1 : 0u, m, bd.enc, bd.tr, bd.misc, bd.a, bd.b, bd.g, bd.h, ftab,
                   ctx->btab, bd.lrpts);
            LAUNCH(ctx, TAG_ENCODE, k_encode_strided, cdiv(2 * B, 64), 64, s, 2 * B, 2u, bd.lrpts, 2u, bd.enc, encw, 8 * (m + 8 + 2 * (r - 1)), 1u);
        }
    }
    LAUNCH(ctx, TAG_TRANSCRIPT, k_ipa_final, tgrid, 64, s, B, m, bd.enc, bd.tr, bd.misc, bd.a, bd.b, tw);
    // a slice on an internal stream runs ahead of the caller's: its records must not land in out_dev before the work that was
    // enqueued on the caller's stream ahead of this call (a consumer of the previous call's records, say) has finished
    if (out_guard) BBP_HIP_TRY(ctx, hipStreamWaitEvent(s, out_guard, 0));
    LAUNCH(ctx, TAG_TRANSCRIPT, k_assemble, cdiv(B, 64), 64, s, B, m, bd.enc, bd.misc, out_dev);
    return BBP_OK;
}

// debug / parity hook: copy a proof's challenge block (MS_COUNT scalars) to the host
int32_t debug_read_misc(bbp_ctx* ctx, u32 B, u32 N, u32 proof, uint8_t* out) {
    const CircuitDev* cp;
    int32_t rc = circuit_get(ctx, N, &cp);
    if (rc) return rc;
    BatchDev bd;
    if ((rc = batch_reserve(ctx, B, *cp, bd, ctx->last_par))) return rc;
    BBP_HIP_TRY(ctx, hipMemcpy(out, bd.misc + (size_t)proof * MS_COUNT, MS_COUNT * 32, hipMemcpyDeviceToHost));
    return BBP_OK;
}

}  // namespace bbp

#include "verifier.inc"
