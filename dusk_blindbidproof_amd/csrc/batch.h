// Device-side layout of one batch of proofs (prover and verifier share it) and small load/store helpers.
#pragma once
#include "circuit.h"
#include "context.h"
#include "keccak.h"

namespace bbp {

// slots of the per-proof `misc` scalar block
enum MiscSlot : int {
    MS_Y = 0, MS_Z, MS_U, MS_X, MS_W, MS_YINV, MS_T1, MS_T2, MS_T3, MS_T4, MS_T5, MS_T6, MS_TB1, MS_TB2, MS_TB3, MS_TB4, MS_TB5,
    MS_TB6, MS_TX, MS_TXB, MS_EBL, MS_UJ, MS_UJI, MS_A0, MS_B0, MS_R, MS_ALLINV, MS_WC, MS_DELTA, MS_RHO /* aggregated verification: the proof's random weight */, MS_COUNT = 32
};

struct CircuitDev {  // compiled circuit tables resident on the device (one per bid-list length N)
    u32 n_items = 0, m = 0, n_mul = 0, n_cons = 0, padded = 0, n_cst = 0;
    u32 *w_terms = nullptr, *w_loff = nullptr, *w_roff = nullptr, *f_off = nullptr, *f_ent = nullptr, *c_q = nullptr, *c_cst = nullptr;
    u32 n_cterms = 0;
    u32* idx_ai = nullptr;   // base indices of A_I1's terms: B_blinding, G[0..n1), H[0..n1), equal-valued inputs merged (circuit_get)
    u32* idx_s1 = nullptr;   // the same layout unmerged, for S1 (independent random scalars)
    u32 n_ai_terms = 0;      // terms of A_I1 that are not skipped
    u32* idx_ao = nullptr;   // B_blinding, G[0..n1)
    u32* idx_ipa = nullptr;  // [11 rounds][2 (L,R)][2049]
    u32* idx_ver = nullptr;  // verifier fixed part: G[0..2048), H[0..2048), B, B_blinding
};

// All per-proof arrays of a batch; every pointer is [B][stride] with the stride noted.
struct BatchDev {
    u32 B = 0;
    u32 n_items = 0, m = 0, n1 = 0, n_cons = 0;
    sc* cst;      // [n_cst]
    sc* v;        // [m]
    sc* vb;       // [m]
    sc* ai1;      // [1 + 2 n1]   i_blinding, a_L, a_R
    sc* ao1;      // [1 + n1]     o_blinding, a_O
    sc* s1;       // [1 + 2 n1]   s_blinding, s_L, s_R
    merlin_transcript* tr;   // [1]
    merlin_transcript* rng;  // [1]
    sc* misc;     // [MS_COUNT]
    sc* zpow;     // [n_cons + 1]   z^0 .. z^n_cons
    sc* ypow;     // [2049]
    sc* yipow;    // [2048]
    sc* wl;       // [n1] (prover) / [2048] zero padded (verifier)
    sc* wr;
    sc* wo;
    sc* wv;       // [m]
    sc* l1;       // [n1]
    sc* r0;
    sc* r1;
    sc* r3;
    sc* a;        // [2048] IPA vectors, folded in place
    sc* b;
    sc* g;        // [2048] per-generator accumulated factors
    sc* h;
    sc* lr;       // [2][2049] scalars of L and R of the current round
    ge* pts;      // [m + 8] V points, then A_I1, A_O1, S1, T_1, T_3, T_4, T_5, T_6
    ge* lrpts;    // [2]
    ge* fpts;     // [2][FOLD_CLS] explicit folded generators of the IPA tail: F_G then F_H
    u32* enc;     // [(m + 8 + 22) * 8] encodings: V[m], A_I1, A_O1, S1, T1,T3,T4,T5,T6, (L_j, R_j) x 11
    u8* entropy;  // [32 (4+N) + 32]
};

__device__ __forceinline__ sc ld_sc(const sc* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    return BBP_SC_LIT(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}
__device__ __forceinline__ void st_sc(sc* p, const sc& s) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(s.v[0], s.v[1], s.v[2], s.v[3]);
    q[1] = make_uint4(s.v[4], s.v[5], s.v[6], s.v[7]);
}

// prover.hip / verifier.hip
int32_t circuit_get(bbp_ctx* ctx, uint32_t n_items, const CircuitDev** out);
int32_t batch_reserve(bbp_ctx* ctx, uint32_t B, const CircuitDev& c, BatchDev& bd, int parity);
int32_t commit_launch(bbp_ctx* ctx, u32 count, const sc* values, const sc* blindings, u32 stride_v, u32 stride_b, u32 per_proof,
                      ge* out, u32 out_stride, hipStream_t s);

}  // namespace bbp
