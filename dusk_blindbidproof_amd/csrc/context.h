// Engine context shared by the translation units of libbbp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bbp.h"
#include "point.h"
#include "scalar.h"

namespace bbp {

// ---- MSM geometry (see DESIGN.md "K1") ---------------------------------------------------------------
constexpr int MSM_NAF = 12;               // scalar recoding: width-12 NAF, odd digits |d| < 2048
constexpr int MSM_POS = 256;              // table rows per generator: 2^b * P for every bit position b
constexpr int MSM_W = 22;                 // most digits one scalar can have (positions >= 12 apart, last <= 253)
constexpr int MSM_K = 1 << (MSM_NAF - 2); // 1024 buckets: bucket k holds the digit magnitude 2k - 1
#ifndef BBP_MSM_T
#define BBP_MSM_T 128
#endif
constexpr int MSM_T = BBP_MSM_T;          // lanes of the accumulate workgroup: 128 (2 wavefronts; -DBBP_MSM_T=64 for experiments)
#ifndef BBP_ACC_T
#define BBP_ACC_T 256
#endif
// lanes of k_msm_acc = equal chunks the sorted entries are cut into (k_msm_fold keeps MSM_T lanes).  Since the fold moved out the
// accumulate kernel is nothing but the chunk loop, and 256 lanes (four wavefronts) per MSM fill the machine better than 128: a
// third-of-a-batch launch is 682 MSMs = 2728 wavefronts for 2048 resident slots instead of 1364 (measured 18.2-18.4 k -> 19.0-19.6 k
// proofs/s; 384 lanes 17.0 k, 512 lanes 18.2 k; more than two waves per SIMD with 256 lanes 18.5-18.8 k)
constexpr int ACC_T = BBP_ACC_T;
static_assert(ACC_T % MSM_T == 0, "the fold kernel deals the accumulate kernel's chunks to its lanes");
constexpr int MSM_G = MSM_K / MSM_T;      // consecutive buckets per lane (8)
constexpr int MSM_LOG_G = MSM_T == 128 ? 3 : 4;
static_assert(MSM_T == 128 || MSM_T == 64, "the cross-lane fold is written for one or two wavefronts");
// IPA tail (prover.hip): from round FOLD_ROUND the folded generators are explicit points; they are materialised by a
// composite-bucket Pippenger pass over the same row table (msm.hip, the <1> instances of k_msm_sort / k_msm_acc)
constexpr int FOLD_ROUND = 7;              // first tail round: vectors of length 32 (halves of 16)
constexpr int FOLD_CLS = 2048 >> (FOLD_ROUND - 1);  // 32 folded generators per side
constexpr int FOLD_NAF = 9;                // width-9 NAF: odd digits |d| < 256
constexpr int FOLD_W = 29;                 // most digits per scalar
constexpr int FOLD_M = 128;                // buckets per class
constexpr int FOLD_K = FOLD_CLS * FOLD_M;  // 4096 composite buckets
// Geometry of SPLIT MSMs (round 4; small batches: an MSM is cut into sub-MSMs of a few hundred terms, one workgroup each, so that a
// launch of a handful of MSMs fills the GPU): width-9 NAF digits into 128 buckets.  A sub-MSM of 128 terms has no use for 1024 buckets
// (2.5 entries each), and the bucket FOLD -- 38 dependent point additions over 1024 buckets on 128 lanes, 160 us -- is the longest
// link of a single proof's heavy chain; over 128 buckets it is 21.  More additions per term (25.6 against 19.85) in an accumulate
// launch that takes 40 us.
constexpr int SMALL_NAF = 9;
constexpr int SMALL_K = 1 << (SMALL_NAF - 2);  // 128
constexpr int SMALL_W = FOLD_W;                // most digits one scalar can have at width 9
// Extra table bases (internal, after the 4098 public ones): PAD_BASE0 + N - 1 = sum_{k = 418 + 3N}^{1023} H[k], the generators
// that the first IPA round multiplies by ONE common scalar (the zero-padded multipliers n1 = 1442 + 3N .. 2047 contribute
// b[i] h[i - 1024] = -y^1024 for every i): 582 table-row walks become one (prover.hip k_ipa_round, circuit_get).
constexpr int PAD_BASES = 202;                             // one per list length N = 1..202
constexpr unsigned PAD_BASE0 = BBP_NUM_BASES;
// Merged bases for multiplier inputs that always carry the same value (the MiMC rounds wire a to L_i, R_i, R_{i+1} and a^2 to
// L_{i+1}, L_{i+2}, R_{i+2}: circuit_get): MRG1(i) = G[i] + H[i] + H[i+1], MRG2(i) = G[i] + G[i+1] + H[i+1], i = 0..2046 --
// A_I1 then needs one term where it had three.
constexpr unsigned MRG_BASE0 = PAD_BASE0 + PAD_BASES;
constexpr int MRG_BASES = 2 * 2048;
constexpr unsigned MSM_SKIP_BASE = 0xffffffffu;            // index-list entry: this term's scalar rides on another term's merged base
constexpr unsigned TAB_BASES = BBP_NUM_BASES + PAD_BASES + MRG_BASES;  // bases with rows in ptable
constexpr int GE_WORDS = sizeof(ge) / 4;  // 40: a point in registers / LDS / scratch

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

}  // namespace bbp

struct bbp_ctx {
    // One lock per context: every extern "C" entry point that touches the context holds it for the whole call (api_guard below), so
    // the reference's per-connection worker threads (src/main.rs:55, src/futures/main.rs:46-56) may share ONE context.  Recursive:
    // bbp_prove -> bbp_prove_batch, bbp_msm_batch -> bbp_msm_batch_dev nest.  Concurrent single-proof callers do not queue on this
    // lock one by one: submit.hip coalesces them into one batch call (group commit).
    std::recursive_mutex mu;
    void* combiner = nullptr;  // bbp::Combiner* (submit.hip): coalesces concurrent bbp_prove / bbp_verify calls into batch calls
    // Device pool (pool.cpp, bbp_pool_init / bbp_init_all): a context with `members` owns no device state of its own -- it is the
    // handle the reference's prove() / verify() callers share when the node has several GPUs: ONE combiner deals their batches to
    // the members (one ordinary context per GPU), the host-pointer batch calls block-split over them.  A member knows its pool.
    std::vector<bbp_ctx*> members;
    void* pool_workers = nullptr;  // bbp::PoolWorkers* (pool.cpp): the members' persistent host threads for the block-split batch calls
    bbp_ctx* owner = nullptr;
    uint32_t member_index = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;            // opening stage of the prover pipeline (prover.hip)
    hipStream_t copy = nullptr;            // the caller's ingest stream (bbp_context_copy_stream): never used by the engine
    hipStream_t side2 = nullptr;           // second opening stream: batches too small to fill three heavy slices alternate between the two
    int varbase_lanes = 65536;             // lanes the verifier's variable-base kernel is launched with (BBP_VARBASE_LANES): ~1 wave per SIMD
    int rotate_below = 1023;               // batches of at most this many proofs run their heavy stage unsliced on a rotating internal stream (BBP_ROTATE_BELOW, 0 = never)
    int rotate_deep_max = 4096;            // calls of up to this many proofs issued while deep_from or more earlier prove calls are still in flight take the rotating path too (BBP_ROTATE_DEEP_MAX, 0 = never; never with BBP_SLICES=1)
    bool deep_mode = false, force_deep = false;  // (state of that rule; force_deep: bbp_reserve warming the rotating path's buffers)
    int deep_idle_seen = 0;
    std::vector<hipStream_t> spare_streams;  // (experiment BBP_VL_SKIP: placeholders in the hardware-queue round-robin)
    bool trace_prove = false;              // BBP_TRACE_PROVE: one stderr line per prove call with the schedule it took
    static constexpr int CALL_RING = 8;
    hipEvent_t ev_call[CALL_RING] = {};    // completion of the last CALL_RING prove calls (how many are still in flight)
    bool ev_call_valid[CALL_RING] = {};
    int deep_from = 3;                     // earlier prove calls in flight that switch a caller to the rotating path (BBP_ROTATE_DEEP_FROM)
    int mixed_from = 512;                  // ... but a batch of at least this many proofs that arrives while a sliced heavy stage is in flight takes the sliced path too (BBP_ROTATE_MIXED_FROM, 0 = never)
    int last_prove_par = -1;               // buffer of the last prove call (last_par is also set by the verifier lanes)
    bool last_sliced = false;              // the last prove call's heavy stage ran as slices on the caller's stream + lanes
    int dual_open_below = 1024;            // batches smaller than this open on alternating streams (BBP_DUAL_OPEN_BELOW, 0 = never)
    static constexpr int MAX_SLICES = 4;   // heavy-stage slices of one batch, one stream each (slice 0 = caller's stream)
    hipStream_t lane[MAX_SLICES] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_join[MAX_SLICES] = {nullptr, nullptr, nullptr, nullptr}, ev_stagger[MAX_SLICES] = {nullptr, nullptr, nullptr, nullptr};
    int slices = 3;
    int sort_staged = 3;       // bit 0: generic MSMs, bit 1: the generator-fold pass sort with the scatter staged through LDS (msm.hip k_msm_sort_staged; BBP_SORT_STAGED)
    bool sort_lds_attr = false; // k_msm_sort_staged's dynamic-LDS limit has been raised on this context's device
    bool sort_lds_attr1 = false;  // ... and that of the generator-fold instance
    int debug_corrupt = 0;      // bbp_debug_corrupt_scratch: poison the next MSM launch's sorted scratch (tests)
    int fold_half_from = 512;  // MSM launches with at least this many MSMs fold on half a wavefront per MSM (msm.hip k_msm_fold_half; BBP_FOLD_HALF_FROM)
    int tail_small_below = 65;    // heavy stages of fewer proofs than this keep ALL eleven IPA rounds on the fixed-base MSM kernels: a short chain of
                                  // launches that fill the GPU by splitting beats the generator fold + table + tail kernels when latency is what counts
                                  // (one proof 24.7 -> 23.5 ms, 8 proofs 26.2 -> 23.8, 64 proofs 30.2 -> 29.3; 256 proofs 38.9 -> 40.9: not there).  BBP_TAIL_SMALL_BELOW
    int tail_round = bbp::FOLD_ROUND;  // first IPA round run on explicit folded generators (BBP_TAIL_ROUND=12 disables)
    int serial_lds = 160 * 1024;  // LDS the one-lane-per-proof opening kernels reserve to keep their CU to themselves (BBP_SERIAL_LDS, 0 = off)
    // TranscriptRng draw chain on 25 lanes per sponge (k_open_bulk) instead of one lane per proof.  The cooperative form is bound
    // by the CU's LDS crossbar (18 ds_bpermute per round): with ONE wavefront (two proofs) per CU a permutation takes a third of
    // the single-lane time -- a single proof 44 -> 25 ms, 256 proofs 55 -> 41 ms -- but 1024 proofs would need 512 CUs' worth of
    // crossbar, and the single-lane chain hides under the previous batch's MSM stage anyway.  Auto (-1): cooperative for batches
    // of at most rng_coop_below proofs (768; with four wavefronts per CU a 512-proof chain takes ~15 ms on 64 CUs).  BBP_RNG_COOP=0 / 1 forces, BBP_RNG_COOP_BELOW, BBP_RNG_BLOCK tune.
    int rng_coop = -1;
    int rng_coop_below = 768;
    int tr_wave_below = 32;          // launches of at most this many proofs run the transcript kernels with one proof per wavefront and the permutations spread over its lanes (BBP_TR_WAVE_BELOW, 0 = never)
    int ipa_wide_below = 32;         // launches of at most this many proofs run k_ipa_round on 1024 lanes per proof instead of 256 (BBP_IPA_WIDE_BELOW, 0 = never)
    int commit_split_below = 1024;   // Pedersen-commitment launches of at most this many commitments put each on eight lanes (prover.hip k_commit_split; BBP_COMMIT_SPLIT_BELOW, 0 = never)
    int witness_native = 1;          // the witness blocks of the cooperative opening launches write the gates from the gadget wiring itself instead of interpreting the compiled program (BBP_WITNESS_NATIVE=0)
    int msm_small = 1;               // split MSMs (launches of fewer than 128) use width-9 digits and 128 buckets (msm.hip msm_geom<2>); BBP_MSM_SMALL=0: 1024 like the others
    int rng_coop_idle_below = 2300;  // ... and up to this many proofs when the call finds the device without an earlier prove call (BBP_RNG_COOP_IDLE_BELOW; 0: never)
    int rng_dpp = 2;              // cooperative chain on one wavefront per proof: 2 = one half-word per lane, bit-interleaved (k_open_bulk50, keccak_wave.h); BBP_RNG_DPP=1: one word per lane, DPP / permlane-swap theta (k_open_bulk8); 0: the 25-lane ds_bpermute form (k_open_bulk)
    int rng_block = 0;            // threads per workgroup of k_open_bulk (BBP_RNG_BLOCK); 0 = by batch size: 64 (one wavefront = two proofs per reserved CU) up to 128 proofs, 128 up to 256, 256 above
    int serial_block = 64;        // threads per workgroup of those kernels: 256 = one serial wave per SIMD of the reserved CU (BBP_SERIAL_BLOCK)
    std::map<const void*, int> serial_attr;
    int stagger_mode = 0;  // 0: slices start together, 1: next slice starts after this slice's first MSM, 3: after its third (BBP_STAGGER)
#ifndef BBP_PROVE_BUFS
#define BBP_PROVE_BUFS 5
#endif
    static constexpr int PROVE_BUFS = BBP_PROVE_BUFS;  // prover batch buffers in rotation (call k of the small-batch path uses buffer k % PROVE_BUFS; large batches alternate between 0 and 1)
    hipEvent_t ev_open[PROVE_BUFS] = {}, ev_done[PROVE_BUFS] = {};
    hipEvent_t ev_entry[PROVE_BUFS] = {};  // caller's stream at entry of a prove call: out_dev is not written before it
    bool ev_done_valid[PROVE_BUFS] = {}, ev_open_valid[PROVE_BUFS] = {};
    int verify_overlap = 1;                              // BBP_VERIFY_OVERLAP: lane 0's variable-base kernels on a side stream (no measurable difference with four lanes)
    // BBP_VERIFY_SERIAL_ACC=1: the verifier lanes' MSM accumulate launches are chained by events so that no two of them are
    // co-resident (each then runs beside the other lanes' thin front-end kernels only).  Helps two lanes that share hardware queues
    // with other streams (6.5 -> 6.3 ms per 1024); with four lanes on queues of their own free-running is faster (5.2 vs 6.05 ms): off.
    int verify_serial_acc = 0;
    static constexpr int VACC_RING = 4;
    hipEvent_t ev_vacc[VACC_RING] = {};
    uint32_t vacc_seq = 0;
    bool vacc_valid = false;
    uint32_t verify_group = 0;  // BBP_VERIFY_AGGREGATE=G: bbp_verify / bbp_verify_batch (host API, hence the UDS server) check proofs in
                                // groups of G with per-proof fallback -- same statuses, 2-3x the rate; 0 = one MSM per proof like the reference
    uint32_t seq_at_last_verify = 0;                     // prover call counter seen by the last verification (interleaving test)
    hipEvent_t ev_prep = nullptr;  // end of the last bbp_prepare_bids_dev: the next prove call's opening stage waits for it
    bool ev_prep_valid = false;
    uint32_t seq = 0;
    int last_par = 0;
    // calls on one context share scratch buffers: a call issued on a different caller stream than the previous one is ordered
    // behind it (stream_guard_enter / stream_guard_leave)
    // two families with disjoint scratch: 0 = prover, MSM hook, witness, setup read-backs; 1 = verifier (its own batch buffer,
    // misc scratch and MSM scratch slot VERIFY_SLOT) -- a verification issued on another stream than a prove call is NOT ordered
    // behind it and overlaps its heavy stage on the device
#ifndef BBP_VLANES
#define BBP_VLANES 4
#endif
    static constexpr int VLANES = BBP_VLANES;    // four lanes -- with enough hardware queues (GPU_MAX_HW_QUEUES >= 16), see below
    static constexpr int FAMILIES = 1 + VLANES;  // prover | one per verifier lane
    hipStream_t last_stream[FAMILIES] = {};
    hipEvent_t ev_last[FAMILIES] = {};
    bool ev_last_valid[FAMILIES] = {};
    // Two verifier LANES, each with everything a verification call touches (batch buffer, scratch, MSM scratch slot, aggregation
    // buffers, a stream): two calls on the two lanes share nothing, so the front end of one (parse, transcripts, powers, flatten,
    // scalars: latency-bound) runs under the MSM of the other.  The host-pointer API alternates lanes with its staging slots;
    // device-API callers pick a lane by passing that lane's stream (bbp_context_verify_stream), any other stream is lane 0.
    // FOUR lanes (round 3; two before).  A 1024-proof call is a chain of ~10 ms of kernels of which only the 4 ms MSM accumulate fills
    // the machine, so several chains must be in flight.  A third lane first measured 8-10 % SLOWER than two -- because its stream
    // shared a HARDWARE QUEUE with another lane's (GPU_MAX_HW_QUEUES was 8, the context creates 9+ streams): with 16 queues, 1024
    // verifications per call take 6.3 ms on two lanes, 5.45 on three, 5.17 on four, 5.45 / 5.25 on five / six.
    struct VLane {
        hipStream_t stream = nullptr;
        bbp::DevBuf misc, agg, agg_io;
        void *agg_vs = nullptr, *agg_varsum = nullptr;  // weighted generator scalars [B][4098] / per-proof variable-base sums of the last group pass
        int32_t* agg_gstatus = nullptr;                 // per-group verdicts of that pass (inside agg)
        bbp::u32* agg_count = nullptr;                  // [2] device counters: proofs of the current call on the per-proof path / running total
        hipEvent_t ev_vfork = nullptr, ev_vjoin = nullptr;  // variable-base kernel on lane[1] beside the generator MSM (lane 0 only)
    };
    VLane vl[VLANES];
    std::string err;
    bbp::u32* health = nullptr;  // device word, bit 0: an MSM table gather had to be clamped (corrupted scratch) -- bbp_check_health
    // resident tables
    bbp::ge* gens = nullptr;           // [TAB_BASES] extended points: B_blinding, G[2048], H[2048], B, then the PAD_BASES range sums and the MRG_BASES merged bases
    bbp::niels_row* ptable = nullptr;      // [TAB_BASES * MSM_POS] affine cached 2^b * P_i, 128-byte limb rows (275 MB)
    bbp::ge* btab = nullptr;               // [32] m * 2^(64 k) * B, m = 1..8, k = 0..3 (prover.hip tail rounds)
    bbp::niels_packed* comb = nullptr;     // [2][64][8] radix-16 comb for B and B_blinding (small commits)
    bbp::sc* mimc_c = nullptr;         // [90]
    uint8_t gens_enc_host_valid = 0;
    std::vector<uint8_t> mimc_host;    // 90 * 32
    // grow-only scratch
    bbp::DevBuf scal, idx, sorted, pts, enc, batch[PROVE_BUFS + VLANES], io_in, io_out, io_ent, raw[2];  // batch[3 + lane]: the verifier lanes'; raw[i]: draw buffer of opening stream i
    static constexpr int VERIFY_SLOT = MAX_SLICES;  // MSM scratch slots of the verifier lanes: VERIFY_SLOT + lane
    bbp::DevBuf slice_sorted[MAX_SLICES + VLANES], slice_pts[MAX_SLICES + VLANES], slice_fold[MAX_SLICES], slice_vtab[MAX_SLICES];  // per-slice MSM scratch (slice 0 uses sorted / pts)
    // Host-pointer batch calls stage through one of three slots (device in / entropy / out + a pinned host mirror of the results):
    // a call holds the context lock only while it ENQUEUES; it waits for its results on the slot's event with the lock released,
    // so other host threads can enqueue the next batches meanwhile and the engine's cross-call pipeline (opening stage of call
    // k+1 under the MSM stage of call k) also works for bbp_prove_batch / bbp_prove / the UDS server (capi_prove.hip).
    struct IoSlot {
        bbp::DevBuf in, ent, out;
        uint32_t* h_flag = nullptr;              // pinned: the context's health word as read back with this slot's results
        void *h_out = nullptr, *h_in = nullptr;  // pinned mirrors: results / inputs (a copy from PAGEABLE memory waits for the whole
        size_t h_cap = 0, h_in_cap = 0;          // device to go idle -- measured 87 ms behind a running batch -- a pinned one does not)
        hipEvent_t ev = nullptr, ev_in = nullptr;
        bool busy = false;
    };
    static constexpr int IO_SLOTS = 3;           // prove calls rotate over slots 0..2
    static constexpr int IO_VSLOTS = VLANES;     // verify calls over their own slots 3..4 (= verifier lane 0 / 1): a verification never
    IoSlot io[IO_SLOTS + IO_VSLOTS];             // waits for a slot that a 100 ms prove call is holding
    std::mutex io_mu;
    std::condition_variable io_cv;
    uint32_t io_next = 0, io_vnext = 0;
    std::map<uint32_t, void*> circuits;  // N -> CircuitDev* (compiled blind-bid circuit tables on the device)
    std::map<uint64_t, bbp::u32*> layout_idx;  // (layout << 32 | n_terms) -> device base-index list of bbp_msm_batch (capi_msm.hip)
    std::vector<float> timings;
    uint64_t scratch_allocs = 0;  // dev_reserve: how many times a grow-only scratch buffer was (re)allocated since bbp_init ...
    size_t scratch_bytes = 0;     // ... and what they hold now (bbp_describe)
    // optional per-kernel HIP-event timing (bbp_set_profiling): (tag, start, stop) on the launch stream
    bool profile = false;
    struct Ev {
        int tag;
        hipEvent_t a, b;
    };
    std::vector<Ev> events;
};

namespace bbp {

// BBP_DEBUG_DEVICE_CHECK=1 (on in every -m gpu test): before EVERY HIP call the engine makes on behalf of a context -- launches
// (their hipGetLastError), event records / waits, allocations, copies -- the calling thread's current device must be the context's.
// A process with one context per GPU (the pool, bbp-uds-server --devices, one combiner thread hopping between members) depends on
// every entry point, every combiner batch thread and every pool worker having called hipSetDevice first; on a one-GPU box a missing
// call is invisible, on the first 8-GPU node it is a stream / event of device A used with device B current.  With the check on such a
// call fails with BBP_ERR_DEVICE and names the call site instead.  (hipSetDevice itself is exempt: it is the call that establishes
// the state.)  Cost when on: one thread-local read per HIP call; when off: one predictable branch.
inline bool device_check_on() {
    static const bool on = [] { const char* e = getenv("BBP_DEBUG_DEVICE_CHECK"); return e && atoi(e) != 0; }();
    return on;
}
inline bool device_affinity_ok(bbp_ctx* ctx, const char* what) {
    if (!device_check_on() || ctx->device < 0) return true;
    if (!strncmp(what, "hipSetDevice", 12) || !strncmp(what, "hipGetDeviceCount", 17)) return true;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == ctx->device) return true;
    ctx->err = "device affinity violated: current device " + std::to_string(cur) + ", context lives on device " + std::to_string(ctx->device) + ", at " + what;
    fprintf(stderr, "[bbp] %s\n", ctx->err.c_str());
    return false;
}

#define BBP_HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                               \
        if (!bbp::device_affinity_ok(ctx, #expr)) return BBP_ERR_DEVICE;                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                            \
            return BBP_ERR_DEVICE;                                                                     \
        }                                                                                              \
    } while (0)

inline int32_t dev_reserve(bbp_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.cap >= bytes) return BBP_OK;
    if (b.p) {
        BBP_HIP_TRY(ctx, hipFree(b.p));
        ctx->scratch_bytes -= b.cap;
    }
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + (bytes >> 3) + 4096;
    BBP_HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    ctx->scratch_bytes += want;
    ctx->scratch_allocs++;  // bbp_describe reports both: after bbp_reserve the count must stand still (tests/test_gpu_boundary.py)
    static const bool trace = getenv("BBP_TRACE_ALLOC") != nullptr;  // which buffer grew: its offset inside the context names the field
    if (trace)
        fprintf(stderr, "[bbp alloc] device %d: buffer at context offset %zu grows to %zu bytes (allocation %llu)\n", ctx->device,
                (size_t)(reinterpret_cast<const char*>(&b) - reinterpret_cast<const char*>(ctx)), want, (unsigned long long)ctx->scratch_allocs);
    return BBP_OK;
}

enum { TAG_MSM = 1, TAG_ENCODE = 2, TAG_WITNESS = 3, TAG_RNG = 4, TAG_POLY = 5, TAG_IPA_SCALARS = 6, TAG_COMMIT = 7,
       TAG_TRANSCRIPT = 8, TAG_VERIFY_SCALARS = 9, TAG_VARBASE = 10, TAG_MSM_SORT = 11, TAG_MSM_FOLD = 12 };

struct ScopedEvent {  // records start now, stop at scope exit, when profiling is on
    bbp_ctx* ctx;
    hipStream_t s;
    int idx = -1;
    ScopedEvent(bbp_ctx* c, int tag, hipStream_t st) : ctx(c), s(st) {
        if (!c->profile) return;
        bbp_ctx::Ev e;
        e.tag = tag;
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
        (void)hipEventRecord(e.a, st);
        c->events.push_back(e);
        idx = (int)c->events.size() - 1;
    }
    ~ScopedEvent() {
        if (idx >= 0) (void)hipEventRecord(ctx->events[idx].b, s);
    }
};

// see prover.hip "serial_lds_bytes": kernels that must not share a CU with the long-lived serial waves ask for a few bytes of LDS
inline unsigned lds_token(const bbp_ctx* ctx) { return ctx->serial_lds >= 160 * 1024 ? 64u : 0u; }

// One-lane-per-item serial kernels that run beside the MSM stage ask for (nearly) a whole CU's LDS so that nothing else is placed
// on their CU (prover.hip "Serial waves get their own CUs"): raise the kernel's dynamic-LDS limit once per kernel.
inline int32_t serial_lds_bytes(bbp_ctx* ctx, const void* kernel, unsigned* dyn_bytes = nullptr) {
    if (dyn_bytes) *dyn_bytes = 0;
    if (ctx->serial_lds <= 0) return BBP_OK;
    if (!ctx->serial_attr.count(kernel)) {
        // the reservation is dynamic LDS on top of whatever the kernel declares statically: together they must fit the CU's 160 KB
        hipFuncAttributes fa;
        BBP_HIP_TRY(ctx, hipFuncGetAttributes(&fa, kernel));
        int dyn = ctx->serial_lds - (int)fa.sharedSizeBytes;
        if (dyn < 0) dyn = 0;
        BBP_HIP_TRY(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
        ctx->serial_attr[kernel] = dyn;
    }
    if (dyn_bytes) *dyn_bytes = (unsigned)ctx->serial_attr[kernel];
    return BBP_OK;
}

// Calls on one context share scratch: a call issued on a different caller stream than the previous one is ordered behind it.
// RAII so that every exit path -- error returns included -- leaves ev_last recorded on the stream that may have work queued.
struct StreamGuard {
    bbp_ctx* ctx;
    hipStream_t s;
    int fam;  // 0: prover family, 1 + lane: a verifier lane (context.h: disjoint scratch, no ordering between families)
    bool entered = false;
    StreamGuard(bbp_ctx* c, hipStream_t st, int family = 0) : ctx(c), s(st), fam(family) {}
    int32_t enter() {
        if (ctx->ev_last_valid[fam] && ctx->last_stream[fam] != s) BBP_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_last[fam], 0));
        entered = true;
        return BBP_OK;
    }
    ~StreamGuard() {
        if (!entered) return;
        if (hipEventRecord(ctx->ev_last[fam], s) == hipSuccess) {
            ctx->ev_last_valid[fam] = true;
            ctx->last_stream[fam] = s;
        }
    }
};

// `stream` argument of the _dev entry points: BBP_STREAM_CONTEXT selects the context's own stream, anything else -- NULL, the
// legacy default stream, included -- is the caller's hipStream_t and is honoured as such (include/bbp.h)
inline hipStream_t pick_stream(bbp_ctx* ctx, void* stream) { return stream == BBP_STREAM_CONTEXT ? ctx->stream : (hipStream_t)stream; }

// ---- the extern "C" barrier: lock + no exception ever crosses the boundary (include/bbp.h promises both) ---------------------
// bbp_last_error reports per calling thread AND per context: a failing call leaves (context, message) in a thread-local slot;
// bbp_last_error(ctx) answers from the slot only if it belongs to ctx, otherwise with the context's own last message -- a thread
// that drives several contexts (one per GPU) never reads one context's failure as another's.
std::string& tls_error();
const void*& tls_error_owner();
inline void set_tls_error(const void* ctx, const std::string& msg) {
    try {
        tls_error() = msg;
        tls_error_owner() = ctx;
    } catch (...) {
    }
}
inline const char* status_text(int32_t rc) {
    static const char* const names[] = {"ok", "verification failed", "bid list needs more than 2048 multipliers", "malformed input", "invalid argument",
                                        "device failure", "internal error"};
    return rc >= 0 && rc <= 6 ? names[rc] : "unknown status";
}
int32_t fault_injected(const char* site);  // BBP_FAULT_INJECT=<site>: throw at that site (tests/test_capi_symbols.py)

template <class F>
int32_t api_guard(bbp_ctx* ctx, F&& body) noexcept {
    int32_t rc;
    try {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        try {
            ctx->err.clear();  // whatever this call reports is this call's own message, never an older failure's
            rc = body();
        } catch (const std::invalid_argument& e) {
            ctx->err = std::string("invalid argument: ") + e.what();
            rc = BBP_ERR_BAD_ARG;
        } catch (const std::bad_alloc&) {
            ctx->err = "host allocation failed";
            rc = BBP_ERR_INTERNAL;
        } catch (const std::exception& e) {
            ctx->err = std::string("internal error: ") + e.what();
            rc = BBP_ERR_INTERNAL;
        } catch (...) {
            ctx->err = "internal error: unknown exception";
            rc = BBP_ERR_INTERNAL;
        }
        if (rc != BBP_OK) {
            try {
                if (ctx->err.empty()) ctx->err = status_text(rc);  // a failure path that set no text still reports its own status
                set_tls_error(ctx, ctx->err);
            } catch (...) {
            }
        }
    } catch (...) {  // the lock itself (std::system_error)
        rc = BBP_ERR_INTERNAL;
    }
    return rc;
}

inline bool is_pool(const bbp_ctx* ctx) { return ctx && !ctx->members.empty(); }
// setup.hip: GPU_MAX_HW_QUEUES is exported by the library itself when the process has not initialised HIP yet (state: 1 the caller's
// environment had it, 2 set here, 3 too late)
void own_hw_queues();
int hw_queues_state();
// pool.cpp: the host-pointer batch calls on a pool handle (block split by index over the members, results in request order)
int32_t pool_prove_batch(bbp_ctx* pool, uint32_t B, uint32_t N, const uint8_t* in, const uint8_t* entropy, uint8_t* out, int32_t* status);
int32_t pool_verify_batch(bbp_ctx* pool, uint32_t B, uint32_t N, const uint8_t* in, int32_t* status, bool aggregated, uint32_t group, uint32_t* n_fallback);
int32_t pool_msm_batch(bbp_ctx* pool, uint32_t B, uint32_t n_terms, const uint8_t* scalars, uint32_t layout, uint8_t* out32);
int32_t pool_reject(bbp_ctx* pool, const char* what);  // BBP_ERR_BAD_ARG + message: entry points that need ONE device
void pool_workers_start(bbp_ctx* pool);
void pool_workers_stop(bbp_ctx* pool);
int pool_member_numa_node(const bbp_ctx* pool, uint32_t i);

// msm.hip
// base_idx_dev holds n_idx_sets lists of n_terms indices; MSM number i uses list (i % n_idx_sets)
// msm_map_dev / n_active_dev (both or neither): a device-sized launch -- n_msm is the upper bound, *n_active_dev MSMs exist and MSM j
// takes its scalars from row msm_map_dev[j]
// chain_acc: the accumulate launch waits for the previous chained accumulate launch of this context (verifier lanes, see verify_serial_acc)
int32_t msm_launch(bbp_ctx* ctx, uint32_t n_msm, uint32_t n_terms, const u32* scalars_dev, const u32* base_idx_dev,
                   ge* out_points_dev, hipStream_t stream, uint32_t n_idx_sets = 1, int scratch_slot = 0, const u32* msm_map_dev = nullptr,
                   const u32* n_active_dev = nullptr, bool chain_acc = false);
int32_t fold_generators_launch(bbp_ctx* ctx, uint32_t n_proofs, const sc* g_dev, const sc* h_dev, ge* out_dev, hipStream_t stream,
                               int scratch_slot);
int32_t encode_launch(bbp_ctx* ctx, uint32_t n, const ge* pts_dev, uint8_t* out32_dev, hipStream_t stream);
size_t msm_scratch_bytes(uint32_t n_msm, uint32_t n_terms);
// prover.hip
int32_t tail_btab_build(bbp_ctx* ctx);

}  // namespace bbp
