/* bbp.h -- C ABI of the MI355X blind-bid Bulletproofs engine (libbbp_hip.so).
 *
 * The reference (dusk-network/dusk-blindbidproof) has no FFI layer of its own: its hot path sits behind two
 * `pub` Rust functions and two IPC opcodes.  Each entry point below names the reference interface it replaces;
 * INTEGRATION.md shows the Rust `extern "C"` binding a maintainer would add in src/blindbid/{proof,verify}.rs.
 *
 * Conventions: all scalars are 32-byte little-endian; points are 32-byte ristretto255 encodings; every buffer is
 * caller-owned host memory unless the name ends in `_dev`; functions return a bbp_status; nothing throws or
 * aborts across this boundary (the reference builds with panic='abort', Cargo.toml:29 -- see SURVEY.md 5): every entry point
 * runs inside a try/catch barrier and turns a C++ exception into BBP_ERR_INTERNAL / BBP_ERR_BAD_ARG.
 * The library has NO CPU compute path: bbp_init fails with BBP_ERR_DEVICE when no gfx950 device is usable.
 *
 * Threading (SURVEY.md 8b "thread-safe after bbp_init"): the reference serves every connection on its own worker thread
 * (src/main.rs:55, src/futures/main.rs:46-56, one Proof::prove / Verify::verify per thread).  ONE context may be shared by any
 * number of host threads: each entry point takes the context's lock, and concurrent bbp_prove / bbp_verify calls are coalesced
 * into batch calls on the device (group commit: whatever queued up while the previous batch ran goes out as the next batch), so
 * N concurrent single proofs cost about one batch of N, not N times one.  bbp_last_error is per calling thread.
 * bbp_free must not race with other calls on the same context; asynchronous requests still queued when it is called are run first
 * (their callbacks fire before bbp_free returns).
 *
 * Several GPUs: bbp_init_all / bbp_pool_init return a POOL handle -- one context per GPU behind the same bbp_ctx* type -- that the
 * host-pointer entry points accept like a context: the reference's worker threads keep calling bbp_prove / bbp_verify on ONE
 * shared handle and every GPU of the node works (see "Device pool" below).
 */
#ifndef BBP_H
#define BBP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bbp_ctx bbp_ctx;

typedef enum {
    BBP_OK = 0,
    BBP_ERR_VERIFY = 1,    /* R1CSError::VerificationError (src/error.rs:22 via bulletproofs)            */
    BBP_ERR_GENS_LEN = 2,  /* R1CSError::InvalidGeneratorsLength: N > 202 needs > 2048 multipliers       */
    BBP_ERR_FORMAT = 3,    /* R1CSError::FormatError / Error::Io(InvalidData|UnexpectedEof) (error.rs)    */
    BBP_ERR_BAD_ARG = 4,   /* N == 0 or toggle >= N: the reference panics (src/gadgets.rs:103) or proves garbage */
    BBP_ERR_DEVICE = 5,    /* HIP failure / no device                                                     */
    BBP_ERR_INTERNAL = 6   /* host-side failure (allocation, internal invariant, any C++ exception): reported, never thrown */
} bbp_status;

#define BBP_MIMC_ROUNDS 90      /* src/gadgets.rs:4 */
#define BBP_GENS_CAPACITY 2048  /* src/blindbid/mod.rs:36 */
#define BBP_MAX_ITEMS 202       /* 1442 + 3N <= 2048 */
#define BBP_R1CS_PROOF_BYTES 1121 /* 1-phase compact R1CSProof::to_bytes */

/* Base-table indices for bbp_msm_batch layouts (device generator table order). */
#define BBP_BASE_BBLIND 0u /* PedersenGens::B_blinding */
#define BBP_BASE_G0 1u     /* BulletproofGens G[0..2048) */
#define BBP_BASE_H0 2049u  /* BulletproofGens H[0..2048) */
#define BBP_BASE_B 4097u   /* PedersenGens::B (ristretto basepoint) */
#define BBP_NUM_BASES 4098u

/* Layout ids for bbp_msm_batch. */
#define BBP_LAYOUT_BLIND_G_H 0u /* terms: B_blinding, G[0..m), H[0..m); n_terms = 1 + 2m  (A_I1, S1)   */
#define BBP_LAYOUT_BLIND_G 1u   /* terms: B_blinding, G[0..m);          n_terms = 1 + m   (A_O1)       */

/* Replaces generate_cs_transcript()'s per-call generator derivation (src/blindbid/mod.rs:34-40) and the
 * lazy_static CONSTANTS (src/blindbid/mod.rs:7-24): derives them ONCE on `device` and keeps them resident.
 * `device` is a HIP device ordinal (one context per GPU / per rank); -1 = a pool over every visible GPU (bbp_init_all below). */
int32_t bbp_init(int32_t device, bbp_ctx** out);

/* ---- Device pool -----------------------------------------------------------------------------------------------------------------
 * The reference's concurrency model is "one Proof::prove / Verify::verify per worker thread, all threads share the process-wide
 * state" (src/main.rs:55, src/futures/main.rs:46-56).  On a node with several GPUs that shared state is a pool: ONE handle, one
 * ordinary context per GPU behind it (tables replicated per GPU, no data ever crosses between GPUs -- proofs are independent).
 *   bbp_init_all            one member per visible HIP device (every one must be gfx950); bbp_init(-1, &h) is the same call
 *   bbp_pool_init           an explicit device list; a device may be named more than once (two members on one card: tests, or two
 *                           engine pipelines per GPU)
 * What a pool handle accepts:
 *   bbp_prove / bbp_verify  ONE call combiner for the whole pool: concurrent callers are coalesced into device batches as on a
 *                           single context, and every batch goes to the member with the fewest batches in flight; a burst is cut
 *                           into fair shares over the idle members
 *   bbp_prove_batch / bbp_verify_batch / bbp_verify_batch_aggregated / bbp_msm_batch
 *                           contiguous block split by index over the members (block sizes differ by at most one: member i of n
 *                           gets [i*B/n ...), the split of SURVEY.md 8e), one host thread per member, outputs and per-item
 *                           statuses in request order; the call returns the first member's non-zero status, if any
 *   bbp_witness_batch, bbp_get_generator, bbp_get_mimc_constant, bbp_ubench   served by member 0
 *   bbp_check_health        the members' flags OR-ed;  bbp_set_batching / bbp_batching_stats: the pool's combiner
 *   bbp_free                frees the members too
 * What it refuses (BBP_ERR_BAD_ARG; device pointers and streams belong to one device -- take a member with bbp_pool_member and
 * call it directly): every *_dev entry point, bbp_debug_challenges, bbp_set_profiling / bbp_last_timings; the stream getters
 * return NULL. */
int32_t bbp_init_all(bbp_ctx** out);
int32_t bbp_pool_init(const int32_t* devices, uint32_t n_devices, bbp_ctx** out);
uint32_t bbp_pool_size(const bbp_ctx* ctx);               /* number of members; 0 for an ordinary context */
bbp_ctx* bbp_pool_member(bbp_ctx* ctx, uint32_t i);        /* borrowed: owned and freed by the pool */
/* combined bbp_prove / bbp_verify device calls the pool has dealt to member i, and the requests they carried
 * (bbp_batching_stats on the member handle reports the same pair) */
int32_t bbp_pool_member_stats(bbp_ctx* ctx, uint32_t i, uint64_t* n_calls, uint64_t* n_requests);

/* `stream` arguments of the _dev entry points: a hipStream_t of the caller -- NULL is the legacy default stream and is honoured
 * as such -- or BBP_STREAM_CONTEXT for the context's own (non-blocking) stream. */
#define BBP_STREAM_CONTEXT ((void*)(intptr_t)-1)
/* The context's own stream as a hipStream_t, for callers that want to enqueue their own work (copies, consumers of the records)
 * in order with the engine's: passing this handle is the same as passing BBP_STREAM_CONTEXT.  Recommended for throughput: the
 * engine's four streams are created together at bbp_init and land on distinct hardware queues, whereas a stream the caller
 * created elsewhere may share a hardware queue with one of them (measured on MI355X: 61.5 vs 55.6 ms per 1024-proof batch
 * with a stream from PyTorch's pool as the caller's stream). */
void* bbp_context_stream(bbp_ctx* ctx);
/* A second stream of the context that the engine itself leaves idle, for the caller's ingest side (H2D copies of the next chunk,
 * bbp_prepare_bids_dev) while the first one carries prove / verify calls: same reason as above -- created at bbp_init beside
 * the engine's streams, it does not share a hardware queue with them. */
void* bbp_context_copy_stream(bbp_ctx* ctx);
/* The verifier has four independent lanes (own scratch each): calls issued on the streams of different lanes (lane < 4; NULL beyond)
 * overlap on the device -- the latency-bound front end of one runs under the MSM of another -- instead of being ordered one behind
 * the other like calls on any other pair of streams.  The host-pointer verify calls rotate over the lanes by themselves. */
void* bbp_context_verify_stream(bbp_ctx* ctx, uint32_t lane);
void bbp_free(bbp_ctx* ctx);
const char* bbp_last_error(const bbp_ctx* ctx);

/* Setup read-back for parity tests: compressed generator `index` (table order above) / MiMC constant i. */
int32_t bbp_get_generator(bbp_ctx* ctx, uint32_t index, uint8_t out32[32]);
int32_t bbp_get_mimc_constant(bbp_ctx* ctx, uint32_t i, uint8_t out32[32]);

/* Kernel-level hook (BASELINE.json configs[1]): B independent multiscalar multiplications over the shared
 * generator table; what bulletproofs' Prover::prove does with RistrettoPoint::multiscalar_mul for
 * A_I1 / A_O1 / S1 (reached from src/blindbid/proof.rs:88).  scalars: B * n_terms * 32 bytes, canonical (< l).
 * out32: B * 32 bytes, compressed results. */
int32_t bbp_msm_batch(bbp_ctx* ctx, uint32_t B, uint32_t n_terms, const uint8_t* scalars, uint32_t layout,
                      uint8_t* out32);

/* Device-resident variant used by bench.py so the timed region starts with inputs in HBM: same semantics with
 * `scalars_dev` / `out32_dev` being device pointers; `stream` is a hipStream_t (or BBP_STREAM_CONTEXT); does not
 * synchronise.  The device scalars cannot be screened on the host: they MUST be canonical (< l < 2^253) -- the recoding
 * only looks at bits 0..255 and silently drops a final carry, so a non-canonical scalar yields a wrong point (never a fault).
 * Scratch is grown inside the context on first use of a batch shape. */
int32_t bbp_msm_batch_dev(bbp_ctx* ctx, uint32_t B, uint32_t n_terms, const void* scalars_dev, uint32_t layout,
                          void* out32_dev, void* stream);

/* Witness helper: what the Go caller computes upstream of Proof::prove (defined by src/gadgets.rs:20-33,70-86):
 * m = mimc(k,0), x = mimc(d,m), y = mimc(seed,x), z_img = mimc(seed,m), y_inv = 1/y, q = d*y_inv.  Batched on
 * the device.  in: B * 96 bytes (d,k,seed); out: B * 192 bytes (m,x,y,y_inv,q,z_img). */
int32_t bbp_witness_batch(bbp_ctx* ctx, uint32_t B, const uint8_t* dks, uint8_t* out);

/* the caller-side pass on the device, feeding the batch prover / verifier directly (all pointers device
 * pointers, `stream` a hipStream_t, no synchronisation).  Per bid: bids_dev 96 B (d,k,seed as for bbp_witness_batch), lists_dev
 * N*32 B (the public bid list; entry `toggle` is overwritten with the bid's own x = mimc(d, mimc(k,0))), toggles_dev u64 (< N,
 * caller's responsibility).  Writes prove_in_dev rows in bbp_prove_batch_dev's input layout (7*32 + N*32 + 8 bytes) and, unless
 * NULL, verify_tail_dev rows score || z_img || seed || pub_list (96 + N*32 bytes) -- what follows the record in a
 * bbp_verify_batch row.  The next bbp_prove_batch_dev on this context waits for these rows by itself (the one exception to
 * "inputs must be complete at call time" below); anything else that reads them must be ordered after `stream` by the caller.
 * Give it a stream of its own when proving back to back: on the prover's stream it would queue behind the previous batch. */
int32_t bbp_prepare_bids_dev(bbp_ctx* ctx, uint32_t B, uint32_t N, const void* bids_dev, const void* lists_dev,
                             const void* toggles_dev, void* prove_in_dev, void* verify_tail_dev, void* stream);

/* Replaces Proof::prove (src/blindbid/proof.rs:36-46).
 * scalars7 = d,k,y,y_inv,q,z_img,seed; pub_list = N*32 bytes (Scalar::from_bits semantics, src/blindbid/bid.rs:27);
 * entropy = (4+N)*32 bytes of commitment blindings + 32 bytes rng seed (replaces thread_rng, proof.rs:53-64), or NULL
 * to draw from the OS.  proof_out record = R1CSProof bytes || 4*32 commitments || N*32 t_c;
 * *proof_len receives the R1CSProof byte count (1121). */
int32_t bbp_prove(bbp_ctx* ctx, const uint8_t scalars7[7 * 32], const uint8_t* pub_list, uint32_t N, uint64_t toggle,
                  const uint8_t* entropy, uint8_t* proof_out, uint32_t* proof_len);
uint32_t bbp_proof_record_size(uint32_t N); /* 1121 + 32*(4+N) */
uint32_t bbp_entropy_size(uint32_t N);      /* 32*(4+N) + 32 */

/* Replaces Verify::new(..).verify() (src/blindbid/verify.rs:27-89). record layout as produced by bbp_prove. */
int32_t bbp_verify(bbp_ctx* ctx, const uint8_t* record, uint32_t record_len, const uint8_t score[32],
                   const uint8_t z_img[32], const uint8_t seed[32], const uint8_t* pub_list, uint32_t N);

/* Asynchronous forms of the two calls above, for hosts that cannot park a thread per request (an epoll server; a Rust Future --
 * the reference's ProveFuture / VerifyFuture, src/futures/prove.rs:21-26, verify.rs:21-26, can store its Waker in `user` and be
 * woken by `done`).  Same arguments, same screening, same combining into device batches.  Return value: BBP_OK = queued, and
 * `done(user, status)` will be called exactly once, on an engine thread, with the status bbp_prove / bbp_verify would have
 * returned; anything else = decided at once (bad arguments, a record that fails the structural parse), `done` is NOT called.
 * The inputs are copied before the call returns; proof_out must stay valid until `done` runs (it is written before).  Inside
 * the callback bbp_last_error(ctx) is the request's message.  `done` runs on a thread that every request of the next batch is
 * waiting for: hand the result over and return (no blocking, no engine calls on the same context from inside it). */
typedef void (*bbp_done_fn)(void* user, int32_t status);
int32_t bbp_prove_async(bbp_ctx* ctx, const uint8_t scalars7[7 * 32], const uint8_t* pub_list, uint32_t N, uint64_t toggle,
                        const uint8_t* entropy, uint8_t* proof_out, bbp_done_fn done, void* user);
int32_t bbp_verify_async(bbp_ctx* ctx, const uint8_t* record, uint32_t record_len, const uint8_t score[32], const uint8_t z_img[32],
                         const uint8_t seed[32], const uint8_t* pub_list, uint32_t N, bbp_done_fn done, void* user);

/* The data-parallel path: B independent proofs with a common list length N, fixed-stride records.
 * in:  B * (7*32 + N*32 + 8) bytes: scalars7 || pub_list || toggle(u64 LE)
 * entropy: B * bbp_entropy_size(N) or NULL.  out: B * bbp_proof_record_size(N).  status: B entries. */
int32_t bbp_prove_batch(bbp_ctx* ctx, uint32_t B, uint32_t N, const uint8_t* in, const uint8_t* entropy,
                        uint8_t* out, int32_t* status);
/* in: B * (record_size(N) + 3*32 + N*32): record || score || z_img || seed || pub_list.  status: B entries
 * (BBP_OK / BBP_ERR_VERIFY / BBP_ERR_FORMAT). */
int32_t bbp_verify_batch(bbp_ctx* ctx, uint32_t B, uint32_t N, const uint8_t* in, int32_t* status);

/* Device-resident variants (bench.py / pipelined callers): same record layouts, every pointer a device pointer,
 * `stream` a hipStream_t (or BBP_STREAM_CONTEXT), no host synchronisation, no host-side argument screening
 * (toggle < N and canonical inputs are the caller's responsibility).  in_dev / entropy_dev must be COMPLETE when the call
 * is made: the prover's opening stage (witness, commitments, transcript rng) starts at once on an internal stream so that
 * it overlaps the previous call's MSM stage; outputs are ordered on `stream` as usual: complete for anything enqueued on
 * `stream` after the call, and not written before everything enqueued on `stream` ahead of the call has finished.  entropy_dev: B * bbp_entropy_size(N) for prove,
 * B * 32 for verify (the verifier's TranscriptRng seed).  status_dev: B * int32.
 * Scheduling (results never depend on it): a prove call's MSM-heavy stage runs as three slices on three streams; calls below 1024
 * proofs, and calls of any size up to 4096 made while THREE OR MORE earlier prove calls are still in flight, run it unsliced on one
 * of three internal streams in rotation instead (whole calls overlap; higher throughput for a caller that queues ahead, at the
 * price of a longer time to each call's records: BBP_ROTATE_DEEP_FROM / BBP_ROTATE_DEEP_MAX). */
int32_t bbp_prove_batch_dev(bbp_ctx* ctx, uint32_t B, uint32_t N, const void* in_dev, const void* entropy_dev, void* out_dev,
                            void* stream);
int32_t bbp_verify_batch_dev(bbp_ctx* ctx, uint32_t B, uint32_t N, const void* in_dev, const void* entropy_dev, void* status_dev,
                             void* stream);

/* Aggregated verification -- SURVEY.md 8f-4, an extension: the reference verifies one proof per call (verify.rs:88) and has
 * no equivalent.  Proofs are checked in groups of `group` (0 = BBP_AGG_GROUP_DEFAULT) with ONE generator MSM per group: the
 * per-proof mega-checks are summed with random weights drawn from each proof's verifier TranscriptRng (seeded by the OS /
 * entropy_dev).  The members of every group that fails are then checked one by one (an MSM each over the scalars already
 * computed), so status[] is what bbp_verify_batch reports (a bad proof slipping through needs a ~2^-250 accident).  Same record layout as bbp_verify_batch.  *n_fallback (may be NULL)
 * receives how many proofs were checked individually.  The _dev variant is stream-ordered like bbp_verify_batch_dev -- which
 * groups failed is decided on the device, the per-proof pass sizes itself from a device counter -- unless n_fallback is non-NULL:
 * delivering that count synchronises `stream`. */
#define BBP_AGG_GROUP_DEFAULT 32u
int32_t bbp_verify_batch_aggregated(bbp_ctx* ctx, uint32_t B, uint32_t N, const uint8_t* in, int32_t* status, uint32_t group,
                                    uint32_t* n_fallback);
int32_t bbp_verify_batch_aggregated_dev(bbp_ctx* ctx, uint32_t B, uint32_t N, const void* in_dev, const void* entropy_dev,
                                        void* status_dev, uint32_t group, uint32_t* n_fallback, void* stream);

/* Optional, once after bbp_init (or whenever a new list length N shows up): grow every per-batch buffer of the context (every member
 * of a pool) to what batches of up to max_batch proofs / verifications of list length N need, and compile the circuit for N.
 * Without it the buffers grow on demand, and a call that finds them too small frees and reallocates gigabytes under load (the
 * whole device waits: ~0.1-1 s, once per new high-water mark).  Costs about a dozen prove batches of that size (every schedule's buffers:
 * two for sliced calls, five for calls in rotation, the staging slots of the host-pointer calls); ~1.3 MB of device memory per proof and buffer. */
int32_t bbp_reserve(bbp_ctx* ctx, uint32_t max_batch, uint32_t N);

/* Micro-batching window of the call combiner, microseconds (default 0: a batch leaves as soon as the engine is free).  With a
 * window the leader of a batch waits that long for more concurrent bbp_prove / bbp_verify callers before it goes to the device --
 * what the UDS server (server/) uses to turn concurrent connections into GPU batches.  max_batch bounds one combined call. */
int32_t bbp_set_batching(bbp_ctx* ctx, uint32_t window_us, uint32_t max_batch);
/* Combiner statistics since bbp_init: combined device calls issued / requests they carried / largest batch (any may be NULL). */
int32_t bbp_batching_stats(bbp_ctx* ctx, uint64_t* n_calls, uint64_t* n_requests, uint32_t* max_seen);

/* Host-only synthesis check (no context, no device): compiles the blind-bid circuit for list length N exactly as bbp_prove would
 * (csrc/circuit.h mirrors src/gadgets.rs) and reports its size: n_mul = 1442 + 3N, n_cons = 2 n_mul + 3 + 3N.  N == 0 is
 * BBP_ERR_BAD_ARG (the reference panics at src/gadgets.rs:103), N > 202 BBP_ERR_GENS_LEN.  Also the place where the exception
 * barrier can be exercised without a GPU (BBP_FAULT_INJECT=compile in the environment makes the synthesis throw). */
int32_t bbp_debug_compile_circuit(uint32_t N, uint32_t* n_mul, uint32_t* n_cons);

/* Diagnostics: a short text report of what the context (every member of a pool) runs on and how it is configured -- device, free
 * memory, scheduling knobs -- into buf (NUL-terminated, truncated to cap).  Conditions known to cost throughput silently are
 * reported as lines starting with "WARNING:" (today: GPU_MAX_HW_QUEUES below 8, or unset while HIP was already initialised when
 * the engine was created; little free device memory).  The report also says where the hardware-queue setting came from -- bbp_init
 * exports GPU_MAX_HW_QUEUES=16 itself when the variable is unset and the process has not initialised HIP yet -- and how much
 * scratch the context holds in how many allocations (after bbp_reserve that count stands still).
 * Environment, read once per process: BBP_DEBUG_DEVICE_CHECK=1 asserts before every HIP call made for a context that the calling
 * thread's current device is the context's (first multi-GPU bring-up); BBP_TRACE_ALLOC=1 names every scratch buffer that grows. */
int32_t bbp_describe(bbp_ctx* ctx, char* buf, uint32_t cap);

/* Test hook: poisons the sorted scratch of the context's NEXT MSM launch with an out-of-range entry (what a stray write would leave).
 * The accumulate kernel clamps the gather (no fault), raises health bit 0, and -- the point of the hook -- the host-pointer call whose
 * results were fetched next returns BBP_ERR_DEVICE instead of BBP_OK with a wrong proof. */
int32_t bbp_debug_corrupt_scratch(bbp_ctx* ctx);

/* Engine self-check (synchronises the device).  *flags bit 0: an MSM table gather was out of range since bbp_init and had to be
 * clamped, i.e. engine scratch was corrupted (the one GPU fault of round 1 was such a state); 0 = healthy.  The flag is
 * sticky.  Every host-pointer call (bbp_prove[_batch], bbp_verify[_batch][_aggregated], bbp_msm_batch, the asynchronous forms)
 * reads it back with its results and returns BBP_ERR_DEVICE for the whole call once it is set -- never BBP_OK with results computed
 * from corrupted scratch; callers of the stream-ordered *_dev entry points poll this function at their own synchronisation points. */
int32_t bbp_check_health(bbp_ctx* ctx, uint32_t* flags);

/* Parity hook: the 32-scalar challenge block of proof `proof` of the LAST batch call of geometry (B, N):
 * y z u x w y^-1 t1..t6 tb1..tb6 t_x t_x~ e~ ... (MiscSlot order in csrc/batch.h), 32 x 32 bytes. */
int32_t bbp_debug_challenges(bbp_ctx* ctx, uint32_t B, uint32_t N, uint32_t proof, uint8_t* out32x32);

/* Integer-ALU roofline microbenchmarks (register-resident chains, no memory): kind 0 = v_mad_u64_u32, 1 = field multiply,
 * 2 = field square, 3 = mixed point addition, 4 = Montgomery product mod l, 5 = mixed point addition with an operand the
 * compiler cannot hoist (a gathered table row).  *ops_per_sec receives operations per second. */
int32_t bbp_ubench(bbp_ctx* ctx, int32_t kind, uint32_t blocks, uint32_t iters, double* ops_per_sec);

/* Per-kernel device timings: with profiling on, every kernel launch is bracketed by HIP events on its launch stream.
 * bbp_last_timings synchronises, drains them as (tag, microseconds) float pairs (tags: 1 = MSM accumulate kernel, 11 = MSM sort kernel, 12 = MSM fold kernel, 2 = encode, ...)
 * and reports the number of floats written in *n. */
int32_t bbp_set_profiling(bbp_ctx* ctx, int32_t on);
int32_t bbp_last_timings(bbp_ctx* ctx, float* out, uint32_t cap, uint32_t* n);

#ifdef __cplusplus
}
#endif
#endif
