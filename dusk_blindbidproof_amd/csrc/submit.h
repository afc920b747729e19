// Call combiner: turns concurrent single-proof callers into batch calls (group commit).
//
// The reference runs one Proof::prove / Verify::verify per dusk-uds worker thread (src/main.rs:55, src/futures/main.rs:46-56,
// src/futures/prove.rs:21-26, verify.rs:21-26): N connections = N independent CPU proofs side by side.  On the GPU the unit of
// efficiency is the batch, so bbp_prove / bbp_verify (and through them the UDS server under server/) hand their request to this
// queue: the first caller to find the engine free becomes the leader, takes every queued request of its class (same kind, list
// length, record layout, entropy mode), runs ONE bbp_prove_batch / bbp_verify_batch under the context lock and distributes the
// results; requests that arrive while a batch is on the device form the next batch.  Up to TWO batches are in flight at a time
// per device: the host-pointer batch calls hold the context lock only while they enqueue, so the second batch's host work and
// its opening stage on the device run under the first batch's MSM stage -- the engine's cross-call pipeline, kept full by the
// queue.
// Round 3: the batches are run by the combiner's OWN threads (two per target, started with the first request) instead of by
// whichever caller found a free leader slot.  That is what makes the asynchronous entry points possible (bbp_prove_async /
// bbp_verify_async: a request is queued and a callback fires when its batch is done -- what an epoll server or a Rust Future
// needs, since it cannot park a thread per request); the blocking bbp_prove / bbp_verify are the same path plus a wait.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct bbp_ctx;

namespace bbp {

struct Request {
    int kind = 0;                    // 0 = prove, 1 = verify
    uint32_t N = 0;                  // bid-list length
    uint32_t rec_ver = 0;            // verify: 0 = compact 1121-byte R1CSProof, 1 = two-phase 1217-byte layout
    const uint8_t* in = nullptr;     // prove: scalars7 || pub_list || toggle ; verify: record || score || z_img || seed || pub_list
    size_t in_len = 0;
    const uint8_t* entropy = nullptr;  // prove only: bbp_entropy_size(N) bytes, or NULL = OS randomness
    uint8_t* out = nullptr;          // prove only: bbp_proof_record_size(N) bytes
    int32_t status = 6;              // BBP_ERR_INTERNAL until the batch has run
    std::string err;
    bool done = false;
    std::condition_variable cv;      // blocking callers: each waiter has its own, a finished batch wakes its members only
    // asynchronous requests: called exactly once, on a combiner thread, with no lock held, after status / err / out are final;
    // the hook owns the Request from then on (it may delete it).  nullptr = a blocking caller waits on cv instead.
    void (*on_done)(Request*) = nullptr;
    void (*user_fn)(void*, int32_t) = nullptr;  // what the C ABI's hook forwards to
    void* user = nullptr;
    const void* origin = nullptr;  // the handle the request was made on (a pool or a context): whose error slot the message belongs to
    std::vector<uint8_t> own_in, own_entropy;  // storage of an asynchronous request's inputs (in / entropy point into them)
};

// what a combined call runs (capi_prove.hip); both take the context lock themselves
int32_t prove_batch_locked(bbp_ctx* ctx, uint32_t B, uint32_t N, const uint8_t* in, const uint8_t* entropy, uint8_t* out, int32_t* status,
                           std::string* err);
int32_t verify_batch_locked(bbp_ctx* ctx, uint32_t B, uint32_t N, uint32_t rec_ver, const uint8_t* in, int32_t* status, std::string* err);

class Combiner {
  public:
    // `ctx` is where the batch runs unless the combiner has been given targets (a device pool): then every batch goes to the
    // least-loaded target
    int32_t submit(bbp_ctx* ctx, Request& r);
    // queue and return: r->on_done fires when the batch that carried it is done.  false = the combiner has no thread to run it
    // (thread creation failed): nothing was queued, the hook will not fire
    bool submit_async(bbp_ctx* ctx, Request* r);
    Combiner();
    ~Combiner();  // runs whatever is still queued (every callback fires, every waiter returns), then joins the threads
    Combiner(const Combiner&) = delete;
    Combiner& operator=(const Combiner&) = delete;
    // Device pool (pool.cpp): one combiner in front of several contexts (one per GPU).  A leader reserves the target with the
    // fewest combined calls in flight, takes its fair share of what is queued (queue / idle targets, so that a burst spreads over
    // the idle devices instead of landing on one) and runs it there; up to two batches per target are in flight, and the prover's
    // stagger is kept per target.
    void set_targets(const std::vector<bbp_ctx*>& targets);
    void target_stats(size_t i, uint64_t* n_calls, uint64_t* n_requests);
    void configure(uint32_t window_us, uint32_t max_batch);
    // A leader that finds another batch in flight does not start before `us` after that batch STARTED: for the prover this is
    // the length of the opening stage (the next batch's opening cannot begin earlier anyway), and meanwhile the batch grows --
    // without it a few hundred closed-loop callers fragment into many small batches that each pay the full latency floor.
    void set_stagger(uint32_t us);
    // batch threads per target for requests of `kind` (0 prove, 1 verify): as many verification batches in flight as the engine has verifier lanes
    void set_leaders(int kind, int n);
    void set_lopsided_wait(bool on);
    // behind a prove batch of at most `small_batch` proofs the stagger is `us` instead (their opening stage is shorter)
    void set_small_stagger(uint32_t small_batch, uint32_t us);
    // Behind a busy device a prove batch leaves only when the queue has stopped growing for `quiet_us`, at most `cap_us` after its
    // stagger is over (0 = off).
    void set_quiet(uint32_t quiet_us, uint32_t cap_us);
    // Behind a prove batch in flight the next one leaves no earlier than (expected end of that batch) - open_us - margin_us, where a
    // batch of B proofs is expected to take open_us (opening stage) + per_proof_us * B once the device is its own.  margin_us < 0 = off.
    // adapt: per_proof_us is then only the starting value -- the combiner measures what a proof of a pipelined batch really costs
    // (end of the batch minus the later of the previous batch's end and this batch's opening stage, per proof; running average,
    // kept within -40 % / +25 % of the starting value) and paces with that: a constant calibrated for one engine build paces a
    // faster one at the old rate (round 4: the engine got 6 % faster and the socket did not, until this).
    void set_hold(int32_t margin_us, uint32_t open_us, double per_proof_us, bool adapt = false);
    // A prove burst that finds its device idle and holds at least 2 n requests is cut in two (each half >= n): 0 = never.
    void set_split_min(uint32_t n);
    void stats(uint64_t* n_calls, uint64_t* n_requests, uint32_t* max_seen);

  private:
    void run_batch(bbp_ctx* ctx, std::vector<Request*>& batch);
    struct Target {
        bbp_ctx* ctx = nullptr;
        int running[2] = {0, 0};  // combined calls reserved / running on this target, per kind (0 = prove, 1 = verify)
        int prove_inflight = 0;  // prove batches among them
        std::chrono::steady_clock::time_point last_start{};  // start of the last prove batch (stagger)
        uint32_t last_size = 0;                              // ... and its size
        std::chrono::steady_clock::time_point last_done{};   // when the last prove batch of this target came back ...
        size_t last_done_size = 0;                           // ... and how many callers it carried
        std::chrono::steady_clock::time_point est_end{};     // when the prove batches dealt to this target so far are expected to be done
        uint64_t n_calls = 0, n_requests = 0;
    };
    std::vector<Target> targets_;  // empty until the first submit of a plain context (then: that context)
    size_t rr_ = 0;                // tie-break cursor
    size_t pick_target_locked(int kind);
    static constexpr uint32_t MIN_SHARE = 64;  // a fair share is never cut below this many requests: tiny batches pay the full latency floor
    std::mutex mu_;
    // one queue and one set of batch threads per request kind (0 = prove, 1 = verify): the engine runs the two on disjoint
    // scratch and streams, so a verification batch neither waits for a thread that is inside a prove call nor queues behind
    // prove requests (measured through the UDS server, closed loop 2048 connections: verify p50 25-47 ms with shared threads)
    struct Lane {
        std::deque<Request*> q;
        std::condition_variable cv_window;  // arrivals -> a batch thread that sits in its batching window / holds back behind a prove batch
        std::condition_variable cv_work;    // arrivals -> an idle batch thread
        int idle = 0;                       // batch threads waiting for work
        int n_threads = 0;
    };
    Lane lane_[2];
    std::vector<std::thread> threads_;
    bool stop_ = false;
    bool enqueue_locked(bbp_ctx* ctx, Request* r);
    void thread_main(int kind);
    static constexpr int LEADERS_PER_TARGET = 2;  // three (as many as there are staging slots) fragments closed-loop load into more, smaller batches:
                                                  // measured through the UDS server 14.7 k -> 11.5 k proofs/s prove-only, 8.2 k -> 7.1 k ops/s at 2048 connections
    int leaders_[2] = {LEADERS_PER_TARGET, LEADERS_PER_TARGET};  // batch threads per target for prove / verify requests (set_leaders)
    int max_leaders_locked(int kind) const { return leaders_[kind] * (int)(targets_.empty() ? 1 : targets_.size()); }
    uint32_t window_us_ = 0, max_batch_ = 4096, stagger_us_ = 0, split_min_ = 0, quiet_us_ = 300, quiet_cap_us_ = 0, open_us_ = 40000;
    int32_t hold_margin_us_ = -1;
    bool lopsided_ = true;  // set_lopsided_wait: a few requests behind one large batch wait for its callers (closed-loop clients)
    uint32_t small_batch_ = 0, small_stagger_us_ = 0xffffffffu;
    double per_proof_us_ = 48.0, per_proof_us0_ = 48.0;
    bool adapt_ = false;
    uint64_t n_calls_ = 0, n_requests_ = 0;
    uint32_t max_seen_ = 0;
    FILE* log_ = nullptr;  // BBP_BATCH_LOG=path: one line per batch (tools/batch_log.py)
    std::chrono::steady_clock::time_point t0_ = std::chrono::steady_clock::now();
};

}  // namespace bbp
