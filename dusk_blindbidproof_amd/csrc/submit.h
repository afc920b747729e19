// Call combiner: turns concurrent single-proof callers into batch calls (group commit).
//
// The reference runs one Proof::prove / Verify::verify per dusk-uds worker thread (src/main.rs:55, src/futures/main.rs:46-56,
// src/futures/prove.rs:21-26, verify.rs:21-26): N connections = N independent CPU proofs side by side.  On the GPU the unit of
// efficiency is the batch, so bbp_prove / bbp_verify (and through them the UDS server under server/) hand their request to this
// queue: the first caller to find the engine free becomes the leader, takes every queued request of its class (same kind, list
// length, record layout, entropy mode), runs ONE bbp_prove_batch / bbp_verify_batch under the context lock and distributes the
// results; requests that arrive while a batch is on the device form the next batch.  Up to TWO batches are in flight at a time
// (two leaders): the host-pointer batch calls hold the context lock only while they enqueue, so the second batch's host work and
// its opening stage on the device run under the first batch's MSM stage -- the engine's cross-call pipeline, kept full by the
// queue.  A leader runs ONE batch taken from the head of the queue, then leadership goes to a waiting caller.
#pragma once
#include <stdint.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
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
    bool lead = false;     // designated to run a batch, not yet acknowledged
    bool leading = false;  // acting as a leader right now (its own request may still sit in the queue during the window)
    std::condition_variable cv;      // each waiter has its own: a finished batch wakes its members, not every queued caller
};

// what a combined call runs (capi_prove.hip); both take the context lock themselves
int32_t prove_batch_locked(bbp_ctx* ctx, uint32_t B, uint32_t N, const uint8_t* in, const uint8_t* entropy, uint8_t* out, int32_t* status,
                           std::string* err);
int32_t verify_batch_locked(bbp_ctx* ctx, uint32_t B, uint32_t N, uint32_t rec_ver, const uint8_t* in, int32_t* status, std::string* err);

class Combiner {
  public:
    int32_t submit(bbp_ctx* ctx, Request& r);
    void configure(uint32_t window_us, uint32_t max_batch);
    // A leader that finds another batch in flight does not start before `us` after that batch STARTED: for the prover this is
    // the length of the opening stage (the next batch's opening cannot begin earlier anyway), and meanwhile the batch grows --
    // without it a few hundred closed-loop callers fragment into many small batches that each pay the full latency floor.
    void set_stagger(uint32_t us);
    void stats(uint64_t* n_calls, uint64_t* n_requests, uint32_t* max_seen);

  private:
    void run_batch(bbp_ctx* ctx, std::vector<Request*>& batch);
    std::mutex mu_;
    std::condition_variable cv_window_;  // arrivals -> the leader that sits in its batching window
    std::deque<Request*> q_;
    static constexpr int MAX_LEADERS = 2;  // three (as many as there are staging slots) fragments closed-loop load into more, smaller batches:
                                           // measured through the UDS server 14.7 k -> 11.5 k proofs/s prove-only, 8.2 k -> 7.1 k ops/s at 2048 connections
    int leaders_ = 0;         // callers currently designated to run (or running) a batch
    void designate_locked();  // hand free leader slots to queued callers
    uint32_t window_us_ = 0, max_batch_ = 4096, stagger_us_ = 0;
    int inflight_ = 0;
    std::chrono::steady_clock::time_point last_start_{};
    uint64_t n_calls_ = 0, n_requests_ = 0;
    uint32_t max_seen_ = 0;
};

}  // namespace bbp
