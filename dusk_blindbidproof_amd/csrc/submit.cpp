// Call combiner (see submit.h).  Host-only C++: no HIP in this file, so the same translation unit also builds into the stub
// engine the CPU-tier server tests use (tests/stub_engine.cpp).
#include "submit.h"

#include <string.h>

#include <chrono>

namespace bbp {

// cv.wait_until on the steady clock is pthread_cond_clockwait, which GCC 11's ThreadSanitizer runtime does not intercept (it then
// believes the mutex stays locked across the wait and reports phantom double locks and races): the sanitizer build of the tests
// (tests/combiner_tsan.cpp) waits on the system clock instead, the product on the steady one.
static std::cv_status wait_until_steady(std::condition_variable& cv, std::unique_lock<std::mutex>& lk, std::chrono::steady_clock::time_point deadline) {
#if defined(__SANITIZE_THREAD__)
    const auto left = deadline - std::chrono::steady_clock::now();
    return cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::system_clock::duration>(left));
#else
    return cv.wait_until(lk, deadline);
#endif
}

void Combiner::configure(uint32_t window_us, uint32_t max_batch) {
    std::lock_guard<std::mutex> lk(mu_);
    window_us_ = window_us;
    max_batch_ = max_batch ? max_batch : 4096;
}

void Combiner::set_stagger(uint32_t us) {
    std::lock_guard<std::mutex> lk(mu_);
    stagger_us_ = us;
}

void Combiner::stats(uint64_t* n_calls, uint64_t* n_requests, uint32_t* max_seen) {
    std::lock_guard<std::mutex> lk(mu_);
    if (n_calls) *n_calls = n_calls_;
    if (n_requests) *n_requests = n_requests_;
    if (max_seen) *max_seen = max_seen_;
}

static bool same_class(const Request* a, const Request* b) {
    return a->kind == b->kind && a->N == b->N && a->rec_ver == b->rec_ver && a->in_len == b->in_len && (a->entropy == nullptr) == (b->entropy == nullptr);
}

void Combiner::designate_locked() {
    for (Request* q : q_) {
        if (leaders_ >= MAX_LEADERS) return;
        if (!q->lead && !q->leading) {  // a leader in its window still has its own request queued: never designate it twice
            q->lead = true;
            leaders_++;
            q->cv.notify_one();
        }
    }
}

int32_t Combiner::submit(bbp_ctx* ctx, Request& r) {
    std::unique_lock<std::mutex> lk(mu_);
    q_.push_back(&r);
    designate_locked();
    cv_window_.notify_one();  // a leader sitting in its batching window counts arrivals
    auto resign = [&] {       // give the leader slot back and let a queued caller have it
        r.leading = false;
        leaders_--;
        designate_locked();
    };
    for (;;) {
        r.cv.wait(lk, [&] { return r.done || r.lead; });
        if (!r.lead) return r.status;  // done, and not holding a leader slot
        r.lead = false;
        r.leading = true;
        if (r.done || q_.empty()) {  // nothing (left) for this leader to run
            resign();
            if (r.done) return r.status;
            continue;  // r is inside another leader's batch: wait for it
        }
        // Optional window: give concurrent callers a moment to join this batch.
        if (window_us_) {
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us_);
            while (q_.size() < max_batch_ && wait_until_steady(cv_window_, lk, deadline) != std::cv_status::timeout) {
            }
            if (q_.empty()) {  // the other leader took everything meanwhile
                resign();
                if (r.done) return r.status;
                continue;
            }
        }
        if (stagger_us_ && inflight_ > 0 && q_.front()->kind == 0) {  // a PROVE batch is on the device and this would be another:
            // let it grow until that one's opening stage is over (verifications have no such stage and never wait here)
            const auto start_at = last_start_ + std::chrono::microseconds(stagger_us_);
            while (inflight_ > 0 && q_.size() < max_batch_ && wait_until_steady(cv_window_, lk, start_at) != std::cv_status::timeout) {
            }
            if (q_.empty()) {
                resign();
                if (r.done) return r.status;
                continue;
            }
        }
        // one batch from the head of the queue: everything of the head's class, in arrival order (r itself is usually in it)
        const Request* head = q_.front();
        std::vector<Request*> batch;
        for (auto it = q_.begin(); it != q_.end() && batch.size() < max_batch_;) {
            if (same_class(*it, head)) {
                batch.push_back(*it);
                it = q_.erase(it);
            } else {
                ++it;
            }
        }
        for (Request* b : batch)
            if (b->lead) {  // designated but not yet awake, and now inside this batch: it will find nothing to lead
                b->lead = false;
                leaders_--;
            }
        const bool proving = batch[0]->kind == 0;  // inflight_ / last_start_ track prove batches only
        if (proving) {
            inflight_++;
            last_start_ = std::chrono::steady_clock::now();
        }
        lk.unlock();
        run_batch(ctx, batch);
        lk.lock();
        if (proving) inflight_--;
        cv_window_.notify_all();  // a leader holding back behind this batch may go now
        n_calls_++;
        n_requests_ += batch.size();
        if (batch.size() > max_seen_) max_seen_ = (uint32_t)batch.size();
        for (Request* b : batch) {
            b->done = true;
            if (b != &r) b->cv.notify_one();
        }
        resign();  // nobody serves other callers for longer than one batch
        if (r.done && !r.lead) return r.status;
        // (r.lead again: resign() found r still queued -- a different class than the batch it just ran -- and picked it)
    }
}

void Combiner::run_batch(bbp_ctx* ctx, std::vector<Request*>& batch) {
    const Request& h = *batch[0];
    const uint32_t B = (uint32_t)batch.size();
    int32_t rc;
    std::string err;
    std::vector<int32_t> status(B, 6);
    try {
        std::vector<uint8_t> in((size_t)B * h.in_len);
        for (uint32_t i = 0; i < B; i++) memcpy(&in[(size_t)i * h.in_len], batch[i]->in, h.in_len);
        if (h.kind == 0) {
            const size_t ent_len = 32 * (4 + (size_t)h.N) + 32, rec_len = 1121 + 32 * (4 + (size_t)h.N);
            std::vector<uint8_t> ent, out((size_t)B * rec_len);
            if (h.entropy) {
                ent.resize((size_t)B * ent_len);
                for (uint32_t i = 0; i < B; i++) memcpy(&ent[(size_t)i * ent_len], batch[i]->entropy, ent_len);
            }
            rc = prove_batch_locked(ctx, B, h.N, in.data(), h.entropy ? ent.data() : nullptr, out.data(), status.data(), &err);
            if (rc == 0)
                for (uint32_t i = 0; i < B; i++)
                    if (status[i] == 0) memcpy(batch[i]->out, &out[(size_t)i * rec_len], rec_len);
        } else {
            rc = verify_batch_locked(ctx, B, h.N, h.rec_ver, in.data(), status.data(), &err);
        }
    } catch (const std::bad_alloc&) {
        rc = 6;
        err = "host allocation failed";
    } catch (...) {
        rc = 6;
        err = "internal error in combined call";
    }
    for (uint32_t i = 0; i < B; i++) {
        batch[i]->status = rc ? rc : status[i];
        if (rc) batch[i]->err = err;
    }
}

}  // namespace bbp
