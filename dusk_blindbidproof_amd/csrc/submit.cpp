// Call combiner (see submit.h).  Host-only C++: no HIP in this file, so the same translation unit also builds into the stub
// engine the CPU-tier server tests use (tests/stub_engine.cpp).
#include "submit.h"

#include <string.h>

#include <algorithm>
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

void Combiner::set_quiet(uint32_t quiet_us, uint32_t cap_us) {
    std::lock_guard<std::mutex> lk(mu_);
    quiet_us_ = quiet_us ? quiet_us : 1;
    quiet_cap_us_ = cap_us;
}

void Combiner::set_hold(int32_t margin_us, uint32_t open_us, double per_proof_us, bool adapt) {
    std::lock_guard<std::mutex> lk(mu_);
    hold_margin_us_ = margin_us;
    open_us_ = open_us;
    per_proof_us_ = per_proof_us0_ = per_proof_us;
    adapt_ = adapt;
}

void Combiner::set_small_stagger(uint32_t small_batch, uint32_t us) {
    std::lock_guard<std::mutex> lk(mu_);
    small_batch_ = small_batch;
    small_stagger_us_ = us;
}

void Combiner::set_leaders(int kind, int n) {
    std::lock_guard<std::mutex> lk(mu_);
    if ((kind == 0 || kind == 1) && n >= 1 && n <= 16) leaders_[kind] = n;
}

void Combiner::set_lopsided_wait(bool on) {
    std::lock_guard<std::mutex> lk(mu_);
    lopsided_ = on;
}

void Combiner::set_split_min(uint32_t n) {
    std::lock_guard<std::mutex> lk(mu_);
    split_min_ = n;
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

void Combiner::set_targets(const std::vector<bbp_ctx*>& targets) {
    std::lock_guard<std::mutex> lk(mu_);
    targets_.clear();
    for (bbp_ctx* c : targets) {
        Target t;
        t.ctx = c;
        targets_.push_back(t);
    }
}

void Combiner::target_stats(size_t i, uint64_t* n_calls, uint64_t* n_requests) {
    std::lock_guard<std::mutex> lk(mu_);
    if (n_calls) *n_calls = i < targets_.size() ? targets_[i].n_calls : 0;
    if (n_requests) *n_requests = i < targets_.size() ? targets_[i].n_requests : 0;
}

// least-loaded target; ties go round-robin so that an idle pool is used evenly
size_t Combiner::pick_target_locked(int kind) {
    size_t best = 0;
    int best_load = 1 << 30;
    for (size_t k = 0; k < targets_.size(); k++) {
        const size_t i = (rr_ + k) % targets_.size();
        if (targets_[i].running[kind] < best_load) {
            best_load = targets_[i].running[kind];
            best = i;
        }
    }
    rr_ = (best + 1) % targets_.size();
    return best;
}

Combiner::Combiner() {
    if (const char* p = getenv("BBP_BATCH_LOG")) log_ = fopen(p, "a");
}

Combiner::~Combiner() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    for (Lane& L : lane_) {
        L.cv_work.notify_all();
        L.cv_window.notify_all();
    }
    for (auto& t : threads_) t.join();
    if (log_) fclose(log_);
}

// queue a request; start the batch threads with the first one (two per target: a second batch is prepared and its opening stage
// runs while the first is on the device).  false = no thread could be started and none is running
bool Combiner::enqueue_locked(bbp_ctx* ctx, Request* r) {
    if (targets_.empty()) {  // a plain context: its one target is itself
        Target t;
        t.ctx = ctx;
        targets_.push_back(t);
    }
    const int kind = r->kind ? 1 : 0;
    Lane& L = lane_[kind];
    while (L.n_threads < max_leaders_locked(kind)) {  // prove and verify requests have their own queue and their own batch threads:
        try {                                     // a verification never waits for a thread that sits in a 100 ms prove call
            threads_.emplace_back([this, kind] { thread_main(kind); });
            L.n_threads++;
        } catch (...) {
            break;
        }
    }
    if (L.n_threads == 0) return false;
    L.q.push_back(r);
    if (L.idle > 0) L.cv_work.notify_one();
    if (L.q.size() >= max_batch_) L.cv_window.notify_all();  // a thread in its window / holding back has a full batch now
    return true;
}

int32_t Combiner::submit(bbp_ctx* ctx, Request& r) {
    std::unique_lock<std::mutex> lk(mu_);
    r.on_done = nullptr;
    if (!enqueue_locked(ctx, &r)) {
        r.err = "call combiner: cannot start a batch thread";
        return r.status = 6;
    }
    r.cv.wait(lk, [&] { return r.done; });
    return r.status;
}

bool Combiner::submit_async(bbp_ctx* ctx, Request* r) {
    std::lock_guard<std::mutex> lk(mu_);
    return enqueue_locked(ctx, r);
}

void Combiner::thread_main(int kind) {
    Lane& L = lane_[kind];
    std::deque<Request*>& q_ = L.q;
    std::condition_variable& cv_window_ = L.cv_window;
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
        L.idle++;
        L.cv_work.wait(lk, [&] { return stop_ || !q_.empty(); });
        L.idle--;
        if (q_.empty()) return;  // (only when stopping: requests still queued at that moment are run first -- every callback fires, every waiter returns)
        // Optional window: give concurrent callers a moment to join this batch.
        if (window_us_) {
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us_);
            while (!stop_ && q_.size() < max_batch_ && wait_until_steady(cv_window_, lk, deadline) != std::cv_status::timeout) {
            }
            if (q_.empty()) continue;  // another thread took everything meanwhile
        }
        // reserve the least-loaded target now (under the lock), so that two threads never count on the same idle device
        const size_t ti = pick_target_locked(kind);
        size_t idle_targets = 0;  // targets with nothing (of this kind) reserved, this one included if so: what a burst should spread over
        for (const Target& t : targets_) idle_targets += t.running[kind] == 0;
        targets_[ti].running[kind]++;
        if (stagger_us_ && targets_[ti].prove_inflight > 0 && q_.front()->kind == 0) {  // a PROVE batch is on that device and this would be another:
            // let it grow until that one's opening stage is over (verifications have no such stage and never wait here)
            // (a small batch's opening stage is the cooperative rng chain, ~12-15 ms instead of ~37: the next batch may follow sooner)
            const auto start_at = targets_[ti].last_start +
                                  std::chrono::microseconds(targets_[ti].last_size <= small_batch_ && small_stagger_us_ < stagger_us_ ? small_stagger_us_ : stagger_us_);
            while (!stop_ && targets_[ti].prove_inflight > 0 && q_.size() < max_batch_ && wait_until_steady(cv_window_, lk, start_at) != std::cv_status::timeout) {
            }
            // ... nor before its own MSM stage could start anyway: the batch in flight is expected to end at est_end (below), this
            // batch's opening stage takes open_us_, so leaving earlier than est_end - open_us_ only means leaving SMALLER -- through
            // the UDS server, 3072 closed-loop connections: a thread woken by the first few returning callers left with 65 proofs
            // and held a pipeline slot for 90 ms while 1400 more queued up behind it.
            if (hold_margin_us_ >= 0) {
                const auto hold_until = targets_[ti].est_end - std::chrono::microseconds(open_us_ + hold_margin_us_);
                while (!stop_ && targets_[ti].prove_inflight > 0 && q_.size() < max_batch_ && wait_until_steady(cv_window_, lk, hold_until) != std::cv_status::timeout) {
                }
            }
            // ... nor right after ANOTHER batch of this target has come back with far more callers than are queued yet: its callers
            // (closed-loop clients) are on their way in -- 1834 replies take milliseconds to write and to be answered -- and a thread
            // whose hold happens to expire in that moment would leave with the first 175 of them, after which the two batches in flight
            // stay lopsided (175 / 2897 instead of 1536 / 1536: 16 k instead of 20.8 k proofs/s through the socket).  Bounded: 10 ms
            // after that completion, or until half as many requests as it carried are queued.
            if (hold_margin_us_ >= 0) {
                const auto until = targets_[ti].last_done + std::chrono::microseconds(10000);
                while (!stop_ && targets_[ti].prove_inflight > 0 && q_.size() < max_batch_ && 2 * q_.size() < targets_[ti].last_done_size &&
                       std::chrono::steady_clock::now() < until) {
                    wait_until_steady(cv_window_, lk, std::min(until, std::chrono::steady_clock::now() + std::chrono::microseconds(500)));
                }
            }
            // ... and while the queue is still GROWING behind a busy device, keep holding back (quiet-period detection, bounded): the
            // callers of a batch that just finished come back as a burst spread over milliseconds -- 1536 replies written and
            // answered one after the other -- and a thread that left with the first few hundred of them would put a small,
            // inefficient batch in front of the large one that is forming (closed loop, 3072 connections: average batch 1014 of a
            // possible 1536).  Nothing is lost by waiting: this batch's MSM stage cannot start before the one in flight ends.
            const auto quiet_cap = std::chrono::steady_clock::now() + std::chrono::microseconds(quiet_cap_us_);
            while (!stop_ && quiet_cap_us_ && targets_[ti].prove_inflight > 0 && q_.size() < max_batch_) {
                const size_t before = q_.size();
                const auto tick = std::min(quiet_cap, std::chrono::steady_clock::now() + std::chrono::microseconds(quiet_us_));
                wait_until_steady(cv_window_, lk, tick);
                if (q_.size() == before || std::chrono::steady_clock::now() >= quiet_cap) break;
            }
            // ... and a few requests behind ONE large batch, with nobody else arriving, are that batch's own early finishers' next
            // requests (closed-loop callers): sent now they come back early again, and the pair in flight stays lopsided for good
            // (175 / 2897 proofs: the same 3072 connections give 17.1 k proofs/s in that state and 20.4 k with 1536 / 1536 -- each
            // run of the sweep fell into one or the other).  They wait for the large batch to end and for its callers to come back
            // (bounded as above); what is queued then meets an idle device and is cut in half below.  Steady arrivals never see this:
            // their queue passes a quarter of the batch in flight long before that batch ends.
            if (lopsided_ && split_min_ && hold_margin_us_ >= 0 && targets_[ti].prove_inflight == 1 && targets_[ti].last_size >= 2 * split_min_ &&
                4 * q_.size() <= targets_[ti].last_size) {
                const auto cap = targets_[ti].est_end + std::chrono::microseconds(20000);
                while (!stop_ && targets_[ti].prove_inflight > 0 && 4 * q_.size() <= targets_[ti].last_size && std::chrono::steady_clock::now() < cap)
                    wait_until_steady(cv_window_, lk, std::min(cap, std::chrono::steady_clock::now() + std::chrono::microseconds(500)));
                if (targets_[ti].prove_inflight == 0) {
                    const auto until = targets_[ti].last_done + std::chrono::microseconds(10000);
                    while (!stop_ && q_.size() < max_batch_ && 2 * q_.size() < targets_[ti].last_done_size && std::chrono::steady_clock::now() < until)
                        wait_until_steady(cv_window_, lk, std::min(until, std::chrono::steady_clock::now() + std::chrono::microseconds(500)));
                }
            }
            if (q_.empty()) {
                targets_[ti].running[kind]--;
                continue;
            }
        }
        // one batch from the head of the queue: requests of the head's class, in arrival order.  With several targets a thread
        // takes its fair share only -- what is queued divided by the idle targets -- so the rest is there for the threads beside it.
        const Request* head = q_.front();
        size_t limit = max_batch_;
        if (targets_.size() > 1 && idle_targets > 1) {
            size_t n_class = 0;
            for (const Request* q : q_) n_class += same_class(q, head);
            const size_t share = (n_class + idle_targets - 1) / idle_targets;
            limit = std::min<size_t>(limit, std::max<size_t>(share, MIN_SHARE));
        }
        if (head->kind == 0 && targets_[ti].prove_inflight == 0 && split_min_) {
            // The device holds no prove batch: whatever is queued would go out as ONE batch whose opening stage (~40 ms of serial
            // transcript work) runs beside nothing, and -- closed-loop callers -- come back as one synchronised burst again.  A burst
            // of at least two full-efficiency batches is cut in half instead: the second half follows a stagger later, its opening
            // runs under the first half's MSM stage, and the two halves' callers return out of phase from then on (the engine's
            // cross-call pipeline stays full: 3072 connections prove-only through the socket 13.3 k -> see DESIGN.md 6c).
            size_t n_class = 0;
            for (const Request* q : q_) n_class += same_class(q, head);
            if (n_class >= 2 * (size_t)split_min_) limit = std::min<size_t>(limit, std::max<size_t>((n_class + 1) / 2, split_min_));
        }
        std::vector<Request*> batch;
        for (auto it = q_.begin(); it != q_.end() && batch.size() < limit;) {
            if (same_class(*it, head)) {
                batch.push_back(*it);
                it = q_.erase(it);
            } else {
                ++it;
            }
        }
        if (!q_.empty() && L.idle > 0) L.cv_work.notify_one();  // another class, or the rest of a burst: not this thread's batch
        const bool proving = batch[0]->kind == 0;  // prove_inflight / last_start track prove batches only
        if (proving) {
            const auto now = std::chrono::steady_clock::now();
            // expected end of this batch: its MSM stage (per_proof_us_ each) starts when its opening stage is over AND the batch in
            // flight has ended.  An estimate only: it decides how long the NEXT batch may keep growing.
            const auto opened = now + std::chrono::microseconds(open_us_);
            const auto base = targets_[ti].prove_inflight > 0 && targets_[ti].est_end > opened ? targets_[ti].est_end : opened;
            targets_[ti].est_end = base + std::chrono::microseconds((long long)(per_proof_us_ * (double)batch.size()));
            targets_[ti].prove_inflight++;
            targets_[ti].last_start = now;
            targets_[ti].last_size = (uint32_t)batch.size();
        }
        bbp_ctx* const where = targets_[ti].ctx;
        const auto t_batch = std::chrono::steady_clock::now();
        const size_t q_left = q_.size();
        const int inflight_before = proving ? targets_[ti].prove_inflight - 1 : 0;
        lk.unlock();
        run_batch(where, batch);
        lk.lock();
        if (log_) {  // BBP_BATCH_LOG: start (ms since the combiner was made), duration, kind, target, size, prove batches already in flight, queue left behind
            const auto t_end = std::chrono::steady_clock::now();
            fprintf(log_, "%.2f %.2f %s %zu %zu %d %zu %.1f\n", std::chrono::duration<double, std::milli>(t_batch - t0_).count(),
                    std::chrono::duration<double, std::milli>(t_end - t_batch).count(), proving ? "prove" : "verify", ti, batch.size(), inflight_before, q_left,
                    per_proof_us_);  // (last column: the pacing estimate, microseconds per proof, as it stood when this batch ended)
        }
        if (proving) {
            const auto t_end = std::chrono::steady_clock::now();
            if (adapt_ && inflight_before > 0 && batch.size() >= 256) {
                // what a proof of this batch cost once the device was its own: its MSM stage could start when its opening stage was over
                // AND the batch ahead of it had ended (that batch's end is still in last_done: it ended first)
                const auto heavy_from = std::max(targets_[ti].last_done, t_batch + std::chrono::microseconds(open_us_));
                const double us = std::chrono::duration<double, std::micro>(t_end - heavy_from).count() / (double)batch.size();
                if (us > 0.5 * per_proof_us0_ && us < 1.5 * per_proof_us0_) {
                    per_proof_us_ = 0.75 * per_proof_us_ + 0.25 * us;
                    per_proof_us_ = std::min(1.25 * per_proof_us0_, std::max(0.6 * per_proof_us0_, per_proof_us_));
                }
            }
            targets_[ti].prove_inflight--;
            targets_[ti].last_done = t_end;
            targets_[ti].last_done_size = batch.size();
        }
        targets_[ti].running[kind]--;
        cv_window_.notify_all();  // a thread holding back behind this batch may go now
        n_calls_++;
        n_requests_ += batch.size();
        targets_[ti].n_calls++;
        targets_[ti].n_requests += batch.size();
        if (batch.size() > max_seen_) max_seen_ = (uint32_t)batch.size();
        std::vector<Request*> hooks;
        for (Request* b : batch) {
            if (b->on_done) {
                hooks.push_back(b);
            } else {
                b->done = true;
                b->cv.notify_one();  // (the waiter cannot return, and its Request cannot die, before this thread lets go of mu_)
            }
        }
        if (!hooks.empty()) {  // asynchronous requests: completion hooks run with no lock held; each owns its Request from here on
            lk.unlock();
            for (Request* b : hooks) b->on_done(b);
            lk.lock();
        }
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
