// Device pool: ONE handle in front of one engine context per GPU (include/bbp.h "Device pool").
//
// The reference serves every connection on its own dusk-uds worker thread and each of them calls Proof::prove / Verify::verify
// (src/main.rs:55, src/futures/main.rs:46-56, src/futures/prove.rs:21-26, verify.rs:21-26).  With the binding of INTEGRATION.md
// those calls land in bbp_prove / bbp_verify on one shared handle; if that handle is a single context, one GPU of the node
// works and seven idle.  A pool handle is a bbp_ctx without device state whose `members` are ordinary contexts:
//   * bbp_prove / bbp_verify: ONE call combiner (submit.cpp) collects the concurrent callers and deals every batch to the member
//     with the fewest combined calls in flight; a burst is split into fair shares over the idle members;
//   * bbp_prove_batch / bbp_verify_batch[_aggregated] / bbp_msm_batch: contiguous block split by index over the members, sizes
//     differing by at most one (= sharding.shard_range), one host thread per member, results land in request order
//     because every member writes straight into its slice of the caller's buffers;
//   * no data crosses between GPUs (proofs are independent units; tables are replicated per member at init).
// Host-only C++ (no kernels); everything that touches a device goes through the members' own entry points.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sched.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "context.h"
#include "submit.h"

namespace bbp {

// ---- persistent worker threads, a few per member (round 4) -----------------------------------------------------------------------
// The host-pointer batch calls on a pool used to start one std::thread per member per call.  Now every member has POOL_WORKERS
// threads of its own, started at bbp_pool_init: each binds to the member's GPU once (hipSetDevice) and, where sysfs names the NUMA
// node of that GPU's PCIe root, restricts itself to that node's cores (the staging copies go through pinned host buffers the
// thread touches: first touch and memcpy stay on the GPU's socket).  Several per member, not one: bbp_prove_batch on a context
// holds the context lock only while it enqueues and then waits for its staging slot with the lock released -- two or three calls
// in flight per GPU are what keeps the engine's cross-call pipeline full (one caller thread: 11.0 k proofs/s, two: 18.5 k) -- so
// concurrent callers of the pool must not serialise on a single worker per member.
struct PoolWorkers {
    struct Member {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<std::function<void()>> q;
        std::vector<std::thread> th;
        bool stop = false;
        int numa_node = -1;
    };
    std::vector<std::unique_ptr<Member>> m;
    ~PoolWorkers() {
        for (auto& mem : m) {
            {
                std::lock_guard<std::mutex> lk(mem->mu);
                mem->stop = true;
            }
            mem->cv.notify_all();
        }
        for (auto& mem : m)
            for (auto& t : mem->th)
                if (t.joinable()) t.join();
    }
};

// NUMA node of a GPU from sysfs (-1: unknown / single node); the cores of a node as a cpu_set_t
static int gpu_numa_node(int device) {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return -1;
    for (char* c = bdf; *c; c++)
        if (*c >= 'A' && *c <= 'Z') *c = (char)(*c - 'A' + 'a');
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}
static bool numa_node_cpus(int node, cpu_set_t* set) {
    CPU_ZERO(set);
    if (node < 0) return false;
    const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096] = {0};
    const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!got) return false;
    int n = 0;
    for (char* p = buf; *p && *p != '\n';) {  // "0-15,128-143"
        char* end;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) {
            CPU_SET((int)c, set);
            n++;
        }
        if (*p == ',') p++;
    }
    return n > 0;
}

static void pool_worker_main(PoolWorkers::Member* mem, bbp_ctx* member) {
    (void)hipSetDevice(member->device);  // once: every task of this thread runs on this member (its entry points set it again)
    static const bool pin = [] { const char* e = getenv("BBP_POOL_PIN"); return !e || atoi(e) != 0; }();
    cpu_set_t node_set, allowed;
    if (pin && numa_node_cpus(mem->numa_node, &node_set) && sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
        CPU_AND(&node_set, &node_set, &allowed);  // never outside what the container / cpuset allows
        if (CPU_COUNT(&node_set) > 0) (void)sched_setaffinity(0, sizeof node_set, &node_set);
    }
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> lk(mem->mu);
            mem->cv.wait(lk, [&] { return mem->stop || !mem->q.empty(); });
            if (mem->q.empty()) return;  // stop, and nothing left to run
            job = std::move(mem->q.front());
            mem->q.pop_front();
        }
        job();
    }
}

void pool_workers_start(bbp_ctx* pool) {
    const char* e = getenv("BBP_POOL_WORKERS");
    const int per = e && atoi(e) > 0 ? (atoi(e) > 8 ? 8 : atoi(e)) : 3;  // = the staging slots of a context's prove path (IO_SLOTS)
    auto* w = new PoolWorkers();
    pool->pool_workers = w;
    for (bbp_ctx* member : pool->members) {
        w->m.emplace_back(new PoolWorkers::Member());
        PoolWorkers::Member* mem = w->m.back().get();
        mem->numa_node = gpu_numa_node(member->device);
        for (int k = 0; k < per; k++) {
            try {
                mem->th.emplace_back(pool_worker_main, mem, member);
            } catch (...) {  // fewer threads (or none: for_each_block then runs that member's block on the calling thread)
                break;
            }
        }
    }
}
void pool_workers_stop(bbp_ctx* pool) {
    delete static_cast<PoolWorkers*>(pool->pool_workers);  // runs what is queued, joins
    pool->pool_workers = nullptr;
}
int pool_member_numa_node(const bbp_ctx* pool, uint32_t i) {
    const auto* w = static_cast<const PoolWorkers*>(pool->pool_workers);
    return w && i < w->m.size() ? w->m[i]->numa_node : -1;
}

int32_t pool_reject(bbp_ctx* pool, const char* what) {
    try {
        std::lock_guard<std::recursive_mutex> lk(pool->mu);
        pool->err = std::string(what) + ": not available on a pool handle (device pointers and streams belong to ONE device: use bbp_pool_member)";
        set_tls_error(pool, pool->err);
    } catch (...) {
    }
    return BBP_ERR_BAD_ARG;
}

// [lo, hi) of member i when B items are block-split over n members: the same arithmetic as sharding.shard_range
static void shard_range(uint32_t B, uint32_t i, uint32_t n, uint32_t* lo, uint32_t* hi) {
    const uint32_t base = B / n, rem = B % n;
    *lo = i * base + (i < rem ? i : rem);
    *hi = *lo + base + (i < rem ? 1u : 0u);
}

// run body(member, lo, hi) for every member with a non-empty block ON THAT MEMBER'S OWN WORKER THREADS (bound to its GPU, see
// PoolWorkers; no thread is created per call); the calling thread waits.  Returns the first non-zero status in member order and
// leaves that member's message in this thread's bbp_last_error slot
template <class F>
static int32_t for_each_block(bbp_ctx* pool, uint32_t B, F&& body) {
    const uint32_t n = (uint32_t)pool->members.size();
    std::vector<int32_t> rc(n, BBP_OK);
    std::vector<std::string> msg(n);
    auto run = [&](uint32_t i) {
        uint32_t lo, hi;
        shard_range(B, i, n, &lo, &hi);
        if (lo == hi) return;
        try {
            rc[i] = body(pool->members[i], lo, hi);
            if (rc[i] != BBP_OK) msg[i] = bbp_last_error(pool->members[i]);  // the worker thread's slot
        } catch (...) {
            rc[i] = BBP_ERR_INTERNAL;
            msg[i] = "internal error in a pool worker";
        }
    };
    std::mutex done_mu;
    std::condition_variable done_cv;
    uint32_t pending = 0;
    auto* w = static_cast<PoolWorkers*>(pool->pool_workers);
    std::vector<uint32_t> inline_blocks;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t lo, hi;
        shard_range(B, i, n, &lo, &hi);
        if (lo == hi) continue;
        bool posted = false;
        if (w && i < w->m.size() && !w->m[i]->th.empty()) {
            try {
                {
                    std::lock_guard<std::mutex> lk(done_mu);
                    pending++;
                }
                {
                    std::lock_guard<std::mutex> lk(w->m[i]->mu);
                    w->m[i]->q.emplace_back([&, i] {
                        run(i);
                        std::lock_guard<std::mutex> lk2(done_mu);  // (notify under the lock: the waiter's frame outlives this job)
                        pending--;
                        done_cv.notify_one();
                    });
                }
                w->m[i]->cv.notify_one();
                posted = true;
            } catch (...) {  // the queue could not take it
                std::lock_guard<std::mutex> lk(done_mu);
                pending--;
            }
        }
        if (!posted) inline_blocks.push_back(i);
    }
    for (uint32_t i : inline_blocks) run(i);  // (a member without workers: on the calling thread; the member's entry points set the device)
    {
        std::unique_lock<std::mutex> lk(done_mu);
        done_cv.wait(lk, [&] { return pending == 0; });
    }
    for (uint32_t i = 0; i < n; i++)
        if (rc[i] != BBP_OK) {
            try {
                std::lock_guard<std::recursive_mutex> lk(pool->mu);
                pool->err = "member " + std::to_string(i) + " (device " + std::to_string(pool->members[i]->device) + "): " + msg[i];
                set_tls_error(pool, pool->err);
            } catch (...) {
            }
            return rc[i];
        }
    return BBP_OK;
}

int32_t pool_prove_batch(bbp_ctx* pool, uint32_t B, uint32_t N, const uint8_t* in, const uint8_t* entropy, uint8_t* out, int32_t* status) {
    const size_t in_stride = 7 * 32 + (size_t)N * 32 + 8, ent_stride = bbp_entropy_size(N), out_stride = bbp_proof_record_size(N);
    return for_each_block(pool, B, [&](bbp_ctx* m, uint32_t lo, uint32_t hi) {
        return bbp_prove_batch(m, hi - lo, N, in + in_stride * lo, entropy ? entropy + ent_stride * lo : nullptr, out + out_stride * lo, status + lo);
    });
}

int32_t pool_verify_batch(bbp_ctx* pool, uint32_t B, uint32_t N, const uint8_t* in, int32_t* status, bool aggregated, uint32_t group,
                          uint32_t* n_fallback) {
    const size_t stride = (size_t)bbp_proof_record_size(N) + 96 + (size_t)N * 32;
    std::vector<uint32_t> nfb(pool->members.size(), 0);
    const int32_t rc = for_each_block(pool, B, [&](bbp_ctx* m, uint32_t lo, uint32_t hi) {
        if (!aggregated) return bbp_verify_batch(m, hi - lo, N, in + stride * lo, status + lo);
        return bbp_verify_batch_aggregated(m, hi - lo, N, in + stride * lo, status + lo, group, &nfb[m->member_index]);
    });
    if (n_fallback) {
        *n_fallback = 0;
        for (uint32_t v : nfb) *n_fallback += v;
    }
    return rc;
}

int32_t pool_msm_batch(bbp_ctx* pool, uint32_t B, uint32_t n_terms, const uint8_t* scalars, uint32_t layout, uint8_t* out32) {
    return for_each_block(pool, B, [&](bbp_ctx* m, uint32_t lo, uint32_t hi) {
        return bbp_msm_batch(m, hi - lo, n_terms, scalars + (size_t)32 * n_terms * lo, layout, out32 + (size_t)32 * lo);
    });
}

}  // namespace bbp

using namespace bbp;

extern "C" int32_t bbp_pool_init(const int32_t* devices, uint32_t n_devices, bbp_ctx** out) {
    if (!out) return BBP_ERR_BAD_ARG;
    *out = nullptr;
    if (!devices || n_devices == 0 || n_devices > 64) return BBP_ERR_BAD_ARG;
    own_hw_queues();  // before the first HIP call (setup.hip)
    bbp_ctx* pool = nullptr;
    try {
        pool = new bbp_ctx();
        pool->device = -1;
        *out = pool;  // returned even on failure so that bbp_last_error works; the caller frees it
        std::vector<bbp_ctx*> made(n_devices, nullptr);
        std::vector<int32_t> rc(n_devices, BBP_OK);
        std::vector<std::string> msg(n_devices);
        // every member derives and uploads its own tables (~1 s each): all at once
        auto one = [&](uint32_t i) {
            rc[i] = bbp_init(devices[i], &made[i]);
            if (rc[i] != BBP_OK) msg[i] = made[i] ? bbp_last_error(made[i]) : "no usable HIP device";
        };
        {
            std::vector<std::thread> th;
            std::vector<uint32_t> inline_members;
            for (uint32_t i = 1; i < n_devices; i++) {
                try {
                    th.emplace_back(one, i);
                } catch (...) {
                    inline_members.push_back(i);
                }
            }
            one(0);
            for (uint32_t i : inline_members) one(i);
            for (auto& t : th) t.join();
        }
        for (uint32_t i = 0; i < n_devices; i++)
            if (rc[i] != BBP_OK) {
                pool->err = "bbp_pool_init: member " + std::to_string(i) + " (device " + std::to_string(devices[i]) + "): " + msg[i];
                set_tls_error(pool, pool->err);
                for (bbp_ctx* m : made)
                    if (m) bbp_free(m);
                return rc[i];
            }
        for (uint32_t i = 0; i < n_devices; i++) {
            made[i]->owner = pool;
            made[i]->member_index = i;
        }
        pool->members = made;
        pool_workers_start(pool);
        Combiner* c = new Combiner();
        pool->combiner = c;
        c->set_targets(made);
        if (const char* e = getenv("BBP_BATCH_WINDOW_US")) c->configure((uint32_t)atoi(e), 0);
        const char* e = getenv("BBP_BATCH_STAGGER_US");
        c->set_stagger(e ? (uint32_t)atoi(e) : 35000u);
        c->set_leaders(1, bbp_ctx::VLANES);
        if (const char* lw = getenv("BBP_BATCH_LOPSIDED_WAIT")) c->set_lopsided_wait(atoi(lw) != 0);
        if (const char* pl = getenv("BBP_BATCH_PROVE_LEADERS")) c->set_leaders(0, atoi(pl));
        const char* ss = getenv("BBP_BATCH_STAGGER_SMALL_US");
        c->set_small_stagger(256, ss ? (uint32_t)atoi(ss) : 15000u);
        const char* sp = getenv("BBP_BATCH_SPLIT_MIN");
        c->set_split_min(sp ? (uint32_t)atoi(sp) : 1024u);
        const char* qc = getenv("BBP_BATCH_QUIET_CAP_US");
        const char* qu = getenv("BBP_BATCH_QUIET_US");
        c->set_quiet(qu ? (uint32_t)atoi(qu) : 300u, qc ? (uint32_t)atoi(qc) : 8000u);
        const char* hm = getenv("BBP_BATCH_HOLD_MARGIN_US");
        c->set_hold(hm ? atoi(hm) : 4000, 40000u, 48.0, !getenv("BBP_BATCH_HOLD_FIXED"));
        return BBP_OK;
    } catch (const std::exception& e) {
        if (pool) pool->err = std::string("bbp_pool_init: ") + e.what();
        return BBP_ERR_INTERNAL;
    } catch (...) {
        return BBP_ERR_INTERNAL;
    }
}

extern "C" int32_t bbp_init_all(bbp_ctx** out) {
    if (!out) return BBP_ERR_BAD_ARG;
    *out = nullptr;
    own_hw_queues();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        fprintf(stderr, "bbp_init_all: no usable HIP device (count=%d); this library has no CPU path\n", n);
        return BBP_ERR_DEVICE;
    }
    std::vector<int32_t> devs;
    for (int i = 0; i < n && i < 64; i++) devs.push_back(i);
    return bbp_pool_init(devs.data(), (uint32_t)devs.size(), out);
}

extern "C" uint32_t bbp_pool_size(const bbp_ctx* ctx) { return ctx ? (uint32_t)ctx->members.size() : 0u; }

extern "C" bbp_ctx* bbp_pool_member(bbp_ctx* ctx, uint32_t i) { return ctx && i < ctx->members.size() ? ctx->members[i] : nullptr; }

extern "C" int32_t bbp_pool_member_stats(bbp_ctx* ctx, uint32_t i, uint64_t* n_calls, uint64_t* n_requests) {
    if (!is_pool(ctx) || i >= ctx->members.size() || !ctx->combiner) return BBP_ERR_BAD_ARG;
    try {
        static_cast<Combiner*>(ctx->combiner)->target_stats(i, n_calls, n_requests);
        return BBP_OK;
    } catch (...) {
        return BBP_ERR_INTERNAL;
    }
}
