// TEST INFRASTRUCTURE: the call combiner (dusk_blindbidproof_amd/csrc/submit.cpp, the product's own code) under ThreadSanitizer.
// 48 threads submit prove / verify requests of three classes through one Combiner -- once in front of one engine, once in front
// of three (a device pool: batches dealt to the least-loaded target); the "engine" behind it is a stand-in that
// checks batch consistency (one class per batch, max_batch respected, never more than two batches inside at once), sleeps a
// little and answers with a function of the request.  Exit code 0 = every answer right, invariants held, no data race reported.
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../dusk_blindbidproof_amd/csrc/submit.h"

struct bbp_ctx {
    std::atomic<int> inside[2] = {{0}, {0}}, max_inside{0}, bad{0};  // per kind: the combiner runs at most two batches of a kind per target
    uint32_t max_batch = 8;
};

static void enter(bbp_ctx* c, uint32_t B, int kind) {
    const int n = ++c->inside[kind];
    int m = c->max_inside.load();
    while (n > m && !c->max_inside.compare_exchange_weak(m, n)) {
    }
    if (B == 0 || B > c->max_batch) c->bad++;
    usleep(300);
}

namespace bbp {
int32_t prove_batch_locked(bbp_ctx* c, uint32_t B, uint32_t N, const uint8_t* in, const uint8_t*, uint8_t* out, int32_t* status, std::string*) {
    enter(c, B, 0);
    const size_t stride = 7 * 32 + 32 * (size_t)N + 8, rec = 1121 + 32 * (4 + (size_t)N);
    for (uint32_t i = 0; i < B; i++) {
        status[i] = 0;
        memset(out + rec * i, in[stride * i], rec);  // the record echoes the request's first byte
    }
    --c->inside[0];
    return 0;
}
int32_t verify_batch_locked(bbp_ctx* c, uint32_t B, uint32_t N, uint32_t, const uint8_t* in, int32_t* status, std::string*) {
    enter(c, B, 1);
    const size_t stride = 1121 + 32 * (4 + (size_t)N) + 96 + 32 * (size_t)N;
    for (uint32_t i = 0; i < B; i++) status[i] = in[stride * i] & 1;  // "verdict" = low bit of the first byte
    --c->inside[1];
    return 0;
}
}  // namespace bbp

// one scenario: `n_targets` stand-in engines behind ONE combiner (n_targets == 1: a plain context, the target is whatever submit is
// handed; more: a device pool -- set_targets)
static int scenario(int n_targets) {
    std::vector<bbp_ctx> ctxs(n_targets);
    bbp::Combiner comb;
    comb.configure(100, ctxs[0].max_batch);
    comb.set_stagger(500);
    if (n_targets > 1) {
        std::vector<bbp_ctx*> t;
        for (auto& c : ctxs) t.push_back(&c);
        comb.set_targets(t);
    }
    std::atomic<int> wrong{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 48; t++)
        th.emplace_back([&, t] {
            for (int j = 0; j < 40; j++) {
                const uint32_t N = 2 + (uint32_t)((t + j) % 3);
                const uint8_t tag = (uint8_t)(t * 5 + j);
                bbp::Request r;
                std::vector<uint8_t> in, out(1121 + 32 * (4 + N));
                if ((t + j) & 1) {
                    in.assign(7 * 32 + 32 * N + 8, tag);
                    r.kind = 0;
                    r.out = out.data();
                } else {
                    in.assign(1121 + 32 * (4 + N) + 96 + 32 * N, tag);
                    r.kind = 1;
                }
                r.N = N;
                r.in = in.data();
                r.in_len = in.size();
                const int32_t st = comb.submit(&ctxs[0], r);
                if (r.kind == 0 ? (st != 0 || out[0] != tag || out[out.size() - 1] != tag) : st != (tag & 1)) wrong++;
            }
        });
    for (auto& x : th) x.join();
    uint64_t calls = 0, reqs = 0;
    uint32_t biggest = 0;
    comb.stats(&calls, &reqs, &biggest);
    int max_inside = 0, bad = 0, idle_targets = 0;
    uint64_t per_target_sum = 0;
    for (int i = 0; i < n_targets; i++) {
        if (ctxs[i].max_inside.load() > max_inside) max_inside = ctxs[i].max_inside.load();
        bad += ctxs[i].bad.load();
        uint64_t c = 0, q = 0;
        comb.target_stats((size_t)i, &c, &q);
        per_target_sum += q;
        idle_targets += c == 0;
    }
    printf("targets %d: requests %llu in %llu combined calls, largest %u, max concurrent per target %d, wrong %d, bad batches %d, unused targets %d\n",
           n_targets, (unsigned long long)reqs, (unsigned long long)calls, biggest, max_inside, wrong.load(), bad, idle_targets);
    // per target never more than two batches inside at once; every target of a pool gets work; the per-target counters add up
    return (wrong.load() || bad || reqs != 48 * 40 || per_target_sum != reqs || max_inside > 2 || biggest > ctxs[0].max_batch || idle_targets) ? 1 : 0;
}

// destruction with asynchronous requests still queued: the combiner runs them first, every completion hook fires exactly once
static std::atomic<int> g_hooks{0}, g_hook_bad{0};
static void hook(bbp::Request* r) {
    if (r->status != 0 || r->out[0] != r->own_in[0]) g_hook_bad++;
    g_hooks++;
    delete[] r->out;
    delete r;
}
static int scenario_drain() {
    bbp_ctx ctx;
    const int n = 300;
    {
        bbp::Combiner comb;
        comb.configure(2000, ctx.max_batch);  // a long window: most requests are still queued when the destructor runs
        comb.set_stagger(500);
        for (int i = 0; i < n; i++) {
            bbp::Request* r = new bbp::Request();
            const uint32_t N = 3;
            r->own_in.assign(7 * 32 + 32 * N + 8, (uint8_t)i);
            r->kind = 0;
            r->N = N;
            r->in = r->own_in.data();
            r->in_len = r->own_in.size();
            r->out = new uint8_t[1121 + 32 * (4 + N)];
            r->on_done = hook;
            if (!comb.submit_async(&ctx, r)) return 1;
        }
    }  // ~Combiner
    printf("drain: %d hooks of %d fired before the destructor returned, %d wrong\n", g_hooks.load(), n, g_hook_bad.load());
    return g_hooks.load() == n && g_hook_bad.load() == 0 ? 0 : 1;
}

int main() { return scenario(1) | scenario(3) | scenario_drain(); }
