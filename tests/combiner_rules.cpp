// TEST INFRASTRUCTURE: the call combiner's batching rules (dusk_blindbidproof_amd/csrc/submit.cpp, the product's own code) against a
// stand-in engine with CLOSED-LOOP callers: every completed request is submitted again from its completion hook, the way a client
// on a socket sends its next request when the reply arrives.  The stand-in takes 2 ms + 20 us per proof per batch, batches overlap.
// The run starts in the lopsided state the UDS server was seen to fall into (a small batch in flight beside a large one: the small
// one's callers come back early, are sent again at once, come back early again ...).  argv[1] = 1 / 0: the lopsided-pair rule on /
// off.  Prints the sizes of the batches of the run's second half; exit code 0 = every answer right.
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "../dusk_blindbidproof_amd/csrc/submit.h"

using clk = std::chrono::steady_clock;
struct bbp_ctx {
    std::mutex m;
    std::vector<std::pair<double, uint32_t>> log;  // (start ms, size) of every prove batch
    clk::time_point t0 = clk::now();
};

namespace bbp {
int32_t prove_batch_locked(bbp_ctx* c, uint32_t B, uint32_t N, const uint8_t* in, const uint8_t*, uint8_t* out, int32_t* status, std::string*) {
    {
        std::lock_guard<std::mutex> lk(c->m);
        c->log.push_back({std::chrono::duration<double, std::milli>(clk::now() - c->t0).count(), B});
    }
    usleep(2000 + 20 * B);
    const size_t stride = 7 * 32 + 32 * (size_t)N + 8, rec = 1121 + 32 * (4 + (size_t)N);
    for (uint32_t i = 0; i < B; i++) {
        status[i] = 0;
        memset(out + rec * i, in[stride * i], rec);
    }
    return 0;
}
int32_t verify_batch_locked(bbp_ctx*, uint32_t B, uint32_t, uint32_t, const uint8_t*, int32_t* status, std::string*) {
    for (uint32_t i = 0; i < B; i++) status[i] = 0;
    return 0;
}
}  // namespace bbp

static bbp_ctx g_ctx;
static bbp::Combiner* g_comb;
static std::atomic<bool> g_stop{false};
static std::atomic<int> g_wrong{0}, g_live{0}, g_done{0};
static void again(bbp::Request* r) {
    if (r->status != 0 || r->out[0] != r->own_in[0]) g_wrong++;
    g_done++;
    if (g_stop.load()) {
        delete[] r->out;
        delete r;
        g_live--;
        return;
    }
    if (!g_comb->submit_async(&g_ctx, r)) g_wrong++;
}
static void client(int tag) {
    bbp::Request* r = new bbp::Request();
    const uint32_t N = 3;
    r->own_in.assign(7 * 32 + 32 * N + 8, (uint8_t)tag);
    r->kind = 0;
    r->N = N;
    r->in = r->own_in.data();
    r->in_len = r->own_in.size();
    r->out = new uint8_t[1121 + 32 * (4 + N)];
    r->on_done = again;
    g_live++;
    if (!g_comb->submit_async(&g_ctx, r)) g_wrong++;
}

int main(int argc, char** argv) {
    const bool rule = argc < 2 || atoi(argv[1]) != 0;
    {
        bbp::Combiner comb;
        g_comb = &comb;
        comb.configure(100, 4096);
        comb.set_stagger(1000);
        comb.set_small_stagger(64, 500);
        comb.set_hold(400, 2000, 20.0);  // the stand-in's opening stage: 2 ms; 20 us per proof
        comb.set_quiet(300, 2000);
        comb.set_split_min(128);
        comb.set_lopsided_wait(rule);
        for (int i = 0; i < 100; i++) client(i);      // a small batch goes out alone ...
        usleep(600);
        for (int i = 100; i < 600; i++) client(i);    // ... and the large one follows while it is in flight
        usleep(400 * 1000);
        g_stop = true;
        for (int k = 0; k < 2000 && g_live.load() > 0; k++) usleep(1000);
    }  // ~Combiner
    std::vector<uint32_t> tail;
    for (auto& e : g_ctx.log)
        if (e.first >= 200.0) tail.push_back(e.second);
    uint32_t mn = ~0u, mx = 0;
    for (uint32_t b : tail) {
        mn = b < mn ? b : mn;
        mx = b > mx ? b : mx;
    }
    printf("rule %d: %d requests served in %zu batches; second half: %zu batches, sizes %u..%u:", (int)rule, g_done.load(), g_ctx.log.size(), tail.size(), mn, mx);
    for (size_t i = 0; i < tail.size() && i < 24; i++) printf(" %u", tail[i]);
    printf("\nRESULT min %u max %u wrong %d live %d\n", tail.empty() ? 0 : mn, mx, g_wrong.load(), g_live.load());
    return g_wrong.load() || g_live.load() ? 1 : 0;
}
