// TEST INFRASTRUCTURE: the call combiner (dusk_blindbidproof_amd/csrc/submit.cpp, the product's own code) under ThreadSanitizer.
// 48 threads submit prove / verify requests of three classes through one Combiner; the "engine" behind it is a stand-in that
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
    std::atomic<int> inside{0}, max_inside{0}, bad{0};
    uint32_t max_batch = 8;
};

static void enter(bbp_ctx* c, uint32_t B) {
    const int n = ++c->inside;
    int m = c->max_inside.load();
    while (n > m && !c->max_inside.compare_exchange_weak(m, n)) {
    }
    if (B == 0 || B > c->max_batch) c->bad++;
    usleep(300);
}

namespace bbp {
int32_t prove_batch_locked(bbp_ctx* c, uint32_t B, uint32_t N, const uint8_t* in, const uint8_t*, uint8_t* out, int32_t* status, std::string*) {
    enter(c, B);
    const size_t stride = 7 * 32 + 32 * (size_t)N + 8, rec = 1121 + 32 * (4 + (size_t)N);
    for (uint32_t i = 0; i < B; i++) {
        status[i] = 0;
        memset(out + rec * i, in[stride * i], rec);  // the record echoes the request's first byte
    }
    --c->inside;
    return 0;
}
int32_t verify_batch_locked(bbp_ctx* c, uint32_t B, uint32_t N, uint32_t, const uint8_t* in, int32_t* status, std::string*) {
    enter(c, B);
    const size_t stride = 1121 + 32 * (4 + (size_t)N) + 96 + 32 * (size_t)N;
    for (uint32_t i = 0; i < B; i++) status[i] = in[stride * i] & 1;  // "verdict" = low bit of the first byte
    --c->inside;
    return 0;
}
}  // namespace bbp

int main() {
    bbp_ctx ctx;
    bbp::Combiner comb;
    comb.configure(100, ctx.max_batch);
    comb.set_stagger(500);
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
                const int32_t st = comb.submit(&ctx, r);
                if (r.kind == 0 ? (st != 0 || out[0] != tag || out[out.size() - 1] != tag) : st != (tag & 1)) wrong++;
            }
        });
    for (auto& x : th) x.join();
    uint64_t calls = 0, reqs = 0;
    uint32_t biggest = 0;
    comb.stats(&calls, &reqs, &biggest);
    printf("requests %llu in %llu combined calls, largest %u, max concurrent %d, wrong %d, bad batches %d\n", (unsigned long long)reqs,
           (unsigned long long)calls, biggest, ctx.max_inside.load(), wrong.load(), ctx.bad.load());
    return (wrong.load() || ctx.bad.load() || reqs != 48 * 40 || ctx.max_inside.load() > 2 || biggest > ctx.max_batch) ? 1 : 0;
}
