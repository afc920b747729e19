// TEST INFRASTRUCTURE -- not a CPU path of the product.  A stand-in "engine" exporting the handful of include/bbp.h symbols the
// UDS server binds, so the CPU tier can exercise the server's framing, dispatch, error behaviour and micro-batching without a
// GPU.  It proves nothing: a "proof" is a tag derived from the public inputs repeated 1120 times, "verification" recomputes the
// tag.  The call combiner in front of it is the product's own (csrc/submit.cpp, compiled into this library unchanged).
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <vector>

#include "../dusk_blindbidproof_amd/csrc/submit.h"
#include "../include/bbp.h"

struct bbp_ctx {
    bbp::Combiner combiner;
    std::vector<bbp_ctx*> members;  // a stub pool: fake member contexts behind this handle's combiner (bbp_pool_init)
    std::atomic<uint64_t> batch_calls{0};
    std::atomic<int> inside[2] = {{0}, {0}}, max_inside{0};  // combined calls of one kind running at the same time (the combiner allows two per kind)
};
struct Inside {
    bbp_ctx* c;
    int kind;
    Inside(bbp_ctx* ctx, int k) : c(ctx), kind(k) {
        const int n = ++c->inside[kind];
        int m = c->max_inside.load();
        while (n > m && !c->max_inside.compare_exchange_weak(m, n)) {
        }
    }
    ~Inside() { --c->inside[kind]; }
};

static thread_local std::string t_err;

static uint8_t tag_of(const uint8_t* q, const uint8_t* z, const uint8_t* seed, const uint8_t* pub, uint32_t N) {
    unsigned s = 0x5a;
    for (int i = 0; i < 32; i++) s += q[i] * 3u + z[i] * 5u + seed[i] * 7u;
    for (uint32_t i = 0; i < 32 * N; i++) s += pub[i] * (1u + (i & 3u));
    return (uint8_t)(s | 1u);
}
static bool canonical(const uint8_t* s) {  // < 2^253 is enough for a stub: the real check is the engine's
    return (s[31] & 0xe0) == 0;
}

// STUB_DEVICE_FAIL_AFTER=n: the n-th combined call and every later one return BBP_ERR_DEVICE (the real engine's sticky health word)
static std::atomic<int> g_calls{0};
static bool device_dead(std::string* err) {
    static const int after = getenv("STUB_DEVICE_FAIL_AFTER") ? atoi(getenv("STUB_DEVICE_FAIL_AFTER")) : 0;
    if (!after || ++g_calls < after) return false;
    if (err) *err = "stub: device health flag raised";
    return true;
}

namespace bbp {
int32_t prove_batch_locked(bbp_ctx* ctx, uint32_t B, uint32_t N, const uint8_t* in, const uint8_t*, uint8_t* out, int32_t* status, std::string* err) {
    if (device_dead(err)) return BBP_ERR_DEVICE;
    ctx->batch_calls++;
    Inside in_call(ctx, 0);
    static const int prove_us = getenv("STUB_PROVE_US") ? atoi(getenv("STUB_PROVE_US")) : 3000;
    usleep(prove_us);  // one "device call" costs the same whatever B is: combining is what pays
    const size_t in_stride = 7 * 32 + 32 * (size_t)N + 8, rec = 1121 + 32 * (4 + (size_t)N);
    for (uint32_t i = 0; i < B; i++) {
        const uint8_t* r = in + in_stride * i;
        uint64_t toggle;
        memcpy(&toggle, r + 7 * 32 + 32 * (size_t)N, 8);
        status[i] = BBP_OK;
        for (int k = 0; k < 7; k++)
            if (!canonical(r + 32 * k)) status[i] = BBP_ERR_FORMAT;
        if (toggle >= N) status[i] = BBP_ERR_BAD_ARG;
        if (status[i] != BBP_OK) continue;
        uint8_t* o = out + rec * i;
        const uint8_t t = tag_of(r + 32 * 4, r + 32 * 5, r + 32 * 6, r + 7 * 32, N);
        o[0] = 0;
        memset(o + 1, t, 1120);
        for (uint32_t c = 0; c < 4 + N; c++) memset(o + 1121 + 32 * c, (uint8_t)(0xc0 + c), 32);
    }
    return BBP_OK;
}
int32_t verify_batch_locked(bbp_ctx* ctx, uint32_t B, uint32_t N, uint32_t rec_ver, const uint8_t* in, int32_t* status, std::string* err) {
    if (device_dead(err)) return BBP_ERR_DEVICE;
    ctx->batch_calls++;
    Inside in_call(ctx, 1);
    usleep(1000);
    const size_t rec = (rec_ver ? 1217 : 1121) + 32 * (4 + (size_t)N), stride = rec + 96 + 32 * (size_t)N;
    for (uint32_t i = 0; i < B; i++) {
        const uint8_t* r = in + stride * i;
        const uint8_t t = tag_of(r + rec, r + rec + 32, r + rec + 64, r + rec + 96, N);
        status[i] = BBP_OK;
        for (int k = 1; k < 1121; k++)
            if (r[k] != t) status[i] = BBP_ERR_VERIFY;
    }
    return BBP_OK;
}
}  // namespace bbp

extern "C" {
int32_t bbp_pool_init(const int32_t* devices, uint32_t n, bbp_ctx** out);
int32_t bbp_init(int32_t device, bbp_ctx** out) {
    if (device == -1) {  // "every visible GPU": the stub pretends there are two
        const int32_t two[2] = {0, 1};
        return bbp_pool_init(two, 2, out);
    }
    *out = new bbp_ctx();
    return BBP_OK;
}
void bbp_free(bbp_ctx* c) {
    if (!c) return;
    std::vector<bbp_ctx*> members = c->members;
    delete c;  // joins the combiner's threads first: nothing runs on a member afterwards
    for (bbp_ctx* m : members) delete m;
}
int32_t bbp_pool_init(const int32_t* devices, uint32_t n, bbp_ctx** out) {
    if (!devices || !n || !out) return BBP_ERR_BAD_ARG;
    bbp_ctx* p = new bbp_ctx();
    for (uint32_t i = 0; i < n; i++) p->members.push_back(new bbp_ctx());
    p->combiner.set_targets(p->members);
    *out = p;
    return BBP_OK;
}
uint32_t bbp_pool_size(const bbp_ctx* c) { return c ? (uint32_t)c->members.size() : 0; }
int32_t bbp_pool_member_stats(bbp_ctx* c, uint32_t i, uint64_t* calls, uint64_t* reqs) {
    if (!c || i >= c->members.size()) return BBP_ERR_BAD_ARG;
    c->combiner.target_stats(i, calls, reqs);
    return BBP_OK;
}
const char* bbp_last_error(const bbp_ctx*) { return t_err.c_str(); }
uint32_t bbp_proof_record_size(uint32_t N) { return 1121 + 32 * (4 + N); }
int32_t bbp_set_batching(bbp_ctx* c, uint32_t w, uint32_t m) {
    c->combiner.configure(w, m);
    return BBP_OK;
}
int32_t bbp_batching_stats(bbp_ctx* c, uint64_t* a, uint64_t* b, uint32_t* m) {
    c->combiner.stats(a, b, m);
    return BBP_OK;
}
int32_t bbp_reserve(bbp_ctx*, uint32_t, uint32_t N) { return N == 0 ? BBP_ERR_BAD_ARG : N > BBP_MAX_ITEMS ? BBP_ERR_GENS_LEN : BBP_OK; }
int32_t bbp_check_health(bbp_ctx*, uint32_t* flags) {
    if (flags) *flags = 0;
    return BBP_OK;
}
int32_t stub_max_concurrency(bbp_ctx* c) {  // per device: the largest number of combined calls any one (member) context saw at once
    int m = c->max_inside.load();
    for (bbp_ctx* x : c->members)
        if (x->max_inside.load() > m) m = x->max_inside.load();
    return m;
}
int32_t bbp_prove(bbp_ctx* c, const uint8_t s7[7 * 32], const uint8_t* pub, uint32_t N, uint64_t toggle, const uint8_t* ent, uint8_t* out,
                  uint32_t* plen) {
    if (N == 0) return BBP_ERR_BAD_ARG;
    if (N > BBP_MAX_ITEMS) return BBP_ERR_GENS_LEN;
    std::string in((const char*)s7, 7 * 32);
    in.append((const char*)pub, 32 * (size_t)N);
    in.append((const char*)&toggle, 8);
    bbp::Request r;
    r.kind = 0;
    r.N = N;
    r.in = (const uint8_t*)in.data();
    r.in_len = in.size();
    r.entropy = ent;
    r.out = out;
    const int32_t st = c->combiner.submit(c, r);
    if (st != BBP_OK) t_err = "stub: rejected input";
    if (st == BBP_OK && plen) *plen = 1121;
    return st;
}
int32_t bbp_verify(bbp_ctx* c, const uint8_t* rec, uint32_t rec_len, const uint8_t score[32], const uint8_t z[32], const uint8_t seed[32],
                   const uint8_t* pub, uint32_t N) {
    if (rec_len != 1121 + 32 * (4 + N)) return BBP_ERR_FORMAT;
    if (!canonical(score) || !canonical(z) || !canonical(seed)) return BBP_ERR_FORMAT;
    std::string in((const char*)rec, rec_len);
    in.append((const char*)score, 32).append((const char*)z, 32).append((const char*)seed, 32).append((const char*)pub, 32 * (size_t)N);
    bbp::Request r;
    r.kind = 1;
    r.N = N;
    r.in = (const uint8_t*)in.data();
    r.in_len = in.size();
    return c->combiner.submit(c, r);
}

static void stub_done(bbp::Request* r) {
    void (*fn)(void*, int32_t) = r->user_fn;
    void* user = r->user;
    const int32_t st = r->status;
    if (st != BBP_OK) t_err = "stub: rejected input";
    delete r;
    fn(user, st);
}
int32_t bbp_prove_async(bbp_ctx* c, const uint8_t s7[7 * 32], const uint8_t* pub, uint32_t N, uint64_t toggle, const uint8_t* ent, uint8_t* out,
                        bbp_done_fn done, void* user) {
    if (N == 0) return BBP_ERR_BAD_ARG;
    if (N > BBP_MAX_ITEMS) return BBP_ERR_GENS_LEN;
    bbp::Request* r = new bbp::Request();
    r->own_in.assign(s7, s7 + 7 * 32);
    r->own_in.insert(r->own_in.end(), pub, pub + 32 * (size_t)N);
    r->own_in.insert(r->own_in.end(), (const uint8_t*)&toggle, (const uint8_t*)&toggle + 8);
    r->kind = 0;
    r->N = N;
    r->in = r->own_in.data();
    r->in_len = r->own_in.size();
    r->entropy = ent;
    r->out = out;
    r->on_done = stub_done;
    r->user_fn = done;
    r->user = user;
    if (!c->combiner.submit_async(c, r)) {
        delete r;
        return BBP_ERR_INTERNAL;
    }
    return BBP_OK;
}
int32_t bbp_verify_async(bbp_ctx* c, const uint8_t* rec, uint32_t rec_len, const uint8_t score[32], const uint8_t z[32], const uint8_t seed[32],
                         const uint8_t* pub, uint32_t N, bbp_done_fn done, void* user) {
    if (rec_len != 1121 + 32 * (4 + N)) return BBP_ERR_FORMAT;
    if (!canonical(score) || !canonical(z) || !canonical(seed)) return BBP_ERR_FORMAT;
    bbp::Request* r = new bbp::Request();
    r->own_in.assign(rec, rec + rec_len);
    r->own_in.insert(r->own_in.end(), score, score + 32);
    r->own_in.insert(r->own_in.end(), z, z + 32);
    r->own_in.insert(r->own_in.end(), seed, seed + 32);
    r->own_in.insert(r->own_in.end(), pub, pub + 32 * (size_t)N);
    r->kind = 1;
    r->N = N;
    r->in = r->own_in.data();
    r->in_len = r->own_in.size();
    r->on_done = stub_done;
    r->user_fn = done;
    r->user = user;
    if (!c->combiner.submit_async(c, r)) {
        delete r;
        return BBP_ERR_INTERNAL;
    }
    return BBP_OK;
}
}
