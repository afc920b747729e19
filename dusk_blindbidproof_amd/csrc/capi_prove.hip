// C-ABI: prove / verify / witness entry points -- see include/bbp.h for the reference interface each replaces.
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <thread>

#include "batch.h"
#include "submit.h"

namespace bbp {
int32_t prove_batch_dev(bbp_ctx* ctx, u32 B, u32 N, const u8* in_dev, const u8* ent_dev, u8* out_dev, hipStream_t s);
int32_t verify_batch_dev(bbp_ctx* ctx, u32 B, u32 N, const u8* in_dev, const u8* ent_dev, int32_t* status_dev, hipStream_t s);
int32_t verify_batch_dev_ex(bbp_ctx* ctx, u32 B, u32 N, u32 rec_ver, u32 G, const u8* in_dev, const u8* ent_dev, int32_t* status_dev, hipStream_t s);
int32_t verify_batch_agg_dev(bbp_ctx* ctx, u32 B, u32 N, u32 G, const u8* in_dev, const u8* ent_dev, int32_t* status_dev, hipStream_t s,
                             u32* n_fallback, u32* total_out_dev = nullptr);
int32_t debug_read_misc(bbp_ctx* ctx, u32 B, u32 N, u32 proof, uint8_t* out);

// native (non-circuit) image of the gadget wiring: what the reference's Go caller computes before Proof::prove
// (src/gadgets.rs:20-33 for m,x,y,z; :70-86 for y_inv and q)
__device__ sc mimc_native(sc x, const sc& key, const sc* __restrict__ c) {
    for (int i = 0; i < BBP_MIMC_ROUNDS; i++) {
        sc a = sc_add(sc_add(x, key), ld_sc(&c[i]));
        sc a2 = sc_mul(a, a), a3 = sc_mul(a2, a), a4 = sc_mul(a2, a2);
        x = sc_mul(a4, a3);
    }
    return sc_add(x, key);
}

__global__ void k_witness_native(u32 B, const u8* __restrict__ dks, const sc* __restrict__ mimc, u8* __restrict__ out) {
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    const u32* in = reinterpret_cast<const u32*>(dks + 96 * (size_t)p);
    sc d = sc_reduce256(in), k = sc_reduce256(in + 8), seed = sc_reduce256(in + 16);
    sc m = mimc_native(k, sc_zero(), mimc);
    sc x = mimc_native(d, m, mimc);
    sc y = mimc_native(seed, x, mimc);
    sc z = mimc_native(seed, m, mimc);
    sc yi = sc_invert(y);
    sc q = sc_mul(d, yi);
    sc* o = reinterpret_cast<sc*>(out + 192 * (size_t)p);
    st_sc(&o[0], m);
    st_sc(&o[1], x);
    st_sc(&o[2], y);
    st_sc(&o[3], yi);
    st_sc(&o[4], q);
    st_sc(&o[5], z);
}

// the caller-side pass (Go, upstream of Proof::prove) on the device, writing the prover's and the verifier's
// input rows directly.  One lane per bid.  bids: d || k || seed (96 B); lists: N x 32 B (entry `toggle` is replaced by the
// bid's own x); toggles: u64.  prove_in row: d,k,y,y_inv,q,z_img,seed || list || toggle; verify_tail row: q || z_img || seed || list.
__global__ void k_prepare_bids(u32 B, u32 N, const u8* __restrict__ bids, const u8* __restrict__ lists, const u64* __restrict__ toggles,
                               const sc* __restrict__ mimc, u32* __restrict__ prove_in, u32* __restrict__ verify_tail) {
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    const u32* in = reinterpret_cast<const u32*>(bids + 96 * (size_t)p);
    sc d = sc_reduce256(in), k = sc_reduce256(in + 8), seed = sc_reduce256(in + 16);
    sc m = mimc_native(k, sc_zero(), mimc);
    sc x = mimc_native(d, m, mimc);
    sc y = mimc_native(seed, x, mimc);
    sc z = mimc_native(seed, m, mimc);
    sc yi = sc_invert(y);
    sc q = sc_mul(d, yi);
    const u64 toggle = toggles[p];
    const size_t pw = 7 * 8 + (size_t)N * 8 + 2, vw = 3 * 8 + (size_t)N * 8;  // row lengths in words
    u32* o = prove_in + pw * p;
    const sc seven[7] = {d, k, y, yi, q, z, seed};
    for (int i = 0; i < 7; i++)
        for (int w = 0; w < 8; w++) o[8 * i + w] = seven[i].v[w];
    o[56 + 8 * N] = (u32)toggle;
    o[56 + 8 * N + 1] = (u32)(toggle >> 32);
    u32* v = verify_tail ? verify_tail + vw * p : nullptr;
    if (v)
        for (int w = 0; w < 8; w++) {
            v[w] = q.v[w];
            v[8 + w] = z.v[w];
            v[16 + w] = seed.v[w];
        }
    const u32* li = reinterpret_cast<const u32*>(lists + 32 * (size_t)N * p);
    for (u32 i = 0; i < N; i++)
        for (int w = 0; w < 8; w++) {
            const u32 word = (u64)i == toggle ? x.v[w] : li[8 * i + w];
            o[56 + 8 * i + w] = word;
            if (v) v[24 + 8 * i + w] = word;
        }
}

// proofs per engine call when a host-pointer batch API is handed more (BBP_HOST_CHUNK_PROVE / BBP_HOST_CHUNK_VERIFY)
static uint32_t env_chunk(const char* name, uint32_t dflt) {
    const char* e = getenv(name);
    const long v = e ? atol(e) : 0;
    return v >= 1 ? (uint32_t)v : dflt;
}
static uint32_t host_chunk_prove() { return env_chunk("BBP_HOST_CHUNK_PROVE", 16384); }
static uint32_t host_chunk_verify() { return env_chunk("BBP_HOST_CHUNK_VERIFY", 32768); }

static bool os_random(uint8_t* buf, size_t n) {
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f) return false;
    size_t got = fread(buf, 1, n, f);
    fclose(f);
    return got == n;
}

}  // namespace bbp

using namespace bbp;

namespace bbp {
std::string& tls_error() {
    static thread_local std::string e;
    return e;
}
const void*& tls_error_owner() {
    static thread_local const void* c = nullptr;
    return c;
}
int32_t fault_injected(const char* site) {
    const char* e = getenv("BBP_FAULT_INJECT");
    if (e && strcmp(e, site) == 0) throw std::runtime_error(std::string("injected fault at ") + site);
    if (e && strcmp(e, "alloc") == 0 && strcmp(site, "prove_batch") == 0) throw std::bad_alloc();
    return 0;
}
// entry points without a context to lock still keep every exception on this side of the boundary
template <class F>
static int32_t no_throw(F&& body, const void* ctx = nullptr) noexcept {
    try {
        return body();
    } catch (const std::invalid_argument& e) {
        try { set_tls_error(ctx, std::string("invalid argument: ") + e.what()); } catch (...) {}
        return BBP_ERR_BAD_ARG;
    } catch (const std::exception& e) {
        try { set_tls_error(ctx, std::string("internal error: ") + e.what()); } catch (...) {}
        return BBP_ERR_INTERNAL;
    } catch (...) {
        set_tls_error(ctx, "internal error: unknown exception");
        return BBP_ERR_INTERNAL;
    }
}
}  // namespace bbp

extern "C" uint32_t bbp_proof_record_size(uint32_t N) { return BBP_R1CS_PROOF_BYTES + 32u * (4u + N); }
extern "C" uint32_t bbp_entropy_size(uint32_t N) { return 32u * (4u + N) + 32u; }

extern "C" int32_t bbp_debug_compile_circuit(uint32_t N, uint32_t* n_mul, uint32_t* n_cons) {
    return no_throw([&]() -> int32_t {
        if (N > BBP_MAX_ITEMS) return BBP_ERR_GENS_LEN;
        fault_injected("compile");
        const circuit::Compiled c = circuit::compile(N);  // throws std::invalid_argument for N == 0 (src/gadgets.rs:103 panics)
        if (n_mul) *n_mul = c.n_mul;
        if (n_cons) *n_cons = c.n_cons;
        return BBP_OK;
    });
}

extern "C" int32_t bbp_witness_batch(bbp_ctx* ctx, uint32_t B, const uint8_t* dks, uint8_t* out) {
    if (!ctx || !dks || !out) return BBP_ERR_BAD_ARG;
    if (B == 0) return BBP_OK;
    if (is_pool(ctx)) return bbp_witness_batch(ctx->members[0], B, dks, out);  // microseconds of work: one member does it
    return api_guard(ctx, [&]() -> int32_t {
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        int32_t rc;
        if ((rc = dev_reserve(ctx, ctx->io_in, 96 * (size_t)B)) || (rc = dev_reserve(ctx, ctx->io_out, 192 * (size_t)B))) return rc;
        StreamGuard guard(ctx, ctx->stream);  // io_in / io_out are shared with the other host-pointer entry points
        if ((rc = guard.enter())) return rc;
        BBP_HIP_TRY(ctx, hipMemcpyAsync(ctx->io_in.p, dks, 96 * (size_t)B, hipMemcpyHostToDevice, ctx->stream));
        {
            ScopedEvent ev(ctx, TAG_WITNESS, ctx->stream);
            hipLaunchKernelGGL(k_witness_native, dim3((B + 63) / 64), dim3(64), 0, ctx->stream, B, (const u8*)ctx->io_in.p, ctx->mimc_c,
                               (u8*)ctx->io_out.p);
            BBP_HIP_TRY(ctx, hipGetLastError());
        }
        BBP_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->io_out.p, 192 * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
        BBP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return BBP_OK;
    });
}

static int32_t check_n(bbp_ctx* ctx, uint32_t N) {
    if (N == 0) {
        ctx->err = "empty bid list (the reference panics at src/gadgets.rs:103)";
        return BBP_ERR_BAD_ARG;
    }
    if (N > BBP_MAX_ITEMS) {
        ctx->err = "bid list needs more than 2048 multipliers (R1CSError::InvalidGeneratorsLength)";
        return BBP_ERR_GENS_LEN;
    }
    return BBP_OK;
}

extern "C" int32_t bbp_prepare_bids_dev(bbp_ctx* ctx, uint32_t B, uint32_t N, const void* bids_dev, const void* lists_dev,
                                        const void* toggles_dev, void* prove_in_dev, void* verify_tail_dev, void* stream) {
    if (!ctx || !bids_dev || !lists_dev || !toggles_dev || !prove_in_dev) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) return pool_reject(ctx, "bbp_prepare_bids_dev");
    return api_guard(ctx, [&]() -> int32_t {
        int32_t rc = check_n(ctx, N);
        if (rc) return rc;
        if (B == 0) return BBP_OK;
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        hipStream_t s = pick_stream(ctx, stream);
        // one lane per bid, 4 x 90 sequential MiMC rounds: a serial wave that lives for milliseconds.  In a pipeline it runs beside
        // another chunk's MSM stage, so it is fenced onto CUs of its own like the prover's opening kernels
        unsigned hog = 0;  // what the launch may ask for: the reservation minus the kernel's own static LDS (serial_lds_bytes sets the attribute to exactly this)
        if ((rc = serial_lds_bytes(ctx, (const void*)k_prepare_bids, &hog))) return rc;
        {
            ScopedEvent ev(ctx, TAG_WITNESS, s);
            hipLaunchKernelGGL(k_prepare_bids, dim3((B + 63) / 64), dim3(64), hog, s, B, N, (const u8*)bids_dev, (const u8*)lists_dev,
                               (const u64*)toggles_dev, ctx->mimc_c, (u32*)prove_in_dev, (u32*)verify_tail_dev);
            BBP_HIP_TRY(ctx, hipGetLastError());
        }
        // the prover's opening stage does not wait for the caller's stream (bbp.h); it does wait for this event
        BBP_HIP_TRY(ctx, hipEventRecord(ctx->ev_prep, s));
        ctx->ev_prep_valid = true;
        return BBP_OK;
    });
}

// ---- host-pointer batch calls: three staging slots, lock held only while enqueueing (context.h IoSlot) ---------------------------
namespace bbp {
struct SlotLease {
    bbp_ctx* ctx;
    bbp_ctx::IoSlot* sl;
    explicit SlotLease(bbp_ctx* c, bool verify = false) : ctx(c) {
        std::unique_lock<std::mutex> lk(c->io_mu);
        // strict rotation keeps consecutive calls on different slots; prove and verify calls have their own
        const uint32_t want = verify ? bbp_ctx::IO_SLOTS + c->io_vnext++ % bbp_ctx::IO_VSLOTS : c->io_next++ % bbp_ctx::IO_SLOTS;
        c->io_cv.wait(lk, [&] { return !c->io[want].busy; });
        sl = &c->io[want];
        sl->busy = true;
    }
    ~SlotLease() {
        {
            std::lock_guard<std::mutex> lk(ctx->io_mu);
            sl->busy = false;
        }
        ctx->io_cv.notify_all();
    }
};
// Waiting for a slot's event with the context lock released.  hipEventSynchronize by default; BBP_WAIT_POLL_US=n polls
// hipEventQuery every n microseconds instead (no measurable difference on MI355X / ROCm 7.2 once the copies were kept out of
// the DMA queues' way -- see fetch_results -- 18.7 k vs 18.6 k proofs/s from two threads; kept as a knob for runtimes where a
// thread parked in the runtime gets in the way of other threads' launches).
static hipError_t wait_event_polling(hipEvent_t ev) {
    static const int poll_us = [] {
        const char* e = getenv("BBP_WAIT_POLL_US");
        return e ? atoi(e) : 0;
    }();
    if (poll_us <= 0) return hipEventSynchronize(ev);
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
        usleep((useconds_t)poll_us);
    }
}

static int32_t pinned_reserve(bbp_ctx* ctx, void*& p, size_t& cap, size_t bytes) {
    if (cap >= bytes) return BBP_OK;
    if (p) BBP_HIP_TRY(ctx, hipHostFree(p));
    p = nullptr;
    cap = 0;
    const size_t want = bytes + (bytes >> 3) + 4096;
    BBP_HIP_TRY(ctx, hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return BBP_OK;
}
// inputs of a host-pointer call: caller's (pageable) memory -> the slot's pinned mirror -> device, on the context's copy stream.
// The main stream may still be busy with the previous call's MSM stage, and the opening stage of THIS call is meant to run under
// it; it only needs complete inputs (include/bbp.h), hence the wait -- on the copy's own event, polled.
static int32_t upload_inputs(bbp_ctx* ctx, bbp_ctx::IoSlot& sl, const uint8_t* a, size_t na, const uint8_t* b, size_t nb) {
    int32_t rc;
    if ((rc = dev_reserve(ctx, sl.in, na)) || (rc = dev_reserve(ctx, sl.ent, nb)) || (rc = pinned_reserve(ctx, sl.h_in, sl.h_in_cap, na + nb))) return rc;
    memcpy(sl.h_in, a, na);
    memcpy((uint8_t*)sl.h_in + na, b, nb);
    BBP_HIP_TRY(ctx, hipMemcpyAsync(sl.in.p, sl.h_in, na, hipMemcpyHostToDevice, ctx->copy));
    BBP_HIP_TRY(ctx, hipMemcpyAsync(sl.ent.p, (uint8_t*)sl.h_in + na, nb, hipMemcpyHostToDevice, ctx->copy));
    BBP_HIP_TRY(ctx, hipEventRecord(sl.ev_in, ctx->copy));
    BBP_HIP_TRY(ctx, wait_event_polling(sl.ev_in));
    return BBP_OK;
}
// Second half of a host-pointer call, context lock NOT held: wait for the slot's compute to finish, THEN copy the results down.
// The copy is issued only now, on the copy stream, on purpose: a D2H copy enqueued behind the kernels would sit at the head of a
// DMA engine's in-order queue waiting for them, and the NEXT call's input copy -- which that call's opening stage is waiting for
// -- would queue up behind it (measured: the input copy "took" 87 ms, i.e. the previous batch's whole MSM stage; two host
// threads got no overlap at all).  Issued after the wait, either copy takes ~0.2 ms whatever the compute queues are doing.
static int32_t fetch_results(bbp_ctx* ctx, bbp_ctx::IoSlot& sl, size_t bytes) {
    hipError_t e = hipSetDevice(ctx->device);  // the lock is not held here and nothing has set this thread's device yet (pool workers)
    if (e == hipSuccess) e = wait_event_polling(sl.ev);
    if (e == hipSuccess) e = hipMemcpyAsync(sl.h_out, sl.out.p, bytes, hipMemcpyDeviceToHost, ctx->copy);
    // ... and the context's health word with them: a call whose kernels (or any earlier call's) had to clamp a table gather reports
    // BBP_ERR_DEVICE for the whole call instead of handing out results computed from corrupted scratch with status 0
    if (e == hipSuccess) e = hipMemcpyAsync(sl.h_flag, ctx->health, sizeof(u32), hipMemcpyDeviceToHost, ctx->copy);
    if (e == hipSuccess) e = hipEventRecord(sl.ev_in, ctx->copy);
    if (e == hipSuccess) e = wait_event_polling(sl.ev_in);
    if (e != hipSuccess) {
        api_guard(ctx, [&]() -> int32_t { return ctx->err = std::string("collecting results: ") + hipGetErrorString(e), BBP_ERR_DEVICE; });
        return BBP_ERR_DEVICE;
    }
    if (*sl.h_flag) {
        const u32 flags = *sl.h_flag;
        api_guard(ctx, [&]() -> int32_t {
            char buf[200];
            snprintf(buf, sizeof buf, "engine health flags %#x: an MSM table gather was out of range and had to be clamped (corrupted engine scratch); "
                                      "results since then are not trustworthy -- free this context and create a new one", flags);
            return ctx->err = buf, BBP_ERR_DEVICE;
        });
        return BBP_ERR_DEVICE;
    }
    return BBP_OK;
}
}  // namespace bbp

// body of bbp_prove_batch.  Takes the context lock itself, for the enqueue phase only.
static int32_t prove_batch_host(bbp_ctx* ctx, uint32_t B, uint32_t N, const uint8_t* in, const uint8_t* entropy, uint8_t* out,
                                int32_t* status) {
    int32_t rc = api_guard(ctx, [&]() -> int32_t { return check_n(ctx, N); });
    if (rc) return rc;
    if (B == 0) return BBP_OK;
    fault_injected("prove_batch");
    const size_t in_stride = 7 * 32 + (size_t)N * 32 + 8, ent_stride = bbp_entropy_size(N), out_stride = bbp_proof_record_size(N);
    // host-side argument screening (the reference's typed API cannot express these states: SURVEY.md 8b)
    std::vector<uint8_t> fixed;
    for (uint32_t i = 0; i < B; i++) {
        const uint8_t* r = in + in_stride * i;
        status[i] = BBP_OK;
        uint64_t toggle;
        memcpy(&toggle, r + 7 * 32 + (size_t)N * 32, 8);
        if (toggle >= N) status[i] = BBP_ERR_BAD_ARG;
        for (int k = 0; k < 7 && status[i] == BBP_OK; k++) {
            u32 w[8];
            memcpy(w, r + 32 * k, 32);
            if (!sc_is_canonical(w)) status[i] = BBP_ERR_FORMAT;  // serde Scalar deserialisation is canonical-only
        }
        if (status[i] != BBP_OK) {
            if (fixed.empty()) fixed.assign(in, in + in_stride * B);
            memset(&fixed[in_stride * i], 0, in_stride);  // neutral stand-in so the batch geometry is unchanged
        }
    }
    const uint8_t* src = fixed.empty() ? in : fixed.data();
    std::vector<uint8_t> ent_host;
    if (!entropy) {
        // thread_rng replacement: 64 OS bytes per blinding, wide-reduced like Scalar::random; 32 OS bytes for the rng seed
        const uint32_t m = 4 + N;
        std::vector<uint8_t> raw((size_t)B * (64 * m + 32));
        if (!os_random(raw.data(), raw.size())) {
            api_guard(ctx, [&]() -> int32_t { return ctx->err = "cannot read /dev/urandom", BBP_ERR_DEVICE; });
            return BBP_ERR_DEVICE;
        }
        ent_host.resize(ent_stride * B);
        for (uint32_t i = 0; i < B; i++) {
            const uint8_t* rp = &raw[(size_t)i * (64 * m + 32)];
            for (uint32_t k = 0; k < m; k++) {
                u32 w[16];
                memcpy(w, rp + 64 * k, 64);
                sc s = sc_from_wide(w);
                sc_tobytes(&ent_host[ent_stride * i + 32 * k], s);
            }
            memcpy(&ent_host[ent_stride * i + 32 * m], rp + 64 * m, 32);
        }
        entropy = ent_host.data();
    }
    static const bool trace = getenv("BBP_TRACE") != nullptr;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_enter = now_ms();
    SlotLease lease(ctx);  // may wait for the call two back to collect its results; the context lock is NOT held here
    bbp_ctx::IoSlot& sl = *lease.sl;
    const double t_slot = now_ms();
    double t_lock = 0, t_h2d = 0;
    rc = api_guard(ctx, [&]() -> int32_t {
        int32_t rc;
        t_lock = now_ms();
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        if ((rc = dev_reserve(ctx, sl.out, out_stride * B)) || (rc = pinned_reserve(ctx, sl.h_out, sl.h_cap, out_stride * B)) ||
            (rc = upload_inputs(ctx, sl, src, in_stride * B, entropy, ent_stride * B)))
            return rc;
        t_h2d = now_ms();
        // Very large host batches go through the engine in equal chunks of at most host_chunk_prove proofs so that scratch stays
        // bounded (~1.3 MB per proof of the largest call, three buffers); consecutive calls pipeline -- chunk k+1's opening stage
        // under chunk k's MSMs.  One 16384-proof call was measured 5 % faster than four of 4096, hence the large default.
        const uint32_t n_chunks = (B + host_chunk_prove() - 1) / host_chunk_prove(), chunk = (B + n_chunks - 1) / n_chunks;
        for (uint32_t first = 0; first < B; first += chunk) {
            const uint32_t nb = B - first < chunk ? B - first : chunk;
            if ((rc = prove_batch_dev(ctx, nb, N, (const u8*)sl.in.p + in_stride * first, (const u8*)sl.ent.p + ent_stride * first,
                                      (u8*)sl.out.p + out_stride * first, ctx->stream)))
                return rc;
        }
        BBP_HIP_TRY(ctx, hipEventRecord(sl.ev, ctx->stream));
        return BBP_OK;
    });
    if (rc) return rc;
    const double t_enq = now_ms();
    if ((rc = fetch_results(ctx, sl, out_stride * B))) return rc;  // lock released: another thread may be enqueueing the next batch now
    memcpy(out, sl.h_out, out_stride * B);
    if (trace)
        fprintf(stderr, "[bbp trace] prove_batch B=%u slot=%d: enter %.1f, slot +%.1f, lock +%.1f, h2d +%.1f, enqueued +%.1f, done +%.1f\n", B,
                (int)(&sl - ctx->io), t_enter, t_slot - t_enter, t_lock - t_enter, t_h2d - t_enter, t_enq - t_enter, now_ms() - t_enter);
    for (uint32_t i = 0; i < B; i++)
        if (status[i] != BBP_OK) memset(out + out_stride * i, 0, out_stride);
    return BBP_OK;
}

template <class F>
static int32_t no_throw_ctx(bbp_ctx* ctx, F&& body) noexcept {  // for bodies that lock in phases: same mapping as api_guard
    try {
        return body();
    } catch (const std::invalid_argument& e) {
        api_guard(ctx, [&]() -> int32_t { return ctx->err = std::string("invalid argument: ") + e.what(), BBP_ERR_BAD_ARG; });
        return BBP_ERR_BAD_ARG;
    } catch (const std::bad_alloc&) {
        api_guard(ctx, [&]() -> int32_t { return ctx->err = "host allocation failed", BBP_ERR_INTERNAL; });
        return BBP_ERR_INTERNAL;
    } catch (const std::exception& e) {
        api_guard(ctx, [&]() -> int32_t { return ctx->err = std::string("internal error: ") + e.what(), BBP_ERR_INTERNAL; });
        return BBP_ERR_INTERNAL;
    } catch (...) {
        return BBP_ERR_INTERNAL;
    }
}

extern "C" int32_t bbp_prove_batch(bbp_ctx* ctx, uint32_t B, uint32_t N, const uint8_t* in, const uint8_t* entropy, uint8_t* out,
                                   int32_t* status) {
    if (!ctx || !in || !out || !status) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) {  // block split over the members, records in request order (pool.cpp)
        const int32_t rc = api_guard(ctx, [&]() -> int32_t { return check_n(ctx, N); });
        if (rc || B == 0) return rc;
        return no_throw_ctx(ctx, [&]() -> int32_t { return pool_prove_batch(ctx, B, N, in, entropy, out, status); });
    }
    return no_throw_ctx(ctx, [&]() -> int32_t { return prove_batch_host(ctx, B, N, in, entropy, out, status); });
}

extern "C" int32_t bbp_prove_batch_dev(bbp_ctx* ctx, uint32_t B, uint32_t N, const void* in_dev, const void* entropy_dev, void* out_dev,
                                       void* stream) {
    if (!ctx || !in_dev || !entropy_dev || !out_dev) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) return pool_reject(ctx, "bbp_prove_batch_dev");
    return api_guard(ctx, [&]() -> int32_t {
        int32_t rc = check_n(ctx, N);
        if (rc) return rc;
        if (B == 0) return BBP_OK;
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        return prove_batch_dev(ctx, B, N, (const u8*)in_dev, (const u8*)entropy_dev, (u8*)out_dev, pick_stream(ctx, stream));
    });
}

// what the call combiner runs for a group of concurrent bbp_prove callers (submit.cpp)
int32_t bbp::prove_batch_locked(bbp_ctx* ctx, uint32_t B, uint32_t N, const uint8_t* in, const uint8_t* entropy, uint8_t* out,
                                int32_t* status, std::string* err) {
    const int32_t rc = no_throw_ctx(ctx, [&]() -> int32_t { return prove_batch_host(ctx, B, N, in, entropy, out, status); });
    if (rc && err) *err = tls_error();
    return rc;
}

// completion hook of the asynchronous entry points (runs on a combiner thread): the request's message becomes that thread's
// bbp_last_error text for the duration of the callback, then the request is gone
static void async_done(Request* r) {
    try {
        set_tls_error(r->origin, r->status == BBP_OK ? std::string() : !r->err.empty() ? r->err : r->status == BBP_ERR_BAD_ARG ? "toggle >= N"
                                     : r->status == BBP_ERR_FORMAT && r->kind == 0 ? "non-canonical scalar input"
                                                                                    : status_text(r->status));
    } catch (...) {
    }
    void (*fn)(void*, int32_t) = r->user_fn;
    void* user = r->user;
    const int32_t st = r->status;
    delete r;
    fn(user, st);
}

static void fill_prove_input(std::vector<uint8_t>& in, const uint8_t scalars7[7 * 32], const uint8_t* pub_list, uint32_t N, uint64_t toggle) {
    in.resize(7 * 32 + (size_t)N * 32 + 8);
    memcpy(&in[0], scalars7, 7 * 32);
    memcpy(&in[7 * 32], pub_list, (size_t)N * 32);
    memcpy(&in[7 * 32 + (size_t)N * 32], &toggle, 8);
}

extern "C" int32_t bbp_prove(bbp_ctx* ctx, const uint8_t scalars7[7 * 32], const uint8_t* pub_list, uint32_t N, uint64_t toggle,
                             const uint8_t* entropy, uint8_t* proof_out, uint32_t* proof_len) {
    if (!ctx || !scalars7 || !proof_out) return BBP_ERR_BAD_ARG;
    int32_t rc = api_guard(ctx, [&]() -> int32_t { return check_n(ctx, N); });
    if (rc) return rc;
    if (!pub_list) return BBP_ERR_BAD_ARG;
    return no_throw([&]() -> int32_t {
        std::vector<uint8_t> in;
        fill_prove_input(in, scalars7, pub_list, N, toggle);
        Request r;
        r.kind = 0;
        r.N = N;
        r.in = in.data();
        r.in_len = in.size();
        r.entropy = entropy;
        r.out = proof_out;
        const int32_t st = static_cast<Combiner*>(ctx->combiner)->submit(ctx, r);
        if (st != BBP_OK) {
            set_tls_error(ctx, !r.err.empty() ? r.err : st == BBP_ERR_BAD_ARG ? "toggle >= N" : "non-canonical scalar input");
            return st;
        }
        if (proof_len) *proof_len = BBP_R1CS_PROOF_BYTES;
        return BBP_OK;
    }, ctx);
}

extern "C" int32_t bbp_prove_async(bbp_ctx* ctx, const uint8_t scalars7[7 * 32], const uint8_t* pub_list, uint32_t N, uint64_t toggle,
                                   const uint8_t* entropy, uint8_t* proof_out, bbp_done_fn done, void* user) {
    if (!ctx || !scalars7 || !proof_out || !done) return BBP_ERR_BAD_ARG;
    int32_t rc = api_guard(ctx, [&]() -> int32_t { return check_n(ctx, N); });
    if (rc) return rc;
    if (!pub_list) return BBP_ERR_BAD_ARG;
    return no_throw([&]() -> int32_t {
        Request* r = new Request();
        r->kind = 0;
        r->N = N;
        fill_prove_input(r->own_in, scalars7, pub_list, N, toggle);  // inputs are copied: the caller's buffers are free on return
        r->in = r->own_in.data();
        r->in_len = r->own_in.size();
        if (entropy) {
            r->own_entropy.assign(entropy, entropy + bbp_entropy_size(N));
            r->entropy = r->own_entropy.data();
        }
        r->out = proof_out;
        r->on_done = async_done;
        r->origin = ctx;
        r->user_fn = done;
        r->user = user;
        if (!static_cast<Combiner*>(ctx->combiner)->submit_async(ctx, r)) {
            delete r;
            set_tls_error(ctx, "call combiner: cannot start a batch thread");
            return BBP_ERR_INTERNAL;
        }
        return BBP_OK;
    }, ctx);
}

// rec_ver 0: compact 1121-byte proofs; 1: the 2-phase 1217-byte R1CSProof layout (both parse in the reference)
// Takes the context lock itself, for the enqueue phase only (aggregated mode synchronises inside it: the host reads group verdicts).
static int32_t verify_batch_host(bbp_ctx* ctx, uint32_t B, uint32_t N, uint32_t rec_ver, const uint8_t* in, int32_t* status,
                                 uint32_t group = 0, uint32_t* n_fallback = nullptr) {
    const size_t stride = (size_t)(rec_ver ? 1217u : 1121u) + 32 * (4 + (size_t)N) + 96 + (size_t)N * 32;
    if (group == 0 && rec_ver == 0 && ctx->verify_group > 1 && B >= 2 * ctx->verify_group) group = ctx->verify_group;  // BBP_VERIFY_AGGREGATE
    std::vector<uint8_t> ent((size_t)B * 32);
    if (!os_random(ent.data(), ent.size())) {  // Verifier::verify mixes thread_rng into its TranscriptRng (A.7)
        api_guard(ctx, [&]() -> int32_t { return ctx->err = "cannot read /dev/urandom", BBP_ERR_DEVICE; });
        return BBP_ERR_DEVICE;
    }
    SlotLease lease(ctx, true);
    bbp_ctx::IoSlot& sl = *lease.sl;
    bbp_ctx::VLane& L = ctx->vl[(&sl - ctx->io) - bbp_ctx::IO_SLOTS];  // verifier lane = staging slot: consecutive host calls overlap on the device
    int32_t rc = api_guard(ctx, [&]() -> int32_t {
        int32_t rc;
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        if ((rc = dev_reserve(ctx, sl.out, 4 * ((size_t)B + 1))) || (rc = pinned_reserve(ctx, sl.h_out, sl.h_cap, 4 * ((size_t)B + 1))) ||
            (rc = upload_inputs(ctx, sl, in, stride * B, ent.data(), ent.size())))
            return rc;
        const uint32_t n_chunks = (B + host_chunk_verify() - 1) / host_chunk_verify(), chunk = (B + n_chunks - 1) / n_chunks;
        for (uint32_t first = 0; first < B; first += chunk) {  // bounded scratch for any B (see bbp_prove_batch)
            const uint32_t nb = B - first < chunk ? B - first : chunk;
            const u8 *cin = (const u8*)sl.in.p + stride * first, *cent = (const u8*)sl.ent.p + 32 * (size_t)first;
            int32_t* cst = (int32_t*)sl.out.p + first;
            if (group > 1) {  // stream-ordered: no synchronisation while the context lock is held
                if (first == 0 && L.agg_count) BBP_HIP_TRY(ctx, hipMemsetAsync(L.agg_count + 1, 0, sizeof(u32), L.stream));
                if ((rc = verify_batch_agg_dev(ctx, nb, N, group, cin, cent, cst, L.stream, nullptr, (u32*)sl.out.p + B))) return rc;
            } else if ((rc = verify_batch_dev_ex(ctx, nb, N, rec_ver, 0, cin, cent, cst, L.stream)))
                return rc;
        }
        BBP_HIP_TRY(ctx, hipEventRecord(sl.ev, L.stream));
        return BBP_OK;
    });
    if (rc) return rc;
    if ((rc = fetch_results(ctx, sl, 4 * ((size_t)B + (group > 1 ? 1 : 0))))) return rc;
    memcpy(status, sl.h_out, 4 * (size_t)B);
    if (group > 1 && n_fallback) memcpy(n_fallback, (const uint8_t*)sl.h_out + 4 * (size_t)B, 4);  // running total of this call's chunks
    return BBP_OK;
}

int32_t bbp::verify_batch_locked(bbp_ctx* ctx, uint32_t B, uint32_t N, uint32_t rec_ver, const uint8_t* in, int32_t* status,
                                 std::string* err) {
    const int32_t rc = no_throw_ctx(ctx, [&]() -> int32_t { return verify_batch_host(ctx, B, N, rec_ver, in, status); });
    if (rc && err) *err = tls_error();
    return rc;
}

extern "C" int32_t bbp_verify_batch(bbp_ctx* ctx, uint32_t B, uint32_t N, const uint8_t* in, int32_t* status) {
    if (!ctx || !in || !status) return BBP_ERR_BAD_ARG;
    int32_t rc = api_guard(ctx, [&]() -> int32_t { return check_n(ctx, N); });
    if (rc || B == 0) return rc;
    if (is_pool(ctx)) return no_throw_ctx(ctx, [&]() -> int32_t { return pool_verify_batch(ctx, B, N, in, status, false, 0, nullptr); });
    return no_throw_ctx(ctx, [&]() -> int32_t { return verify_batch_host(ctx, B, N, 0, in, status); });
}

extern "C" int32_t bbp_verify_batch_dev(bbp_ctx* ctx, uint32_t B, uint32_t N, const void* in_dev, const void* entropy_dev,
                                        void* status_dev, void* stream) {
    if (!ctx || !in_dev || !entropy_dev || !status_dev) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) return pool_reject(ctx, "bbp_verify_batch_dev");
    return api_guard(ctx, [&]() -> int32_t {
        int32_t rc = check_n(ctx, N);
        if (rc) return rc;
        if (B == 0) return BBP_OK;
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        return verify_batch_dev(ctx, B, N, (const u8*)in_dev, (const u8*)entropy_dev, (int32_t*)status_dev, pick_stream(ctx, stream));
    });
}

extern "C" int32_t bbp_verify_batch_aggregated(bbp_ctx* ctx, uint32_t B, uint32_t N, const uint8_t* in, int32_t* status, uint32_t group,
                                               uint32_t* n_fallback) {
    if (n_fallback) *n_fallback = 0;
    if (!ctx || !in || !status) return BBP_ERR_BAD_ARG;
    int32_t rc = api_guard(ctx, [&]() -> int32_t { return check_n(ctx, N); });
    if (rc || B == 0) return rc;
    if (is_pool(ctx))
        return no_throw_ctx(ctx, [&]() -> int32_t { return pool_verify_batch(ctx, B, N, in, status, true, group ? group : BBP_AGG_GROUP_DEFAULT, n_fallback); });
    return no_throw_ctx(ctx, [&]() -> int32_t { return verify_batch_host(ctx, B, N, 0, in, status, group ? group : BBP_AGG_GROUP_DEFAULT, n_fallback); });
}

extern "C" int32_t bbp_verify_batch_aggregated_dev(bbp_ctx* ctx, uint32_t B, uint32_t N, const void* in_dev, const void* entropy_dev,
                                                   void* status_dev, uint32_t group, uint32_t* n_fallback, void* stream) {
    if (n_fallback) *n_fallback = 0;
    if (!ctx || !in_dev || !entropy_dev || !status_dev) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) return pool_reject(ctx, "bbp_verify_batch_aggregated_dev");
    return api_guard(ctx, [&]() -> int32_t {
        int32_t rc = check_n(ctx, N);
        if (rc) return rc;
        if (B == 0) return BBP_OK;
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        return verify_batch_agg_dev(ctx, B, N, group ? group : BBP_AGG_GROUP_DEFAULT, (const u8*)in_dev, (const u8*)entropy_dev,
                                    (int32_t*)status_dev, pick_stream(ctx, stream), n_fallback);
    });
}

// Structural parse exactly as R1CSProof::from_bytes / InnerProductProof::from_bytes order their checks: a record of
// any length is either a FormatError, or well formed with a wrong IPA depth (-> VerificationError from verification_scalars,
// n != 2^lg_n), or one of the two layouts the device kernels take (*ver).  BBP_OK = hand it to the device.
static int32_t screen_verify_args(const uint8_t* record, uint32_t record_len, const uint8_t score[32], const uint8_t z_img[32],
                                  const uint8_t seed[32], uint32_t N, uint8_t* ver_out) {
    const uint32_t tail = 32u * (4u + N);
    if (record_len < tail + 1u) return BBP_ERR_FORMAT;
    const uint32_t plen = record_len - tail;
    const uint8_t ver = record[0];
    if (ver != 0 && ver != 1) return BBP_ERR_FORMAT;
    if ((plen - 1u) % 32u != 0) return BBP_ERR_FORMAT;
    const uint32_t nel = (plen - 1u) / 32u, npts = ver == 0 ? 3u : 6u;
    if (nel < npts + 5u + 3u + 2u) return BBP_ERR_FORMAT;
    auto canonical = [&](uint32_t el) {
        u32 w[8];
        memcpy(w, record + 1 + 32 * (size_t)el, 32);
        return sc_is_canonical(w);
    };
    for (uint32_t k = 0; k < 3; k++)
        if (!canonical(npts + 5u + k)) return BBP_ERR_FORMAT;  // t_x, t_x_blinding, e_blinding
    const uint32_t ipp_el = nel - npts - 8u;
    if (ipp_el < 2u || (ipp_el - 2u) % 2u != 0) return BBP_ERR_FORMAT;
    const uint32_t lg_n = (ipp_el - 2u) / 2u;
    if (lg_n >= 32u) return BBP_ERR_FORMAT;
    if (!canonical(nel - 2u) || !canonical(nel - 1u)) return BBP_ERR_FORMAT;  // a, b
    // score, z_img, seed reach Verify::new as serde-deserialised Scalars (verify.rs:100-104): canonical encodings only
    for (const uint8_t* pub : {score, z_img, seed}) {
        u32 w[8];
        memcpy(w, pub, 32);
        if (!sc_is_canonical(w)) return BBP_ERR_FORMAT;
    }
    if (lg_n != 11u) return BBP_ERR_VERIFY;  // padded_n = 2048 != 2^lg_n
    *ver_out = ver;
    return BBP_OK;
}

static void fill_verify_input(std::vector<uint8_t>& in, const uint8_t* record, uint32_t record_len, const uint8_t score[32],
                              const uint8_t z_img[32], const uint8_t seed[32], const uint8_t* pub_list, uint32_t N) {
    in.resize((size_t)record_len + 96 + (size_t)N * 32);
    memcpy(&in[0], record, record_len);
    memcpy(&in[record_len], score, 32);
    memcpy(&in[record_len + 32], z_img, 32);
    memcpy(&in[record_len + 64], seed, 32);
    memcpy(&in[record_len + 96], pub_list, (size_t)N * 32);
}

extern "C" int32_t bbp_verify(bbp_ctx* ctx, const uint8_t* record, uint32_t record_len, const uint8_t score[32], const uint8_t z_img[32],
                              const uint8_t seed[32], const uint8_t* pub_list, uint32_t N) {
    if (!ctx || !record || !score || !z_img || !seed) return BBP_ERR_BAD_ARG;
    int32_t rc = api_guard(ctx, [&]() -> int32_t { return check_n(ctx, N); });
    if (rc) return rc;
    if (!pub_list) return BBP_ERR_BAD_ARG;
    uint8_t ver = 0;
    if ((rc = screen_verify_args(record, record_len, score, z_img, seed, N, &ver))) {
        set_tls_error(ctx, rc == BBP_ERR_FORMAT ? "malformed proof record or non-canonical public scalar" : "inner-product proof of the wrong depth");
        return rc;
    }
    return no_throw([&]() -> int32_t {
        std::vector<uint8_t> in;
        fill_verify_input(in, record, record_len, score, z_img, seed, pub_list, N);
        Request r;
        r.kind = 1;
        r.N = N;
        r.rec_ver = ver;
        r.in = in.data();
        r.in_len = in.size();
        const int32_t st = static_cast<Combiner*>(ctx->combiner)->submit(ctx, r);
        if (st != BBP_OK) set_tls_error(ctx, !r.err.empty() ? r.err : st == BBP_ERR_VERIFY ? "proof rejected" : "malformed proof");
        return st;
    }, ctx);
}

extern "C" int32_t bbp_verify_async(bbp_ctx* ctx, const uint8_t* record, uint32_t record_len, const uint8_t score[32],
                                    const uint8_t z_img[32], const uint8_t seed[32], const uint8_t* pub_list, uint32_t N, bbp_done_fn done,
                                    void* user) {
    if (!ctx || !record || !score || !z_img || !seed || !done) return BBP_ERR_BAD_ARG;
    int32_t rc = api_guard(ctx, [&]() -> int32_t { return check_n(ctx, N); });
    if (rc) return rc;
    if (!pub_list) return BBP_ERR_BAD_ARG;
    uint8_t ver = 0;
    if ((rc = screen_verify_args(record, record_len, score, z_img, seed, N, &ver))) return rc;  // decided on the host: no callback
    return no_throw([&]() -> int32_t {
        Request* r = new Request();
        r->kind = 1;
        r->N = N;
        r->rec_ver = ver;
        fill_verify_input(r->own_in, record, record_len, score, z_img, seed, pub_list, N);
        r->in = r->own_in.data();
        r->in_len = r->own_in.size();
        r->on_done = async_done;
        r->origin = ctx;
        r->user_fn = done;
        r->user = user;
        if (!static_cast<Combiner*>(ctx->combiner)->submit_async(ctx, r)) {
            delete r;
            set_tls_error(ctx, "call combiner: cannot start a batch thread");
            return BBP_ERR_INTERNAL;
        }
        return BBP_OK;
    }, ctx);
}

// Grow every per-batch buffer of the context to what batches of `max_batch` need, by running the real paths once per staging
// slot / buffer parity on all-zero dummy rows (a zero row is a well-formed request: canonical scalars, toggle 0): whatever a
// later batch of at most that size touches -- batch buffers in rotation, MSM scratch per slice, draw buffers, staging slots and
// their pinned mirrors, the compiled circuit of list length N -- exists afterwards, so no call pays for (and no neighbour stalls
// behind) a hipFree + hipMalloc of gigabytes: growing scratch under load was measured through the UDS server as one ~0.9 s stall
// per run, i.e. the whole p99.
extern "C" int32_t bbp_reserve(bbp_ctx* ctx, uint32_t max_batch, uint32_t N) {
    if (!ctx) return BBP_ERR_BAD_ARG;
    int32_t rc = api_guard(ctx, [&]() -> int32_t { return check_n(ctx, N); });
    if (rc || max_batch == 0) return rc;
    if (is_pool(ctx)) {  // every member, all at once
        std::vector<int32_t> rcs(ctx->members.size(), BBP_OK);
        std::vector<std::thread> th;
        for (size_t i = 0; i < ctx->members.size(); i++) {
            try {
                th.emplace_back([&, i] { rcs[i] = bbp_reserve(ctx->members[i], max_batch, N); });
            } catch (...) {
                rcs[i] = bbp_reserve(ctx->members[i], max_batch, N);
            }
        }
        for (auto& t : th) t.join();
        for (int32_t r : rcs)
            if (r) return r;
        return BBP_OK;
    }
    return no_throw_ctx(ctx, [&]() -> int32_t {
        const uint32_t B = max_batch;
        const size_t in_stride = 7 * 32 + (size_t)N * 32 + 8, rec = bbp_proof_record_size(N), v_stride = rec + 96 + (size_t)N * 32;
        std::vector<uint8_t> in(in_stride * B, 0), ent((size_t)bbp_entropy_size(N) * B, 0), out(rec * B), vin(v_stride * B, 0);
        std::vector<int32_t> st(B);
        int32_t rc = BBP_OK;
        // batches below 1024 proofs rotate three buffers and two opening streams, larger ones two buffers: both shapes, every slot
        for (int k = 0; k < bbp_ctx::IO_SLOTS && rc == BBP_OK; k++) rc = prove_batch_host(ctx, B, N, in.data(), ent.data(), out.data(), st.data());
        // the small-batch path rotates over ALL five batch buffers (buffer = call % PROVE_BUFS): five more calls whatever B is -- with
        // max_batch < 1024 the loop above IS that path but touches only three of them, and the first 4th / 5th call under load would
        // grow its buffer with hipFree + hipMalloc: the stall this function exists to prevent
        const uint32_t small = B < 1023 ? B : 1023;
        for (int k = 0; k < bbp_ctx::PROVE_BUFS && rc == BBP_OK; k++) rc = prove_batch_host(ctx, small, N, in.data(), ent.data(), out.data(), st.data());
        // ... and the shape a caller with several calls in flight gets (prover.hip "deep"): whole calls of B in rotation, five buffers
        if (B >= 1024 && ctx->slices > 1 && ctx->rotate_deep_max > 0 && B <= (uint32_t)ctx->rotate_deep_max) {
            struct Reset { bool& b; ~Reset() { b = false; } } reset{ctx->force_deep};
            ctx->force_deep = true;
            for (int k = 0; k < bbp_ctx::PROVE_BUFS && rc == BBP_OK; k++) rc = prove_batch_host(ctx, B, N, in.data(), ent.data(), out.data(), st.data());
        }
        for (int k = 0; k < bbp_ctx::IO_VSLOTS && rc == BBP_OK; k++) rc = verify_batch_host(ctx, B, N, 0, vin.data(), st.data());
        return rc;
    });
}

extern "C" int32_t bbp_set_batching(bbp_ctx* ctx, uint32_t window_us, uint32_t max_batch) {
    if (!ctx || !ctx->combiner) return BBP_ERR_BAD_ARG;
    return no_throw([&]() -> int32_t {
        static_cast<Combiner*>(ctx->combiner)->configure(window_us, max_batch);
        return BBP_OK;
    });
}

extern "C" int32_t bbp_batching_stats(bbp_ctx* ctx, uint64_t* n_calls, uint64_t* n_requests, uint32_t* max_seen) {
    if (!ctx || !ctx->combiner) return BBP_ERR_BAD_ARG;
    return no_throw([&]() -> int32_t {
        if (ctx->owner && ctx->owner->combiner) {  // a pool member: the batches its pool's combiner has dealt to it
            static_cast<Combiner*>(ctx->owner->combiner)->target_stats(ctx->member_index, n_calls, n_requests);
            if (max_seen) *max_seen = 0;
            return BBP_OK;
        }
        static_cast<Combiner*>(ctx->combiner)->stats(n_calls, n_requests, max_seen);
        return BBP_OK;
    });
}

extern "C" int32_t bbp_debug_challenges(bbp_ctx* ctx, uint32_t B, uint32_t N, uint32_t proof, uint8_t* out32x32) {
    if (!ctx || !out32x32 || proof >= B) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) return pool_reject(ctx, "bbp_debug_challenges");
    return api_guard(ctx, [&]() -> int32_t {
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        BBP_HIP_TRY(ctx, hipDeviceSynchronize());
        return debug_read_misc(ctx, B, N, proof, out32x32);
    });
}

extern "C" int32_t bbp_set_profiling(bbp_ctx* ctx, int32_t on) {
    if (!ctx) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) return pool_reject(ctx, "bbp_set_profiling");
    return api_guard(ctx, [&]() -> int32_t {
        ctx->profile = on != 0;
        return BBP_OK;
    });
}

// Drains the recorded events: out[2i] = kernel tag, out[2i+1] = microseconds. Synchronises the device.
extern "C" int32_t bbp_last_timings(bbp_ctx* ctx, float* out, uint32_t cap, uint32_t* n) {
    if (!ctx || !n) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) return pool_reject(ctx, "bbp_last_timings");
    return api_guard(ctx, [&]() -> int32_t {
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        BBP_HIP_TRY(ctx, hipDeviceSynchronize());
        uint32_t k = 0;
        for (auto& e : ctx->events) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess && out && 2 * k + 1 < cap) {
                out[2 * k] = (float)e.tag;
                out[2 * k + 1] = ms * 1000.f;
                k++;
            }
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        ctx->events.clear();
        *n = 2 * k;
        return BBP_OK;
    });
}
