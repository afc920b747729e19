// Context creation: derive the Pedersen / Bulletproof generators and MiMC constants ONCE and keep them
// resident in HBM, together with the window-shifted affine table the MSM kernels gather from.
//
// Replaces generate_cs_transcript()'s per-call `PedersenGens::default()` + `BulletproofGens::new(2048, 1)`
// (src/blindbid/mod.rs:34-40, re-run by every prove and verify: SURVEY.md F6) and lazy_static CONSTANTS
// (src/blindbid/mod.rs:7-24).  Hashing (SHA-512 chain, SHAKE256 stream, SHA3-512) is host work; every field /
// group operation (Elligator maps, doublings, normalisation) runs on the device.
#include <stdio.h>
#include <stdlib.h>

#include <dirent.h>
#include <unistd.h>

#include <atomic>
#include <mutex>

#include "context.h"
#include "batch.h"
#include "hosthash.h"
#include "submit.h"

namespace bbp {

// thread i: uniform[i] (16 LE words) -> gens[dst(i)]
__global__ void k_derive_generators(const u32* __restrict__ uniform, ge* __restrict__ gens) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > 2 * BBP_GENS_CAPACITY + 1) return;
    if (i == 2 * BBP_GENS_CAPACITY + 1) {
        gens[BBP_BASE_B] = ge_basepoint();
        return;
    }
    u32 w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = uniform[(size_t)i * 16 + k];
    gens[i] = ge_from_uniform_words(w);  // 0: B_blinding, 1..2048: G, 2049..4096: H
}

// PAD_BASE0 + N - 1 = H[418 + 3N] + ... + H[1023] (context.h): one lane walks the suffix sums down
__global__ void k_pad_sums(ge* __restrict__ gens) {
    if (blockIdx.x || threadIdx.x) return;
    ge acc = ge_identity();
    gens[PAD_BASE0 + PAD_BASES - 1] = acc;  // N = 202: no padded multiplier, empty range (never used with a non-zero scalar)
    for (int k = 1023; k >= 421; k--) {
        acc = ge_add(acc, gens[BBP_BASE_H0 + k]);
        if ((k - 418) % 3 == 0) gens[PAD_BASE0 + (k - 418) / 3 - 1] = acc;
    }
}

// merged bases (context.h): MRG1(i) = G[i] + H[i] + H[i+1] at MRG_BASE0 + i, MRG2(i) = G[i] + G[i+1] + H[i+1] at MRG_BASE0 + 2048 + i
__global__ void k_merge_sums(ge* __restrict__ gens) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2048u) return;
    const ge g = gens[BBP_BASE_G0 + i];
    if (i == 2047u) {  // no neighbour: never referenced, any valid point will do
        gens[MRG_BASE0 + i] = g;
        gens[MRG_BASE0 + 2048 + i] = g;
        return;
    }
    const ge h1 = gens[BBP_BASE_H0 + i + 1];
    gens[MRG_BASE0 + i] = ge_add(ge_add(g, gens[BBP_BASE_H0 + i]), h1);
    gens[MRG_BASE0 + 2048 + i] = ge_add(ge_add(g, gens[BBP_BASE_G0 + i + 1]), h1);
}

// thread (i, c): rows 16c .. 16c+15 of generator i: table[i*256 + b] = affine cached form of 2^b * gens[i]
constexpr int PT_CHUNK = 16;
__global__ void k_build_ptable(const ge* __restrict__ gens, niels_row* __restrict__ table) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= TAB_BASES * (MSM_POS / PT_CHUNK)) return;
    const u32 i = t / (MSM_POS / PT_CHUNK), c = t % (MSM_POS / PT_CHUNK);
    ge p = gens[i];
    for (u32 k = 0; k < c * PT_CHUNK; k++) p = ge_dbl(p);
    for (u32 b = c * PT_CHUNK; b < (c + 1) * PT_CHUNK; b++) {
        table[(size_t)i * MSM_POS + b] = niels_to_row(ge_to_niels(p, fe_invert(p.Z)));
        p = ge_dbl(p);
    }
}

// radix-16 comb for the two Pedersen bases: comb[b][j][m-1] = m * 16^j * Base_b, m = 1..8, j = 0..63
__global__ void k_build_comb(const ge* __restrict__ gens, niels_packed* __restrict__ comb) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * 64) return;
    u32 b = t / 64, j = t % 64;
    ge p = gens[b == 0 ? BBP_BASE_B : BBP_BASE_BBLIND];
    for (u32 k = 0; k < 4 * j; k++) p = ge_dbl(p);
    ge m = p;
    for (int k = 0; k < 8; k++) {
        comb[((size_t)b * 64 + j) * 8 + k] = niels_pack(ge_to_niels(m, fe_invert(m.Z)));
        m = ge_add(m, p);
    }
}

static void mimc_constants_host(std::vector<uint8_t>& out) {
    // src/blindbid/mod.rs:7-24: h = SHA512("blind bid"); c_i = wide_reduce(h); h = SHA512(c_i bytes)
    out.resize(BBP_MIMC_ROUNDS * 32);
    uint8_t h[64];
    const uint8_t seed[] = {'b', 'l', 'i', 'n', 'd', ' ', 'b', 'i', 'd'};
    sha512(seed, sizeof(seed), h);
    for (int i = 0; i < BBP_MIMC_ROUNDS; i++) {
        u32 w[16];
        memcpy(w, h, 64);
        sc c = sc_from_wide(w);  // integer reduction of a constant: setup plumbing, not the hot path
        sc_tobytes(&out[32 * i], c);
        sha512(&out[32 * i], 32, h);
    }
}

}  // namespace bbp

namespace bbp {
// ---- the library owns its hardware-queue configuration (round 4) ------------------------------------------------------------------
// A context keeps up to eleven streams (prover: caller's, two opening, three slices, copy; verifier: four lanes); HIP hands hardware
// queues out in stream-creation order and lets later streams SHARE once GPU_MAX_HW_QUEUES (default 4) are taken -- two busy streams on
// one queue serialise (prover up to 30 % slower at 4, a verifier lane on a shared queue costs 1024 verifications 7.2 instead of 5.2 ms).
// The variable is read ONCE, when the HIP runtime initialises in the process.  A host program bound per INTEGRATION.md (Rust, Go, C)
// would have to know; so bbp_init / bbp_init_all / bbp_pool_init look BEFORE their first HIP call: if the caller's environment has
// the variable it is left alone; else if HIP / HSA is not up yet in this process (no descriptor on /dev/kfd) the library exports
// BBP_HWQ_DEFAULT itself; else it is too late, and bbp_describe says so in a WARNING line.
#ifndef BBP_HWQ_DEFAULT
#define BBP_HWQ_DEFAULT "16"
#endif
static std::atomic<int> g_hwq_state{0};  // 1 caller's environment, 2 set by the library, 3 too late (HIP already initialised, variable unset)
static bool hsa_is_up() {
    DIR* d = opendir("/proc/self/fd");
    if (!d) return false;
    bool up = false;
    char path[64], target[256];
    while (dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        snprintf(path, sizeof path, "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(path, target, sizeof target - 1);
        if (n <= 0) continue;
        target[n] = 0;
        if (!strcmp(target, "/dev/kfd")) {
            up = true;
            break;
        }
    }
    closedir(d);
    return up;
}
void own_hw_queues() {
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("GPU_MAX_HW_QUEUES")) g_hwq_state = 1;
        else if (hsa_is_up()) g_hwq_state = 3;
        else {
            setenv("GPU_MAX_HW_QUEUES", BBP_HWQ_DEFAULT, 0);
            g_hwq_state = 2;
        }
    });
}
int hw_queues_state() { return g_hwq_state.load(); }
}  // namespace bbp

using namespace bbp;

static int32_t init_body(int32_t device, bbp_ctx** out);

extern "C" int32_t bbp_init(int32_t device, bbp_ctx** out) {
    if (!out) return BBP_ERR_BAD_ARG;
    *out = nullptr;
    own_hw_queues();  // before the first HIP call of this function (and, if possible, of the process)
    if (device == -1) return bbp_init_all(out);  // a pool over every visible GPU (pool.cpp)
    try {
        const int32_t rc = init_body(device, out);
        if (rc != BBP_OK && *out) set_tls_error(*out, (*out)->err);
        return rc;
    } catch (const std::exception& e) {
        if (*out) {
            (*out)->err = std::string("bbp_init: ") + e.what();
            set_tls_error(*out, (*out)->err);
        }
        return BBP_ERR_INTERNAL;
    } catch (...) {
        if (*out) {
            (*out)->err = "bbp_init: unknown exception";
            set_tls_error(*out, (*out)->err);
        }
        return BBP_ERR_INTERNAL;
    }
}

static int32_t init_body(int32_t device, bbp_ctx** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        fprintf(stderr, "bbp_init: no usable HIP device (count=%d, requested=%d); this library has no CPU path\n", ndev,
                device);
        return BBP_ERR_DEVICE;
    }
    bbp_ctx* ctx = new bbp_ctx();
    ctx->device = device;
    *out = ctx;  // returned even on failure so bbp_last_error works; caller frees
    ctx->combiner = new Combiner();
    if (const char* e = getenv("BBP_BATCH_WINDOW_US")) static_cast<Combiner*>(ctx->combiner)->configure((uint32_t)atoi(e), 0);
    {  // the prover's opening stage lasts ~36 ms whatever the batch size: a second batch starts no earlier
        const char* e = getenv("BBP_BATCH_STAGGER_US");
        static_cast<Combiner*>(ctx->combiner)->set_stagger(e ? (uint32_t)atoi(e) : 35000u);
        static_cast<Combiner*>(ctx->combiner)->set_leaders(1, bbp_ctx::VLANES);  // one verification batch in flight per verifier lane
        if (const char* lw = getenv("BBP_BATCH_LOPSIDED_WAIT")) static_cast<Combiner*>(ctx->combiner)->set_lopsided_wait(atoi(lw) != 0);
        if (const char* pl = getenv("BBP_BATCH_PROVE_LEADERS")) static_cast<Combiner*>(ctx->combiner)->set_leaders(0, atoi(pl));  // prove batches in flight (default 2; the host path has three slots)
        const char* ss = getenv("BBP_BATCH_STAGGER_SMALL_US");  // behind a batch of at most 256 proofs (cooperative rng chain: a ~13 ms opening stage)
        static_cast<Combiner*>(ctx->combiner)->set_small_stagger(256, ss ? (uint32_t)atoi(ss) : 15000u);
        const char* sp = getenv("BBP_BATCH_SPLIT_MIN");  // 1024: halves that still run at the engine's large-batch rate
        static_cast<Combiner*>(ctx->combiner)->set_split_min(sp ? (uint32_t)atoi(sp) : 1024u);
        const char* qc = getenv("BBP_BATCH_QUIET_CAP_US");  // a next batch's opening stage (~40 ms) fits under a 1024-proof MSM stage (~48 ms) with 8 ms to spare
        const char* qu = getenv("BBP_BATCH_QUIET_US");
        static_cast<Combiner*>(ctx->combiner)->set_quiet(qu ? (uint32_t)atoi(qu) : 300u, qc ? (uint32_t)atoi(qc) : 8000u);
        const char* hm = getenv("BBP_BATCH_HOLD_MARGIN_US");  // -1 = off
        static_cast<Combiner*>(ctx->combiner)->set_hold(hm ? atoi(hm) : 4000, 40000u, 48.0, !getenv("BBP_BATCH_HOLD_FIXED"));  // ~40 ms opening stage, ~48 us per proof
    }
    if (getenv("BBP_TRACE_ALLOC")) {  // the field map that goes with dev_reserve's "[bbp alloc] ... context offset" lines
#define BBP_OFF(f) (size_t)(reinterpret_cast<const char*>(&ctx->f) - reinterpret_cast<const char*>(ctx))
        fprintf(stderr, "[bbp alloc] field offsets: scal %zu idx %zu sorted %zu pts %zu enc %zu batch[0] %zu (+%zu each, %d prover then %d verifier) io_in %zu raw[0] %zu "
                        "slice_sorted[0] %zu slice_pts[0] %zu slice_fold[0] %zu slice_vtab[0] %zu vl[0].misc %zu vl[0].agg %zu vl[1].misc %zu io[0].in %zu io[1].in %zu\n",
                BBP_OFF(scal), BBP_OFF(idx), BBP_OFF(sorted), BBP_OFF(pts), BBP_OFF(enc), BBP_OFF(batch[0]), sizeof(bbp::DevBuf), (int)bbp_ctx::PROVE_BUFS, (int)bbp_ctx::VLANES,
                BBP_OFF(io_in), BBP_OFF(raw[0]), BBP_OFF(slice_sorted[0]), BBP_OFF(slice_pts[0]), BBP_OFF(slice_fold[0]), BBP_OFF(slice_vtab[0]), BBP_OFF(vl[0].misc),
                BBP_OFF(vl[0].agg), BBP_OFF(vl[1].misc), BBP_OFF(io[0].in), BBP_OFF(io[1].in));
#undef BBP_OFF
    }
    BBP_HIP_TRY(ctx, hipSetDevice(device));
    hipDeviceProp_t prop;
    BBP_HIP_TRY(ctx, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        ctx->err = std::string("bbp_init: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
        return BBP_ERR_DEVICE;
    }
    BBP_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    BBP_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    for (int i = 1; i < bbp_ctx::MAX_SLICES; i++) {
        BBP_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->lane[i], hipStreamNonBlocking));
        BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming));
        BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_stagger[i - 1], hipEventDisableTiming));
    }
    // created AFTER the engine's own streams: hardware queues are handed out in creation order, and the four streams the heavy
    // stage keeps busy (stream, side, lane[1], lane[2]) must not share one (measured: with the copy stream created third, a
    // 1024-proof batch took 60.9 instead of 55.6 ms)
    BBP_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
    // the small-batch path's second opening stream, BEFORE the verifier lanes: with 8 hardware queues (what bbp-uds-server exports) the seventh
    // stream still gets a queue of its own; created lazily, as the eleventh, it shared lane[1]'s, and a 10 ms opening stage sat in the same
    // queue as a heavy-stage chain (through the socket, 8 000 proofs/s offered: prove p50 / p99 55 / 80 -> 41.5 / 51.5 ms; 16 000: 111 / 149 -> 88 / 108)
    BBP_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->side2, hipStreamNonBlocking));
    for (auto& L : ctx->vl) {
#ifdef BBP_VL_SKIP  // experiment: unused streams after the first verifier lane, so that with 8 hardware queues the other lanes skip the queues of the caller's and the opening stream
        if (&L == &ctx->vl[1])
            for (int k = 0; k < BBP_VL_SKIP; k++) {
                hipStream_t d;
                BBP_HIP_TRY(ctx, hipStreamCreateWithFlags(&d, hipStreamNonBlocking));
                ctx->spare_streams.push_back(d);
            }
#endif
        BBP_HIP_TRY(ctx, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&L.ev_vfork, hipEventDisableTiming));
        BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&L.ev_vjoin, hipEventDisableTiming));
    }
    for (int f = 0; f < bbp_ctx::FAMILIES; f++) BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_last[f], hipEventDisableTiming));
    for (auto& sl : ctx->io) {
        BBP_HIP_TRY(ctx, hipHostMalloc((void**)&sl.h_flag, sizeof(uint32_t), hipHostMallocDefault));
        *sl.h_flag = 0;
        BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
        BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
    }
    BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming));
    if (const char* e = getenv("BBP_VERIFY_OVERLAP")) ctx->verify_overlap = atoi(e) != 0;
    if (const char* e = getenv("BBP_VERIFY_SERIAL_ACC")) ctx->verify_serial_acc = atoi(e) != 0;
    for (auto& e : ctx->ev_vacc) BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (const char* e = getenv("BBP_VERIFY_AGGREGATE")) ctx->verify_group = atoi(e) > 1 ? (uint32_t)atoi(e) : 0u;
    if (const char* e = getenv("BBP_DUAL_OPEN_BELOW")) ctx->dual_open_below = atoi(e);
    if (const char* e = getenv("BBP_ROTATE_BELOW")) ctx->rotate_below = atoi(e);
    if (const char* e = getenv("BBP_ROTATE_MIXED_FROM")) ctx->mixed_from = atoi(e);
    ctx->trace_prove = getenv("BBP_TRACE_PROVE") != nullptr;
    if (const char* e = getenv("BBP_ROTATE_DEEP_MAX")) ctx->rotate_deep_max = atoi(e);
    if (const char* e = getenv("BBP_ROTATE_DEEP_FROM")) ctx->deep_from = atoi(e) < 2 ? 2 : atoi(e);
    if (const char* e = getenv("BBP_VARBASE_LANES")) ctx->varbase_lanes = atoi(e) < 64 ? 64 : atoi(e);
    if (const char* e = getenv("BBP_TAIL_ROUND")) ctx->tail_round = atoi(e) == bbp::FOLD_ROUND ? bbp::FOLD_ROUND : 12;
    if (const char* e = getenv("BBP_STAGGER")) ctx->stagger_mode = atoi(e);
    if (const char* e = getenv("BBP_RNG_COOP")) ctx->rng_coop = atoi(e) != 0;
    if (const char* e = getenv("BBP_RNG_DPP")) ctx->rng_dpp = atoi(e);
    if (const char* e = getenv("BBP_TAIL_SMALL_BELOW")) ctx->tail_small_below = atoi(e);
    if (const char* e = getenv("BBP_RNG_COOP_BELOW")) ctx->rng_coop_below = atoi(e);
    if (const char* e = getenv("BBP_RNG_COOP_IDLE_BELOW")) ctx->rng_coop_idle_below = atoi(e);
    if (const char* e = getenv("BBP_MSM_SMALL")) ctx->msm_small = atoi(e) != 0;
    if (const char* e = getenv("BBP_WITNESS_NATIVE")) ctx->witness_native = atoi(e) != 0;
    if (const char* e = getenv("BBP_COMMIT_SPLIT_BELOW")) ctx->commit_split_below = atoi(e);
    if (const char* e = getenv("BBP_IPA_WIDE_BELOW")) ctx->ipa_wide_below = atoi(e);
    if (const char* e = getenv("BBP_TR_WAVE_BELOW")) ctx->tr_wave_below = atoi(e);
    if (const char* e = getenv("BBP_RNG_BLOCK")) {
        const int v = atoi(e);
        ctx->rng_block = v >= 1024 ? 1024 : v >= 512 ? 512 : v >= 256 ? 256 : v >= 128 ? 128 : 64;
    }
    if (const char* e = getenv("BBP_SERIAL_BLOCK")) ctx->serial_block = atoi(e) == 64 ? 64 : atoi(e) == 128 ? 128 : 256;
    if (const char* e = getenv("BBP_SERIAL_LDS")) ctx->serial_lds = atoi(e) < 0 ? 0 : atoi(e) > 160 * 1024 ? 160 * 1024 : atoi(e);
    if (const char* e = getenv("BBP_SORT_STAGED")) ctx->sort_staged = atoi(e) & 7;
    if (const char* e = getenv("BBP_FOLD_HALF_FROM")) ctx->fold_half_from = atoi(e) < 1 ? 1 : atoi(e);
    if (const char* e = getenv("BBP_SLICES")) ctx->slices = atoi(e) < 1 ? 1 : atoi(e) > bbp_ctx::MAX_SLICES ? bbp_ctx::MAX_SLICES : atoi(e);
    for (int i = 0; i < bbp_ctx::PROVE_BUFS; i++) {
        BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_entry[i], hipEventDisableTiming));
        BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_open[i], hipEventDisableTiming));
        BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_done[i], hipEventDisableTiming));
    }
    for (auto& e : ctx->ev_call) BBP_HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));

    // --- host hashing -----------------------------------------------------------------------------
    const size_t n_uniform = 2 * BBP_GENS_CAPACITY + 1;
    std::vector<uint8_t> uniform(n_uniform * 64);
    // PedersenGens::default(): B_blinding = hash_from_bytes::<Sha3_512>(compress(B))
    static const uint8_t B_ENC[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                      0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
    sha3_512(B_ENC, 32, &uniform[0]);
    // BulletproofGens::new(2048, 1): GeneratorsChain = SHAKE256("GeneratorsChain" || label), label = 'G'/'H' || u32_le(0)
    for (int which = 0; which < 2; which++) {
        uint8_t seed[20] = {'G', 'e', 'n', 'e', 'r', 'a', 't', 'o', 'r', 's', 'C', 'h', 'a', 'i', 'n', (uint8_t)(which ? 'H' : 'G'), 0, 0, 0, 0};
        shake256(seed, 20, &uniform[64 * (1 + (size_t)which * BBP_GENS_CAPACITY)], 64 * BBP_GENS_CAPACITY);
    }
    mimc_constants_host(ctx->mimc_host);

    // --- device tables ----------------------------------------------------------------------------
    u32* d_uniform = nullptr;
    BBP_HIP_TRY(ctx, hipMalloc(&d_uniform, uniform.size()));
    BBP_HIP_TRY(ctx, hipMemcpyAsync(d_uniform, uniform.data(), uniform.size(), hipMemcpyHostToDevice, ctx->stream));
    BBP_HIP_TRY(ctx, hipMalloc(&ctx->gens, sizeof(ge) * TAB_BASES));
    BBP_HIP_TRY(ctx, hipMalloc(&ctx->ptable, sizeof(niels_row) * (size_t)TAB_BASES * MSM_POS));
    BBP_HIP_TRY(ctx, hipMalloc(&ctx->comb, sizeof(niels_packed) * 2 * 64 * 8));
    BBP_HIP_TRY(ctx, hipMalloc(&ctx->mimc_c, sizeof(sc) * BBP_MIMC_ROUNDS));
    BBP_HIP_TRY(ctx, hipMalloc(&ctx->health, sizeof(u32)));
    BBP_HIP_TRY(ctx, hipMemset(ctx->health, 0, sizeof(u32)));
    BBP_HIP_TRY(ctx, hipMemcpyAsync(ctx->mimc_c, ctx->mimc_host.data(), 32 * BBP_MIMC_ROUNDS, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_derive_generators, dim3((BBP_NUM_BASES + 63) / 64), dim3(64), 0, ctx->stream, d_uniform, ctx->gens);
    BBP_HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_pad_sums, dim3(1), dim3(64), 0, ctx->stream, ctx->gens);
    BBP_HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_merge_sums, dim3(2048 / 64), dim3(64), 0, ctx->stream, ctx->gens);
    BBP_HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_build_ptable, dim3((TAB_BASES * (MSM_POS / PT_CHUNK) + 63) / 64), dim3(64), 0, ctx->stream, ctx->gens, ctx->ptable);
    BBP_HIP_TRY(ctx, hipGetLastError());
    if (int32_t rc = tail_btab_build(ctx)) return rc;
    hipLaunchKernelGGL(k_build_comb, dim3(2), dim3(64), 0, ctx->stream, ctx->gens, ctx->comb);
    BBP_HIP_TRY(ctx, hipGetLastError());
    BBP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    BBP_HIP_TRY(ctx, hipFree(d_uniform));
    return BBP_OK;
}

extern "C" void bbp_free(bbp_ctx* ctx) {
    if (!ctx) return;
    if (is_pool(ctx)) {  // a pool owns its members and its combiner, no device state
        delete static_cast<Combiner*>(ctx->combiner);  // FIRST: its threads run what is still queued on the members, then leave
        ctx->combiner = nullptr;
        pool_workers_stop(ctx);  // ... then the members' worker threads (they finish the blocks they hold)
        for (bbp_ctx* m : ctx->members) bbp_free(m);
        ctx->members.clear();
        delete ctx;
        return;
    }
    if (!ctx->stream && !ctx->gens && ctx->device < 0) {  // a pool whose initialisation failed before it had members
        delete static_cast<Combiner*>(ctx->combiner);
        delete ctx;
        return;
    }
    // FIRST the combiner: its threads run whatever asynchronous requests are still queued (their callbacks fire) and leave; only then
    // is it safe to take the device state away
    delete static_cast<Combiner*>(ctx->combiner);
    ctx->combiner = nullptr;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    std::vector<void*> ptrs = {ctx->gens, ctx->ptable, ctx->btab, ctx->comb, ctx->mimc_c, ctx->scal.p, ctx->idx.p, ctx->sorted.p, ctx->pts.p, ctx->enc.p,
                               ctx->io_in.p, ctx->io_out.p, ctx->io_ent.p, ctx->raw[0].p, ctx->raw[1].p};
    for (auto& b : ctx->batch) ptrs.push_back(b.p);
    for (auto& b : ctx->slice_sorted) ptrs.push_back(b.p);
    for (auto& b : ctx->slice_pts) ptrs.push_back(b.p);
    for (auto& b : ctx->slice_fold) ptrs.push_back(b.p);
    for (auto& b : ctx->slice_vtab) ptrs.push_back(b.p);
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (auto& kv : ctx->circuits) {  // compiled circuits (one per list length used)
        bbp::CircuitDev* c = static_cast<bbp::CircuitDev*>(kv.second);
        if (!c) continue;
        void* cp[] = {c->w_terms, c->w_loff, c->w_roff, c->f_off, c->f_ent, c->c_q, c->c_cst, c->idx_ai, c->idx_s1, c->idx_ao, c->idx_ipa, c->idx_ver};
        for (void* p : cp)
            if (p) (void)hipFree(p);
        delete c;
    }
    ctx->circuits.clear();
    for (auto& sl : ctx->io) {
        for (void* p : {sl.in.p, sl.ent.p, sl.out.p})
            if (p) (void)hipFree(p);
        if (sl.h_flag) (void)hipHostFree(sl.h_flag);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.ev) (void)hipEventDestroy(sl.ev);
        if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
    }
    if (ctx->health) (void)hipFree(ctx->health);
    for (auto& e : ctx->ev_vacc)
        if (e) (void)hipEventDestroy(e);
    for (auto& L : ctx->vl) {
        for (void* p : {L.misc.p, L.agg.p, L.agg_io.p, (void*)L.agg_count})
            if (p) (void)hipFree(p);
        if (L.ev_vfork) (void)hipEventDestroy(L.ev_vfork);
        if (L.ev_vjoin) (void)hipEventDestroy(L.ev_vjoin);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    for (auto& kv : ctx->layout_idx)
        if (kv.second) (void)hipFree(kv.second);
    ctx->layout_idx.clear();
    for (int i = 0; i < bbp_ctx::PROVE_BUFS; i++) {
        if (ctx->ev_entry[i]) (void)hipEventDestroy(ctx->ev_entry[i]);
        if (ctx->ev_open[i]) (void)hipEventDestroy(ctx->ev_open[i]);
        if (ctx->ev_done[i]) (void)hipEventDestroy(ctx->ev_done[i]);
    }
    for (auto& e : ctx->ev_call)
        if (e) (void)hipEventDestroy(e);
    for (int i = 1; i < bbp_ctx::MAX_SLICES; i++) {
        if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
        if (ctx->ev_stagger[i - 1]) (void)hipEventDestroy(ctx->ev_stagger[i - 1]);
        if (ctx->lane[i]) (void)hipStreamDestroy(ctx->lane[i]);
    }
    for (int f = 0; f < bbp_ctx::FAMILIES; f++)
        if (ctx->ev_last[f]) (void)hipEventDestroy(ctx->ev_last[f]);
    if (ctx->ev_prep) (void)hipEventDestroy(ctx->ev_prep);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->copy) (void)hipStreamDestroy(ctx->copy);
    if (ctx->side2) (void)hipStreamDestroy(ctx->side2);
    for (hipStream_t d : ctx->spare_streams) (void)hipStreamDestroy(d);
    ctx->spare_streams.clear();
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// per calling thread: the message of this thread's last failing call (api_guard copies it under the context lock); a thread
// that has not failed yet sees the context's last message (single-threaded callers never notice the difference)
extern "C" const char* bbp_last_error(const bbp_ctx* ctx) {
    if (!ctx) return "null context";
    try {
        if (tls_error_owner() == ctx) return tls_error().c_str();  // this thread's last failure on THIS handle
        // this thread has not failed on this handle: the handle's own last message, through a buffer that leaves the slot alone
        static thread_local std::string other;
        std::lock_guard<std::recursive_mutex> lk(const_cast<bbp_ctx*>(ctx)->mu);
        other = ctx->err;
        return other.c_str();
    } catch (...) {
        return "error text unavailable";
    }
}

// (a pool handle has no streams: nullptr)
extern "C" void* bbp_context_stream(bbp_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
extern "C" void* bbp_context_copy_stream(bbp_ctx* ctx) { return ctx ? (void*)ctx->copy : nullptr; }
extern "C" void* bbp_context_verify_stream(bbp_ctx* ctx, uint32_t lane) { return ctx && lane < bbp_ctx::VLANES ? (void*)ctx->vl[lane].stream : nullptr; }

// What this context runs on and how it is set up, as text; the conditions that are known to cost throughput silently are called
// out as WARNING lines (the server logs the report at start-up).
extern "C" int32_t bbp_describe(bbp_ctx* ctx, char* buf, uint32_t cap) {
    if (!ctx || !buf || cap == 0) return BBP_ERR_BAD_ARG;
    buf[0] = 0;
    if (is_pool(ctx)) {
        uint32_t off = (uint32_t)snprintf(buf, cap, "device pool of %zu member context(s)\n", ctx->members.size());
        for (size_t i = 0; i < ctx->members.size() && off + 1 < cap; i++) {
            off += (uint32_t)snprintf(buf + off, cap - off, "member %zu (worker threads on NUMA node %d): ", i, pool_member_numa_node(ctx, (uint32_t)i));
            if (off + 1 >= cap) break;
            if (int32_t rc = bbp_describe(ctx->members[i], buf + off, cap - off)) return rc;
            off += (uint32_t)strlen(buf + off);
        }
        return BBP_OK;
    }
    return api_guard(ctx, [&]() -> int32_t {
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        hipDeviceProp_t prop;
        BBP_HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
        size_t free_b = 0, total_b = 0;
        BBP_HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        const char* hwq = getenv("GPU_MAX_HW_QUEUES");
        uint32_t off = (uint32_t)snprintf(buf, cap,
                                          "device %d: %s (%s), %d CUs, %.0f MHz, %.1f of %.1f GiB free; prover: %d heavy-stage slices, tail from round %d, "
                                          "cooperative rng below %d proofs; verifier: %d lanes, accumulates %s, aggregate groups of %u%s\n",
                                          ctx->device, prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1e3, free_b / 1073741824.0,
                                          total_b / 1073741824.0, ctx->slices, ctx->tail_round, ctx->rng_coop_below, (int)bbp_ctx::VLANES,
                                          ctx->verify_serial_acc ? "chained" : "free-running", ctx->verify_group, ctx->verify_group ? "" : " (off)");
        // The engine keeps four streams busy during a prove call (caller's, opening stage, two more slices) and four more for
        // verification (one per lane), out of eleven it creates: HIP's default of four hardware queues makes them share and serialise
        // (measured 84.5 vs 61.6 ms per prove batch).  How many are best depends on the mix: 8 for prove-dominated host-pointer use
        // (the UDS server: the seven prover streams are created first and have a queue each; a few percent better than 16 at saturation), 16 for verification on all four lanes through the device API
        // (1024 per call: 5.2 ms against 7.2 with 8): INTEGRATION.md section 5.  The variable is read when the HIP runtime
        // initialises, i.e. possibly long before bbp_init.
        const int hst = hw_queues_state();
        if (off + 1 < cap)
            off += (uint32_t)snprintf(buf + off, cap - off, "hardware queues: GPU_MAX_HW_QUEUES=%s (%s); scratch: %.2f GiB in %llu allocation(s) since bbp_init\n",
                                      hwq ? hwq : "unset",
                                      hst == 2 ? "exported by the library before HIP initialised" : hst == 1 ? "from the caller's environment" : hst == 3 ? "unset and HIP was already initialised" : "?",
                                      ctx->scratch_bytes / 1073741824.0, (unsigned long long)ctx->scratch_allocs);
        if ((hst == 3 || !hwq || atoi(hwq) < 8) && off + 1 < cap)
            off += (uint32_t)snprintf(buf + off, cap - off,
                                      "WARNING: GPU_MAX_HW_QUEUES is %s%s: the HIP runtime reads it once, when it initialises; export GPU_MAX_HW_QUEUES=" BBP_HWQ_DEFAULT
                                      " before the process first touches HIP (or call bbp_init first: it does so itself), or streams of this context share "
                                      "hardware queues and serialise (prover up to ~30 %% slower with the default of 4)\n",
                                      hwq ? hwq : "not set", hst == 3 ? " and HIP was initialised before bbp_init could set it" : "");
        if (free_b < ((size_t)6 << 30) && off + 1 < cap)
            off += (uint32_t)snprintf(buf + off, cap - off, "WARNING: less than 6 GiB of device memory free: a 1024-proof batch needs ~14 GiB of scratch\n");
        return BBP_OK;
    });
}

// Test hook: the next MSM launch of this context finds an out-of-range entry in its sorted scratch, as a stray write would leave it.
extern "C" int32_t bbp_debug_corrupt_scratch(bbp_ctx* ctx) {
    if (!ctx) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) return bbp_debug_corrupt_scratch(ctx->members[0]);
    return api_guard(ctx, [&]() -> int32_t {
        ctx->debug_corrupt = 1;
        return BBP_OK;
    });
}

// Synchronises the device.  *flags: bit 0 = some MSM table gather since bbp_init was out of range and had to be clamped -- the
// engine's scratch was corrupted and results computed since then may be wrong (never observed; the clamp exists so that such a
// state cannot fault the GPU, this word so that it cannot pass silently).
extern "C" int32_t bbp_check_health(bbp_ctx* ctx, uint32_t* flags) {
    if (!ctx || !flags) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) {  // the members' words OR-ed
        *flags = 0;
        for (bbp_ctx* m : ctx->members) {
            uint32_t f = 0;
            const int32_t rc = bbp_check_health(m, &f);
            if (rc) return rc;
            *flags |= f;
        }
        return BBP_OK;
    }
    return api_guard(ctx, [&]() -> int32_t {
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        BBP_HIP_TRY(ctx, hipDeviceSynchronize());
        BBP_HIP_TRY(ctx, hipMemcpy(flags, ctx->health, sizeof(u32), hipMemcpyDeviceToHost));
        return BBP_OK;
    });
}

extern "C" int32_t bbp_get_generator(bbp_ctx* ctx, uint32_t index, uint8_t out32[32]) {
    if (!ctx || index >= BBP_NUM_BASES) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) return bbp_get_generator(ctx->members[0], index, out32);  // every member holds the same tables
    return api_guard(ctx, [&]() -> int32_t {
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        int32_t rc = dev_reserve(ctx, ctx->enc, 32);
        if (rc) return rc;
        StreamGuard guard(ctx, ctx->stream);
        if ((rc = guard.enter())) return rc;
        rc = encode_launch(ctx, 1, ctx->gens + index, (uint8_t*)ctx->enc.p, ctx->stream);
        if (rc) return rc;
        BBP_HIP_TRY(ctx, hipMemcpyAsync(out32, ctx->enc.p, 32, hipMemcpyDeviceToHost, ctx->stream));
        BBP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return BBP_OK;
    });
}

extern "C" int32_t bbp_get_mimc_constant(bbp_ctx* ctx, uint32_t i, uint8_t out32[32]) {
    if (!ctx || i >= BBP_MIMC_ROUNDS) return BBP_ERR_BAD_ARG;
    if (is_pool(ctx)) return bbp_get_mimc_constant(ctx->members[0], i, out32);
    return api_guard(ctx, [&]() -> int32_t {
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        BBP_HIP_TRY(ctx, hipMemcpy(out32, ctx->mimc_c + i, 32, hipMemcpyDeviceToHost));  // read back from the device copy
        return BBP_OK;
    });
}
