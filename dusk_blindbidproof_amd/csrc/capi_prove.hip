// C-ABI: prove / verify entry points -- see include/bbp.h.  (Filled in incrementally; NOT_YET paths return BBP_ERR_DEVICE.)
#include "context.h"

using namespace bbp;

#define NOT_YET(ctx, name)                                  \
    do {                                                    \
        if (ctx) (ctx)->err = name ": not implemented yet"; \
        return BBP_ERR_DEVICE;                              \
    } while (0)

extern "C" uint32_t bbp_proof_record_size(uint32_t N) { return BBP_R1CS_PROOF_BYTES + 32u * (4u + N); }
extern "C" uint32_t bbp_entropy_size(uint32_t N) { return 32u * (4u + N) + 32u; }

extern "C" int32_t bbp_witness_batch(bbp_ctx* ctx, uint32_t, const uint8_t*, uint8_t*) { NOT_YET(ctx, "bbp_witness_batch"); }
extern "C" int32_t bbp_prove(bbp_ctx* ctx, const uint8_t*, const uint8_t*, uint32_t, uint64_t, const uint8_t*, uint8_t*, uint32_t*) { NOT_YET(ctx, "bbp_prove"); }
extern "C" int32_t bbp_verify(bbp_ctx* ctx, const uint8_t*, uint32_t, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, uint32_t) { NOT_YET(ctx, "bbp_verify"); }
extern "C" int32_t bbp_prove_batch(bbp_ctx* ctx, uint32_t, uint32_t, const uint8_t*, const uint8_t*, uint8_t*, int32_t*) { NOT_YET(ctx, "bbp_prove_batch"); }
extern "C" int32_t bbp_verify_batch(bbp_ctx* ctx, uint32_t, uint32_t, const uint8_t*, int32_t*) { NOT_YET(ctx, "bbp_verify_batch"); }
extern "C" int32_t bbp_set_profiling(bbp_ctx* ctx, int32_t on) {
    if (!ctx) return BBP_ERR_BAD_ARG;
    ctx->profile = on != 0;
    return BBP_OK;
}

// Drains the recorded events: out[2i] = kernel tag, out[2i+1] = microseconds. Synchronises the device.
extern "C" int32_t bbp_last_timings(bbp_ctx* ctx, float* out, uint32_t cap, uint32_t* n) {
    if (!ctx || !n) return BBP_ERR_BAD_ARG;
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
}
