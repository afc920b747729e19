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
extern "C" int32_t bbp_last_timings(bbp_ctx* ctx, float* out, uint32_t cap, uint32_t* n) {
    if (!ctx || !n) return BBP_ERR_BAD_ARG;
    uint32_t k = (uint32_t)ctx->timings.size();
    if (k > cap) k = cap;
    for (uint32_t i = 0; i < k; i++) out[i] = ctx->timings[i];
    *n = k;
    return BBP_OK;
}
