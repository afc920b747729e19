// C-ABI: kernel-level MSM hook (BASELINE.json configs[1]) -- see include/bbp.h.
#include <string.h>

#include "context.h"

using namespace bbp;

namespace bbp {

// base indices (generator-table order) of a layout's terms
int32_t layout_base_indices(uint32_t layout, uint32_t n_terms, std::vector<u32>& idx, std::string& err) {
    idx.clear();
    if (layout == BBP_LAYOUT_BLIND_G_H) {
        if (n_terms < 1 || (n_terms - 1) % 2 != 0 || (n_terms - 1) / 2 > BBP_GENS_CAPACITY) {
            err = "layout 0 needs n_terms = 1 + 2m, m <= 2048";
            return BBP_ERR_BAD_ARG;
        }
        u32 m = (n_terms - 1) / 2;
        idx.push_back(BBP_BASE_BBLIND);
        for (u32 i = 0; i < m; i++) idx.push_back(BBP_BASE_G0 + i);
        for (u32 i = 0; i < m; i++) idx.push_back(BBP_BASE_H0 + i);
        return BBP_OK;
    }
    if (layout == BBP_LAYOUT_BLIND_G) {
        if (n_terms < 1 || n_terms - 1 > BBP_GENS_CAPACITY) {
            err = "layout 1 needs n_terms = 1 + m, m <= 2048";
            return BBP_ERR_BAD_ARG;
        }
        idx.push_back(BBP_BASE_BBLIND);
        for (u32 i = 0; i + 1 < n_terms; i++) idx.push_back(BBP_BASE_G0 + i);
        return BBP_OK;
    }
    err = "unknown layout";
    return BBP_ERR_BAD_ARG;
}

}  // namespace bbp

namespace bbp {

// the per-(layout, n_terms) base-index list lives on the device once (like circuit_get's idx_ai / idx_ao): nothing is rebuilt or
// uploaded per call, and no function-local staging buffer is ever the source of an asynchronous copy
static int32_t layout_idx_dev(bbp_ctx* ctx, uint32_t layout, uint32_t n_terms, const u32** out) {
    const uint64_t key = ((uint64_t)layout << 32) | n_terms;
    auto it = ctx->layout_idx.find(key);
    if (it != ctx->layout_idx.end()) {
        *out = it->second;
        return BBP_OK;
    }
    std::vector<u32> idx;
    int32_t rc = layout_base_indices(layout, n_terms, idx, ctx->err);
    if (rc) return rc;
    u32* d = nullptr;
    BBP_HIP_TRY(ctx, hipMalloc(&d, idx.size() * 4));
    if (hipMemcpy(d, idx.data(), idx.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {  // synchronous: idx may go out of scope
        (void)hipFree(d);
        ctx->err = "layout index upload failed";
        return BBP_ERR_DEVICE;
    }
    ctx->layout_idx[key] = d;
    *out = d;
    return BBP_OK;
}

static int32_t msm_batch_dev(bbp_ctx* ctx, uint32_t B, uint32_t n_terms, const void* scalars_dev, uint32_t layout, void* out32_dev,
                             hipStream_t stream) {
    BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const u32* idx_dev;
    int32_t rc = layout_idx_dev(ctx, layout, n_terms, &idx_dev);
    if (rc) return rc;
    if ((rc = dev_reserve(ctx, ctx->pts, sizeof(ge) * (size_t)B))) return rc;
    StreamGuard guard(ctx, stream);
    if ((rc = guard.enter())) return rc;
    if ((rc = msm_launch(ctx, B, n_terms, (const u32*)scalars_dev, idx_dev, (ge*)ctx->pts.p, stream))) return rc;
    return encode_launch(ctx, B, (const ge*)ctx->pts.p, (uint8_t*)out32_dev, stream);
}

}  // namespace bbp

extern "C" int32_t bbp_msm_batch_dev(bbp_ctx* ctx, uint32_t B, uint32_t n_terms, const void* scalars_dev, uint32_t layout,
                                     void* out32_dev, void* stream_) {
    if (!ctx || !scalars_dev || !out32_dev) return BBP_ERR_BAD_ARG;
    if (B == 0) return BBP_OK;
    if (is_pool(ctx)) return pool_reject(ctx, "bbp_msm_batch_dev");
    return api_guard(ctx, [&]() -> int32_t { return msm_batch_dev(ctx, B, n_terms, scalars_dev, layout, out32_dev, pick_stream(ctx, stream_)); });
}

extern "C" int32_t bbp_msm_batch(bbp_ctx* ctx, uint32_t B, uint32_t n_terms, const uint8_t* scalars, uint32_t layout,
                                 uint8_t* out32) {
    if (!ctx || !scalars || !out32) return BBP_ERR_BAD_ARG;
    if (B == 0) return BBP_OK;
    if (is_pool(ctx)) {  // block split over the members (pool.cpp)
        try {
            return pool_msm_batch(ctx, B, n_terms, scalars, layout, out32);
        } catch (...) {
            return BBP_ERR_INTERNAL;
        }
    }
    return api_guard(ctx, [&]() -> int32_t {
        BBP_HIP_TRY(ctx, hipSetDevice(ctx->device));
        // canonical scalars only (the reference's Scalars are always reduced)
        const size_t total = (size_t)B * n_terms;
        for (size_t i = 0; i < total; i++) {
            u32 w[8];
            memcpy(w, scalars + 32 * i, 32);
            if (!sc_is_canonical(w)) {
                ctx->err = "bbp_msm_batch: non-canonical scalar";
                return BBP_ERR_FORMAT;
            }
        }
        int32_t rc = dev_reserve(ctx, ctx->scal, total * 32);
        if (rc) return rc;
        rc = dev_reserve(ctx, ctx->enc, (size_t)B * 32);
        if (rc) return rc;
        BBP_HIP_TRY(ctx, hipMemcpyAsync(ctx->scal.p, scalars, total * 32, hipMemcpyHostToDevice, ctx->stream));
        rc = msm_batch_dev(ctx, B, n_terms, ctx->scal.p, layout, ctx->enc.p, ctx->stream);
        if (rc) return rc;
        BBP_HIP_TRY(ctx, hipMemcpyAsync(out32, ctx->enc.p, (size_t)B * 32, hipMemcpyDeviceToHost, ctx->stream));
        u32 flags = 0;
        BBP_HIP_TRY(ctx, hipMemcpyAsync(&flags, ctx->health, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
        BBP_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (flags) {  // (see fetch_results in capi_prove.hip)
            ctx->err = "engine health flags raised: an MSM table gather was out of range (corrupted engine scratch); results are not trustworthy";
            return BBP_ERR_DEVICE;
        }
        return BBP_OK;
    });
}
