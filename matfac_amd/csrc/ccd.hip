// ccd.hip -- placeholder until the CCD++ kernels land.
#include "mfx_internal.h"
void mfx_ccd_free_internal(mfx_ctx* ctx) {
  dev_free(ctx->res_row); dev_free(ctx->res_col); dev_free(ctx->uk); dev_free(ctx->vk);
  ctx->ccd_active = false;
}
extern "C" int mfx_ccdpp_begin(mfx_ctx* ctx) { if (!ctx) return MFX_E_ARG; return mfx_fail(ctx, MFX_E_STATE, "not implemented yet"); }
extern "C" int mfx_ccdpp_rank1(mfx_ctx* ctx, int32_t, int32_t, float, float, int32_t, float) { if (!ctx) return MFX_E_ARG; return mfx_fail(ctx, MFX_E_STATE, "not implemented yet"); }
extern "C" int mfx_ccdpp_end(mfx_ctx* ctx) { if (!ctx) return MFX_E_ARG; return mfx_fail(ctx, MFX_E_STATE, "not implemented yet"); }
extern "C" int mfx_debug_residuals(mfx_ctx* ctx, float*, float*) { if (!ctx) return MFX_E_ARG; return mfx_fail(ctx, MFX_E_STATE, "not implemented yet"); }
