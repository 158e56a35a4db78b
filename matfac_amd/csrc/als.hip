// als.hip -- placeholder until the MFMA Gramian + batched solve kernels land.
#include "mfx_internal.h"
extern "C" int mfx_als_half_sweep(mfx_ctx* ctx, int side, float reg) {
  (void)side; (void)reg;
  if (!ctx) return MFX_E_ARG;
  return mfx_fail(ctx, MFX_E_STATE, "mfx_als_half_sweep: not implemented yet");
}
