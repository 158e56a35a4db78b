// sgd_bias.hip -- ModelMFBias, the bias-only sibling of ModelMF (modelMFBias.cpp): estRating = uBias[u] + iBias[item]
// (:94-99), the sequential SGD loop over shuffled rating tuples (:163-197) and the objective (:40-91).
//
// A visit reads and writes two floats, so there is nothing to tile: the loop is replayed in list order, either by one
// lane (MFX_SGD_SERIAL) or -- the same bits -- by the dataflow schedule of sgd_flow.hip with ONE lane per rating
// (MFX_SGD_LEVELS: the item's ratings stay with one lane, the user's bias carries a version counter).
#include <algorithm>
#include <vector>

#include "mfx_internal.h"

// The reference's statements, one by one (modelMFBias.cpp:178-197):
//   r_ui_est = estRating(u, item)      double <- float sum uBias[u] + iBias[item]
//   diff = itemRat - r_ui_est          double
//   uBias[u]    -= learnRate*(-2.0*diff + 2.0*uReg*uBias[u])        float -= double
//   iBias[item] -= learnRate*(-2.0*diff + 2.0*iReg*iBias[item])     (the SAME diff: taken before the user step)
__device__ __forceinline__ void mfx_bias_visit(float& bu, float& bi, float r, float lr, float uReg, float iReg) {
  const float est = bu + bi;
  const double diff = (double)r - (double)est;
  bu = (float)((double)bu - (double)lr * (-2.0 * diff + (2.0 * (double)uReg) * (double)bu));
  bi = (float)((double)bi - (double)lr * (-2.0 * diff + (2.0 * (double)iReg) * (double)bi));
}

__global__ __launch_bounds__(64) void bias_serial_kernel(const int32_t* __restrict__ eu, const int32_t* __restrict__ ei,
                                                         const float* __restrict__ er, int64_t first, int64_t count, float* ub, float* ib,
                                                         float lr, float uReg, float iReg) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int64_t t = 0; t < count; t++) {
    const int u = eu[first + t], it = ei[first + t];
    float bu = ub[u], bi = ib[it];
    mfx_bias_visit(bu, bi, er[first + t], lr, uReg, iReg);
    ub[u] = bu;
    ib[it] = bi;
  }
}

// sse / count / bias norms: per-block partials in a fixed order, finished on the host (reproducible)
__global__ __launch_bounds__(256) void bias_eval_kernel(const int32_t* __restrict__ ru, const int32_t* __restrict__ ri, const float* __restrict__ rv,
                                                        int64_t nnz, const float* __restrict__ ub, const float* __restrict__ ib,
                                                        const uint8_t* __restrict__ invU, const uint8_t* __restrict__ invI, int32_t nU, int32_t nI,
                                                        int with_norms, double* __restrict__ part) {
  double sse = 0, cnt = 0, un = 0, in = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t t = t0; t < nnz; t += stride) {
    const int u = ru[t], it = ri[t];
    if (u >= nU || it >= nI || invU[u] || invI[it]) continue;      // model.cpp:223-240: invalid users / items, items beyond nItems
    const float est = ub[u] + ib[it];
    const double d = (double)rv[t] - (double)est;
    sse += d * d;
    cnt += 1;
  }
  if (with_norms) {
    for (int64_t t = t0; t < nU; t += stride)
      if (!invU[t]) un += (double)(ub[t] * ub[t]);                 // uBias[u]*uBias[u]: a float product, summed in double
    for (int64_t t = t0; t < nI; t += stride)
      if (!invI[t]) in += (double)(ib[t] * ib[t]);
  }
  __shared__ double sh[4][256];
  sh[0][threadIdx.x] = sse; sh[1][threadIdx.x] = cnt; sh[2][threadIdx.x] = un; sh[3][threadIdx.x] = in;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int k = 0; k < 4; k++) sh[k][threadIdx.x] += sh[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int k = 0; k < 4; k++) part[(size_t)blockIdx.x * 4 + k] = sh[k][0];
}

void mfx_bias_free_internal(mfx_ctx* ctx) {
  dev_free(ctx->ub); dev_free(ctx->ib); dev_free(ctx->ub_best); dev_free(ctx->ib_best);
}

extern "C" int mfx_bias_set(mfx_ctx* ctx, const float* uBias, const float* iBias) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->U, MFX_E_STATE, "mfx_bias_set: call mfx_set_model first");
  NEED(uBias && iBias, MFX_E_ARG, "mfx_bias_set: NULL vector");
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  if (!ctx->ub) {
    if ((rc = dev_alloc(ctx, &ctx->ub, (size_t)ctx->nU)) || (rc = dev_alloc(ctx, &ctx->ib, (size_t)ctx->nI)) ||
        (rc = dev_alloc(ctx, &ctx->ub_best, (size_t)ctx->nU)) || (rc = dev_alloc(ctx, &ctx->ib_best, (size_t)ctx->nI)))
      return rc;
    HIPCHK(hipMemsetAsync(ctx->ub_best, 0, sizeof(float) * (size_t)ctx->nU, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->ib_best, 0, sizeof(float) * (size_t)ctx->nI, ctx->stream));
  }
  HIPCHK(hipMemcpyAsync(ctx->ub, uBias, sizeof(float) * (size_t)ctx->nU, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->ib, iBias, sizeof(float) * (size_t)ctx->nI, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MFX_OK;
}

extern "C" int mfx_bias_get(mfx_ctx* ctx, int snapshot, float* uBias, float* iBias) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->ub, MFX_E_STATE, "mfx_bias_get: no bias vectors (mfx_bias_set)");
  NEED(snapshot == MFX_SNAP_CURRENT || snapshot == MFX_SNAP_BEST, MFX_E_ARG, "mfx_bias_get: snapshot");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (uBias) HIPCHK(hipMemcpy(uBias, snapshot ? ctx->ub_best : ctx->ub, sizeof(float) * (size_t)ctx->nU, hipMemcpyDeviceToHost));
  if (iBias) HIPCHK(hipMemcpy(iBias, snapshot ? ctx->ib_best : ctx->ib, sizeof(float) * (size_t)ctx->nI, hipMemcpyDeviceToHost));
  return MFX_OK;
}

int mfx_launch_sgd_bias(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count) {
  NEED(ctx->ub, MFX_E_STATE, "mfx_bias_epoch: no bias vectors (mfx_bias_set)");
  if (o->mode == MFX_SGD_LEVELS && mfx_flow_usable(ctx, count)) return mfx_launch_bias_flow(ctx, o, first, count);
  ProfScope ps(ctx, MFX_K_SGD);
  hipLaunchKernelGGL(bias_serial_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->eu, ctx->ei, ctx->er, first, count, ctx->ub, ctx->ib,
                     o->learnRate, o->uReg, o->iReg);
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

extern "C" int mfx_bias_epoch(mfx_ctx* ctx, const mfx_sgd_opts* o) {
  if (!ctx) return MFX_E_ARG;
  NEED(o, MFX_E_ARG, "mfx_bias_epoch: opts NULL");
  NEED(ctx->ub, MFX_E_STATE, "mfx_bias_epoch: no bias vectors (mfx_bias_set)");
  NEED(o->mode == MFX_SGD_SERIAL || o->mode == MFX_SGD_LEVELS, MFX_E_ARG, "mfx_bias_epoch: mode must be MFX_SGD_SERIAL or MFX_SGD_LEVELS");
  NEED(!ctx->dimreg && !ctx->ifw && !ctx->tmf_u, MFX_E_STATE, "mfx_bias_epoch: an SGD variant of the factor model is installed");
  ctx->bias_epoch = true;                    // the epoch list is built by mfx_sgd_epoch, the visit is ours (sgd.hip dispatches)
  mfx_sgd_opts oo = *o;
  oo.arith = MFX_ARITH_REF64;
  const int rc = mfx_sgd_epoch(ctx, &oo);
  ctx->bias_epoch = false;
  return rc;
}

extern "C" int mfx_bias_eval(mfx_ctx* ctx, int which, int snapshot, mfx_eval_out* out) {
  if (!ctx) return MFX_E_ARG;
  NEED(out, MFX_E_ARG, "mfx_bias_eval: out NULL");
  NEED(which >= 0 && which < 3 && ctx->mat[which].present, MFX_E_STATE, "mfx_bias_eval: matrix %d not set", which);
  NEED(ctx->ub, MFX_E_STATE, "mfx_bias_eval: no bias vectors (mfx_bias_set)");
  NEED(ctx->have_invalid, MFX_E_STATE, "mfx_bias_eval: call mfx_compute_invalid first");
  NEED(snapshot == MFX_SNAP_CURRENT || snapshot == MFX_SNAP_BEST, MFX_E_ARG, "mfx_bias_eval: snapshot");
  HIPCHK(hipSetDevice(ctx->device));
  const DevCSR& m = ctx->mat[which];
  const int blocks = 512;
  double* part = nullptr;
  int rc = dev_alloc(ctx, &part, (size_t)blocks * 4);
  if (rc) return rc;
  hipLaunchKernelGGL(bias_eval_kernel, dim3(blocks), dim3(256), 0, ctx->stream, m.rowid, m.rowind, m.rowval, m.nnz,
                     snapshot ? ctx->ub_best : ctx->ub, snapshot ? ctx->ib_best : ctx->ib, ctx->invU, ctx->invI, ctx->nU, ctx->nI,
                     which == MFX_MAT_TRAIN ? 1 : 0, part);
  std::vector<double> h((size_t)blocks * 4);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), part, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  dev_free(part);
  NEED(e == hipSuccess, MFX_E_HIP, "mfx_bias_eval: %s", hipGetErrorString(e));
  double s[4] = {0, 0, 0, 0};
  for (int b = 0; b < blocks; b++)
    for (int k = 0; k < 4; k++) s[k] += h[(size_t)b * 4 + k];
  out->sse = s[0]; out->n = (int64_t)s[1]; out->unorm2 = s[2]; out->inorm2 = s[3];
  return MFX_OK;
}
