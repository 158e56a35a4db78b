// sgd_ifw.hip -- inverse-frequency-weighted MF (ModelInvPopMF, --algo=IFWMF): the SGD visit and the objective with
// a per-rating weight on the error term (modelInvPopMF.cpp:152-178 and :3-55).
//
//   wt = invPopI[item];  if (itemFreq[item] > userFreq[u]) wt = invPopU[u];   wt = 1.0/(1.0 + rhoRMS*wt)  (float)
//   p -= lr*(-2.0*wt*diff*q + 2.0*uReg*p);   q -= lr*(-2.0*wt*diff*p + 2.0*iReg*q)     (double bracket, diff double)
//   objective = sum wt*diff*diff + uReg*sum ||p||^2 + iReg*sum ||q||^2
//
// The weight is a function of one user attribute pair and one item attribute pair, so the kernels gather
// (freq, invPop) of the user and of the item next to the factor rows instead of carrying a weight per rating
// through the shuffled list.  Same L-lane group per rating and dot order as sgd.hip.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "mfx_internal.h"

#include "sgd_common.h"
#include "sgd_variants.h"

namespace {
struct IfwState {
  float2 *ua = nullptr, *ia = nullptr;   // (freq, invPop) per user / per item
  float rho = 0.0f;
};
IfwState* ifw(mfx_ctx* ctx) { return (IfwState*)ctx->ifw; }

__device__ __forceinline__ double wave_sum_dd(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
}  // namespace

void mfx_ifw_tables(mfx_ctx* ctx, const float2** ua, const float2** ia, float* rho) {
  IfwState* s = ifw(ctx);
  *ua = s ? s->ua : nullptr; *ia = s ? s->ia : nullptr; *rho = s ? s->rho : 0.0f;
}

void mfx_ifw_free_internal(mfx_ctx* ctx) {
  ctx->var_gen++;
  IfwState* s = ifw(ctx);
  if (!s) return;
  dev_free(s->ua); dev_free(s->ia);
  delete s;
  ctx->ifw = nullptr;
}

template <int L, int C, bool SERIAL>
__global__ __launch_bounds__(256) void sgd_ifw_kernel(const int32_t* __restrict__ eu, const int32_t* __restrict__ ei,
                                                      const float* __restrict__ er, int64_t first, int64_t count, float* U, float* V,
                                                      uint32_t ubytes, uint32_t vbytes, float lr, float uReg, float iReg,
                                                      const float2* __restrict__ ua, const float2* __restrict__ ia, float rho) {
  constexpr int G = 64 / L;
  constexpr int LD = 4 * L * C;
  const int lane = threadIdx.x & 63, g = lane / L, j = lane % L;
  if (SERIAL) {
    if (blockIdx.x != 0 || threadIdx.x >= L) return;
    const Rows<0> Um(U, 0), Vm(V, 0);
    for (int64_t t = 0; t < count; t++) {
      const int u = eu[first + t], it = ei[first + t];
      visit_ifw<L, C, 0>(Um, Vm, (int64_t)u * LD + 4 * j, (int64_t)it * LD + 4 * j, er[first + t], mfx_ifw_weight(ua[u], ia[it], rho), lr,
                         uReg, iReg);
    }
    return;
  }
  const Rows<1> Um(U, ubytes), Vm(V, vbytes);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t base = wave * 64; base < count; base += nwaves * 64) {
    const int nvalid = (int)(count - base < 64 ? count - base : 64);
    const bool ok = lane < nvalid;
    const int mu = ok ? eu[first + base + lane] : 0;
    const int mi = ok ? ei[first + base + lane] : 0;
    const float mr = ok ? er[first + base + lane] : 0.0f;
    const float mw = ok ? mfx_ifw_weight(ua[mu], ia[mi], rho) : 0.0f;      // one weight per lane, then broadcast like (u, i, r)
#pragma unroll 1
    for (int s = 0; s < L; s++) {
      const int e = s * G + g;
      const int u = __shfl(mu, e, 64);
      const int it = __shfl(mi, e, 64);
      const float r = __shfl(mr, e, 64);
      const float wt = __shfl(mw, e, 64);
      if (e < nvalid) visit_ifw<L, C, 1>(Um, Vm, (int64_t)u * LD + 4 * j, (int64_t)it * LD + 4 * j, r, wt, lr, uReg, iReg);
    }
  }
}

template <int L, int C>
static int launch_ifw(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count) {
  ProfScope ps(ctx, MFX_K_SGD);
  IfwState* s = ifw(ctx);
  const uint64_t ub = (uint64_t)ctx->nU * ctx->ld * 4, vb = (uint64_t)ctx->nI * ctx->ld * 4;
  NEED(ub < (1ull << 32) && vb < (1ull << 32), MFX_E_ARG, "weighted sgd: factor matrices must be < 4 GiB");
  if (o->mode == MFX_SGD_SERIAL) {
    hipLaunchKernelGGL((sgd_ifw_kernel<L, C, true>), dim3(1), dim3(64), 0, ctx->stream, ctx->eu, ctx->ei, ctx->er, first, count, ctx->U,
                       ctx->V, (uint32_t)ub, (uint32_t)vb, o->learnRate, o->uReg, o->iReg, s->ua, s->ia, s->rho);
  } else {
    const int64_t waves = (count + 63) / 64;
    const int cap = o->blocks > 0 ? std::min(o->blocks, 8192) : std::max(8, std::min(2048, std::min(ctx->nU, ctx->nI) / 64));
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((waves + 3) / 4, cap));
    hipLaunchKernelGGL((sgd_ifw_kernel<L, C, false>), dim3(blocks), dim3(256), 0, ctx->stream, ctx->eu, ctx->ei, ctx->er, first, count,
                       ctx->U, ctx->V, (uint32_t)ub, (uint32_t)vb, o->learnRate, o->uReg, o->iReg, s->ua, s->ia, s->rho);
  }
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

int mfx_launch_sgd_ifw(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count) {
  NEED(o->mode == MFX_SGD_HOGWILD || o->mode == MFX_SGD_SERIAL, MFX_E_ARG,
       "rating weights (mfx_sgd_set_ifw): this launcher serves MFX_SGD_HOGWILD / MFX_SGD_SERIAL (mode=%d)", o->mode);
  const int L = ctx->L, C = ctx->C;
  if (L == 4) return launch_ifw<4, 1>(ctx, o, first, count);
  if (L == 8) return launch_ifw<8, 1>(ctx, o, first, count);
  switch (C) {
    case 1: return launch_ifw<16, 1>(ctx, o, first, count);
    case 2: return launch_ifw<16, 2>(ctx, o, first, count);
    case 3: return launch_ifw<16, 3>(ctx, o, first, count);
    case 4: return launch_ifw<16, 4>(ctx, o, first, count);
  }
  return mfx_fail(ctx, MFX_E_ARG, "weighted sgd: K <= 256");
}

extern "C" int mfx_sgd_set_ifw(mfx_ctx* ctx, const float* userFreq, const float* invPopU, const float* itemFreq, const float* invPopI,
                               float rhoRMS) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->U, MFX_E_STATE, "mfx_sgd_set_ifw: no model");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  mfx_ifw_free_internal(ctx);
  if (!userFreq && !invPopU && !itemFreq && !invPopI) return MFX_OK;
  NEED(userFreq && invPopU && itemFreq && invPopI, MFX_E_ARG, "mfx_sgd_set_ifw: all four arrays or none");
  IfwState* s = new IfwState;
  ctx->ifw = s;
  s->rho = rhoRMS;
  std::vector<float2> hu((size_t)ctx->nU), hi((size_t)ctx->nI);
  for (int u = 0; u < ctx->nU; u++) hu[u] = make_float2(userFreq[u], invPopU[u]);
  for (int i = 0; i < ctx->nI; i++) hi[i] = make_float2(itemFreq[i], invPopI[i]);
  int rc;
  if ((rc = dev_alloc(ctx, &s->ua, hu.size())) || (rc = dev_alloc(ctx, &s->ia, hi.size()))) return rc;
  HIPCHK(hipMemcpy(s->ua, hu.data(), sizeof(float2) * hu.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->ia, hi.data(), sizeof(float2) * hi.size(), hipMemcpyHostToDevice));
  return MFX_OK;
}

// sum over the valid train ratings of wt*diff*diff
template <int L, int C>
__global__ __launch_bounds__(256) void ifw_sse_kernel(const int32_t* __restrict__ ru, const int32_t* __restrict__ ri,
                                                      const float* __restrict__ rr, int64_t n, const float* __restrict__ U,
                                                      const float* __restrict__ V, const uint8_t* __restrict__ invU,
                                                      const uint8_t* __restrict__ invI, const float2* __restrict__ ua,
                                                      const float2* __restrict__ ia, float rho, double* __restrict__ part) {
  constexpr int G = 64 / L;
  constexpr int LD = 4 * L * C;
  __shared__ double sh[4];
  const int lane = threadIdx.x & 63, g = lane / L, j = lane % L;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double acc = 0.0;
  for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
    const bool ok = base + lane < n;
    const int mu = ok ? ru[base + lane] : 0;
    const int mi = ok ? ri[base + lane] : 0;
    const float mr = ok ? rr[base + lane] : 0.0f;
    const int nvalid = (int)(n - base < 64 ? n - base : 64);
#pragma unroll 1
    for (int s = 0; s < L; s++) {
      const int e = s * G + g;
      const int u = __shfl(mu, e, 64);
      const int it = __shfl(mi, e, 64);
      const float r = __shfl(mr, e, 64);
      if (e < nvalid && !invU[u] && !invI[it]) {
        float4v p[C], q[C];
#pragma unroll
        for (int c = 0; c < C; c++) {
          p[c] = *(const float4v*)(U + (int64_t)u * LD + 4 * j + c * 4 * L);
          q[c] = *(const float4v*)(V + (int64_t)it * LD + 4 * j + c * 4 * L);
        }
        const float est = group_dot<L, C>(p, q);
        const double diff = (double)r - (double)est;
        if (j == 0) acc += ((double)mfx_ifw_weight(ua[u], ia[it], rho) * diff) * diff;     // rmse += wt*diff*diff
      }
    }
  }
  const double wd = wave_sum_dd(acc);
  if (lane == 0) sh[threadIdx.x >> 6] = wd;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

extern "C" int mfx_eval_ifw(mfx_ctx* ctx, int snapshot, mfx_eval_out* out) {
  if (!ctx) return MFX_E_ARG;
  IfwState* s = ifw(ctx);
  NEED(s, MFX_E_STATE, "mfx_eval_ifw: call mfx_sgd_set_ifw first");
  int rc = mfx_eval(ctx, MFX_MAT_TRAIN, snapshot, 1, out);      // count and the two norms
  if (rc) return rc;
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(((m.nnz + 63) / 64 + 3) / 4, 1024));
  double* part = nullptr;
  if ((rc = dev_alloc(ctx, &part, (size_t)nb))) return rc;
  const float* U = snapshot ? ctx->Ubest : ctx->U;
  const float* V = snapshot ? ctx->Vbest : ctx->V;
#define MFX_IFW_SSE(LL, CC)                                                                                                    \
  hipLaunchKernelGGL((ifw_sse_kernel<LL, CC>), dim3(nb), dim3(256), 0, ctx->stream, m.rowid, m.rowind, m.rowval, m.nnz, U, V, \
                     ctx->invU, ctx->invI, s->ua, s->ia, s->rho, part)
  const int L = ctx->L, C = ctx->C;
  if (L == 4) MFX_IFW_SSE(4, 1); else if (L == 8) MFX_IFW_SSE(8, 1);
  else if (C == 1) MFX_IFW_SSE(16, 1); else if (C == 2) MFX_IFW_SSE(16, 2); else if (C == 3) MFX_IFW_SSE(16, 3);
  else if (C == 4) MFX_IFW_SSE(16, 4);
  else { dev_free(part); return mfx_fail(ctx, MFX_E_ARG, "mfx_eval_ifw: K <= 256"); }
#undef MFX_IFW_SSE
  std::vector<double> h((size_t)nb);
  hipError_t e = hipMemcpyAsync(h.data(), part, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  dev_free(part);
  NEED(e == hipSuccess, MFX_E_HIP, "mfx_eval_ifw: %s", hipGetErrorString(e));
  double sse = 0;
  for (double v : h) sse += v;
  out->sse = sse;
  return MFX_OK;
}
