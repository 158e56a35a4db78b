// sgd_tmf.hip -- truncated-rank MF (ModelDropoutSigmoid, --algo=TMF): every rating works on the first
// updMinRank <= K dimensions only, updMinRank = ceil(sigmoid(rhoRMS*(z - alpha)) * K) with z the z-score of the
// SMALLER of the user's and the item's train frequency (modelDropoutSigmoid.cpp:5-24 estRating, :158-188 update).
//
// The rank is a function of one frequency, so the host evaluates it once per user and once per item with the
// reference's double arithmetic (exp, ceil) and the kernels pick `userFreq < itemFreq ? rank_u : rank_i`; element k of
// a row takes part in the dot and in the two axpys iff k < rank.  With the table set (mfx_set_tmf) the evaluation
// kernels (eval.hip) use the same truncated estimate, which is what the class's estRating override does to
// Model::RMSE / objective.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mfx_internal.h"

#include "sgd_common.h"
#include "sgd_variants.h"

template <int L, int C, bool SERIAL>
__global__ __launch_bounds__(256) void sgd_tmf_kernel(const int32_t* __restrict__ eu, const int32_t* __restrict__ ei,
                                                      const float* __restrict__ er, int64_t first, int64_t count, float* U, float* V,
                                                      uint32_t ubytes, uint32_t vbytes, float lr, float uReg, float iReg,
                                                      const int2* __restrict__ tu, const int2* __restrict__ ti,
                                                      const int32_t* __restrict__ du, const int32_t* __restrict__ di,
                                                      const double* __restrict__ dexp, uint32_t seed, uint32_t epoch, int K) {
  constexpr int G = 64 / L;
  constexpr int LD = 4 * L * C;
  const int lane = threadIdx.x & 63, g = lane / L, j = lane % L;
  if (SERIAL) {
    if (blockIdx.x != 0 || threadIdx.x >= L) return;
    const Rows<0> Um(U, 0), Vm(V, 0);
    for (int64_t t = 0; t < count; t++) {
      const int u = eu[first + t], it = ei[first + t];
      const int2 a = tu[u], b = ti[it];
      int rank = mfx_tmf_rank(a, b);
      if (du) {                               // ModelPoissonDropout: Poisson(lambda of the rarer side)
        const int lam = __int_as_float(a.x) < __int_as_float(b.x) ? du[u] : di[it];
        rank = mfx_poisson_rank(lam, dexp[lam], mfx_draw_hash(seed, epoch, (uint32_t)u, (uint32_t)it), K);
      }
      visit_tmf<L, C, 0>(Um, Vm, (int64_t)u * LD + 4 * j, (int64_t)it * LD + 4 * j, er[first + t], rank, j, lr, uReg, iReg);
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
    int mk = 0;
    if (ok) {
      const int2 a = tu[mu], b = ti[mi];
      mk = mfx_tmf_rank(a, b);
      if (du) {
        const int lam = __int_as_float(a.x) < __int_as_float(b.x) ? du[mu] : di[mi];
        mk = mfx_poisson_rank(lam, dexp[lam], mfx_draw_hash(seed, epoch, (uint32_t)mu, (uint32_t)mi), K);
      }
    }
#pragma unroll 1
    for (int s = 0; s < L; s++) {
      const int e = s * G + g;
      const int u = __shfl(mu, e, 64);
      const int it = __shfl(mi, e, 64);
      const float r = __shfl(mr, e, 64);
      const int rank = __shfl(mk, e, 64);
      if (e < nvalid) visit_tmf<L, C, 1>(Um, Vm, (int64_t)u * LD + 4 * j, (int64_t)it * LD + 4 * j, r, rank, j, lr, uReg, iReg);
    }
  }
}

template <int L, int C>
static int launch_tmf(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count) {
  ProfScope ps(ctx, MFX_K_SGD);
  const uint64_t ub = (uint64_t)ctx->nU * ctx->ld * 4, vb = (uint64_t)ctx->nI * ctx->ld * 4;
  NEED(ub < (1ull << 32) && vb < (1ull << 32), MFX_E_ARG, "truncated-rank sgd: factor matrices must be < 4 GiB");
  if (o->mode == MFX_SGD_SERIAL) {
    hipLaunchKernelGGL((sgd_tmf_kernel<L, C, true>), dim3(1), dim3(64), 0, ctx->stream, ctx->eu, ctx->ei, ctx->er, first, count, ctx->U,
                       ctx->V, (uint32_t)ub, (uint32_t)vb, o->learnRate, o->uReg, o->iReg, ctx->tmf_u, ctx->tmf_i, ctx->tmfd_u, ctx->tmfd_i,
                       ctx->tmfd_exp, ctx->tmfd_seed, (uint32_t)o->epoch, ctx->K);
  } else {
    const int64_t waves = (count + 63) / 64;
    const int cap = o->blocks > 0 ? std::min(o->blocks, 8192) : std::max(8, std::min(2048, std::min(ctx->nU, ctx->nI) / 64));
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((waves + 3) / 4, cap));
    hipLaunchKernelGGL((sgd_tmf_kernel<L, C, false>), dim3(blocks), dim3(256), 0, ctx->stream, ctx->eu, ctx->ei, ctx->er, first, count,
                       ctx->U, ctx->V, (uint32_t)ub, (uint32_t)vb, o->learnRate, o->uReg, o->iReg, ctx->tmf_u, ctx->tmf_i,
                       ctx->tmfd_u, ctx->tmfd_i, ctx->tmfd_exp, ctx->tmfd_seed, (uint32_t)o->epoch, ctx->K);
  }
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

int mfx_launch_sgd_tmf(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count) {
  NEED(o->mode == MFX_SGD_HOGWILD || o->mode == MFX_SGD_SERIAL, MFX_E_ARG,
       "truncated ranks (mfx_set_tmf) run on MFX_SGD_HOGWILD / MFX_SGD_SERIAL (mode=%d)", o->mode);
  const int L = ctx->L, C = ctx->C;
  if (L == 4) return launch_tmf<4, 1>(ctx, o, first, count);
  if (L == 8) return launch_tmf<8, 1>(ctx, o, first, count);
  switch (C) {
    case 1: return launch_tmf<16, 1>(ctx, o, first, count);
    case 2: return launch_tmf<16, 2>(ctx, o, first, count);
    case 3: return launch_tmf<16, 3>(ctx, o, first, count);
    case 4: return launch_tmf<16, 4>(ctx, o, first, count);
  }
  return mfx_fail(ctx, MFX_E_ARG, "truncated-rank sgd: K <= 256");
}

void mfx_tmf_free_internal(mfx_ctx* ctx) {
  ctx->var_gen++;
  dev_free(ctx->tmf_u); dev_free(ctx->tmf_i);
  dev_free(ctx->tmfd_u); dev_free(ctx->tmfd_i); dev_free(ctx->tmfd_exp);
}

extern "C" int mfx_set_tmf_dropout(mfx_ctx* ctx, const int32_t* userLambda, const int32_t* itemLambda, uint32_t seed) {
  if (!ctx) return MFX_E_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->var_gen++;
  dev_free(ctx->tmfd_u); dev_free(ctx->tmfd_i); dev_free(ctx->tmfd_exp);
  if (!userLambda && !itemLambda) return MFX_OK;
  NEED(ctx->tmf_u, MFX_E_STATE, "mfx_set_tmf_dropout: set the evaluation ranks first (mfx_set_tmf)");
  NEED(userLambda && itemLambda, MFX_E_ARG, "mfx_set_tmf_dropout: both arrays or none");
  for (int u = 0; u < ctx->nU; u++)
    NEED(userLambda[u] >= 1 && userLambda[u] <= ctx->K, MFX_E_ARG, "mfx_set_tmf_dropout: userLambda[%d]=%d outside [1,%d]", u, userLambda[u], ctx->K);
  for (int i = 0; i < ctx->nI; i++)
    NEED(itemLambda[i] >= 1 && itemLambda[i] <= ctx->K, MFX_E_ARG, "mfx_set_tmf_dropout: itemLambda[%d]=%d outside [1,%d]", i, itemLambda[i], ctx->K);
  std::vector<double> e((size_t)ctx->K + 1);
  for (int l = 0; l <= ctx->K; l++) e[l] = exp(-(double)l);
  int rc;
  if ((rc = dev_alloc(ctx, &ctx->tmfd_u, (size_t)ctx->nU)) || (rc = dev_alloc(ctx, &ctx->tmfd_i, (size_t)ctx->nI)) ||
      (rc = dev_alloc(ctx, &ctx->tmfd_exp, e.size())))
    return rc;
  HIPCHK(hipMemcpy(ctx->tmfd_u, userLambda, sizeof(int32_t) * (size_t)ctx->nU, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->tmfd_i, itemLambda, sizeof(int32_t) * (size_t)ctx->nI, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->tmfd_exp, e.data(), sizeof(double) * e.size(), hipMemcpyHostToDevice));
  ctx->tmfd_seed = seed;
  return MFX_OK;
}

extern "C" int mfx_set_tmf(mfx_ctx* ctx, const float* userFreq, const int32_t* userRank, const float* itemFreq,
                           const int32_t* itemRank) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->U, MFX_E_STATE, "mfx_set_tmf: no model");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  mfx_tmf_free_internal(ctx);
  if (!userFreq && !userRank && !itemFreq && !itemRank) return MFX_OK;
  NEED(userFreq && userRank && itemFreq && itemRank, MFX_E_ARG, "mfx_set_tmf: all four arrays or none");
  std::vector<int2> hu((size_t)ctx->nU), hi((size_t)ctx->nI);
  for (int u = 0; u < ctx->nU; u++) {
    NEED(userRank[u] >= 1 && userRank[u] <= ctx->K, MFX_E_ARG, "mfx_set_tmf: userRank[%d]=%d outside [1,%d]", u, userRank[u], ctx->K);
    int fb; memcpy(&fb, &userFreq[u], 4);
    hu[u] = make_int2(fb, userRank[u]);
  }
  for (int i = 0; i < ctx->nI; i++) {
    NEED(itemRank[i] >= 1 && itemRank[i] <= ctx->K, MFX_E_ARG, "mfx_set_tmf: itemRank[%d]=%d outside [1,%d]", i, itemRank[i], ctx->K);
    int fb; memcpy(&fb, &itemFreq[i], 4);
    hi[i] = make_int2(fb, itemRank[i]);
  }
  int rc;
  if ((rc = dev_alloc(ctx, &ctx->tmf_u, hu.size())) || (rc = dev_alloc(ctx, &ctx->tmf_i, hi.size()))) return rc;
  HIPCHK(hipMemcpy(ctx->tmf_u, hu.data(), sizeof(int2) * hu.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->tmf_i, hi.data(), sizeof(int2) * hi.size(), hipMemcpyHostToDevice));
  return MFX_OK;
}
