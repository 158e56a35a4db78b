// eval.hip -- objective / RMSE pass.
//
// Replaces Model::objective (model.cpp:1770-1815) and Model::RMSE (model.cpp:214-251),
// which the reference runs after EVERY iteration (OBJ_ITER = 1, const.h:4): one fp32
// dot per rating, (r - est)^2 accumulated in double.  Same L-lane group per rating and
// the same dot order as the SGD kernel; partial sums are combined in a fixed order
// (per-lane -> wave -> block -> one finishing block), so the result is reproducible
// run to run (the reference's OpenMP reduction order is thread-dependent).
#include <algorithm>

#include "sgd_common.h"

template <int L, int C>
__device__ __forceinline__ float row_dot(const float* __restrict__ a, const float* __restrict__ b) {
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < C; c++) {
    const float4v x = *(const float4v*)(a + c * 4 * L);
    const float4v y = *(const float4v*)(b + c * 4 * L);
    s = __builtin_fmaf(x.x, y.x, s);
    s = __builtin_fmaf(x.y, y.y, s);
    s = __builtin_fmaf(x.z, y.z, s);
    s = __builtin_fmaf(x.w, y.w, s);
  }
  return group_sum<L>(s);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// block-level fixed-order combine; result valid in thread 0
template <typename T>
__device__ __forceinline__ T block_combine(T wave_val, T* sh) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sh[w] = wave_val;
  __syncthreads();
  T tot = 0;
  if (threadIdx.x == 0)
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) tot += sh[k];
  __syncthreads();
  return tot;
}

// truncated-rank estimate (ModelDropoutSigmoid::estRating, modelDropoutSigmoid.cpp:5-24): elements k < rank only
template <int L, int C>
__device__ __forceinline__ float row_dot_trunc(const float* __restrict__ a, const float* __restrict__ b, int j, int rank) {
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < C; c++) {
    const float4v x = *(const float4v*)(a + c * 4 * L);
    const float4v y = *(const float4v*)(b + c * 4 * L);
#pragma unroll
    for (int e = 0; e < 4; e++)
      if (c * 4 * L + 4 * j + e < rank) s = __builtin_fmaf(x[e], y[e], s);
  }
  return group_sum<L>(s);
}

// MASKS = false when no user and no item is invalid (the usual case): the two mask bytes per rating are not fetched.
// Four ratings per group are in flight at a time: their indices are clamped instead of branched on, so the row loads
// of all four are issued before the first dot product (the loop was a chain of dependent L2 round trips).  The sums
// are accumulated per lane in the same rating order as before.
template <int L, int C, bool TMF, bool MASKS>
__global__ __launch_bounds__(256) void eval_sse_kernel(const int32_t* __restrict__ ru,
                                                       const int32_t* __restrict__ ri,
                                                       const float* __restrict__ rr, int64_t n,
                                                       const float* __restrict__ U, const float* __restrict__ V,
                                                       const uint8_t* __restrict__ invU,
                                                       const uint8_t* __restrict__ invI, int32_t nU, int32_t nI,
                                                       double* __restrict__ part_d, int64_t* __restrict__ part_i,
                                                       const int2* __restrict__ tu, const int2* __restrict__ ti) {
  constexpr int G = 64 / L;
  constexpr int LD = 4 * L * C;
  constexpr int UN = 4;                      // L is 4, 8 or 16
  __shared__ double shd[4];
  __shared__ long long shi[4];
  const int lane = threadIdx.x & 63;
  const int g = lane / L, j = lane % L;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double acc = 0.0;
  long long cnt = 0;
  for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
    const bool ok = base + lane < n;
    const int mu = ok ? ru[base + lane] : 0;
    const int mi = ok ? ri[base + lane] : 0;
    const float mr = ok ? rr[base + lane] : 0.0f;
    const int nvalid = (int)(n - base < 64 ? n - base : 64);
#pragma unroll 1
    for (int s0 = 0; s0 < L; s0 += UN) {
      int u[UN], it[UN];
      float r[UN], est[UN];
      bool use[UN];
#pragma unroll
      for (int x = 0; x < UN; x++) {
        const int e = (s0 + x) * G + g;
        u[x] = __shfl(mu, e, 64);
        it[x] = __shfl(mi, e, 64);
        r[x] = __shfl(mr, e, 64);
        // Model::RMSE: u < nUsers, user valid, item < nItems and valid (model.cpp:223-237)
        use[x] = e < nvalid && u[x] < nU && it[x] < nI;
        if (!use[x]) { u[x] = 0; it[x] = 0; }
      }
      if (MASKS) {
#pragma unroll
        for (int x = 0; x < UN; x++) use[x] = use[x] && !invU[u[x]] && !invI[it[x]];
      }
#pragma unroll
      for (int x = 0; x < UN; x++) {
        if (TMF) {
          const int2 a = tu[u[x]], b = ti[it[x]];
          est[x] = row_dot_trunc<L, C>(U + (int64_t)u[x] * LD + 4 * j, V + (int64_t)it[x] * LD + 4 * j, j, mfx_tmf_rank(a, b));
        } else {
          est[x] = row_dot<L, C>(U + (int64_t)u[x] * LD + 4 * j, V + (int64_t)it[x] * LD + 4 * j);
        }
      }
#pragma unroll
      for (int x = 0; x < UN; x++) {
        const double diff = (double)r[x] - (double)est[x];
        if (use[x] && j == 0) { acc += diff * diff; cnt++; }
      }
    }
  }
  const double wd = wave_sum_d(acc);
  const long long wi = wave_sum_ll(cnt);
  const double bd = block_combine<double>(wd, shd);
  const long long bi = block_combine<long long>(wi, shi);
  if (threadIdx.x == 0) { part_d[blockIdx.x] = bd; part_i[blockIdx.x] = bi; }
}

// sum over valid rows of ||x||^2 (fp32 dot in device order, double sum)
template <int L, int C>
__global__ __launch_bounds__(256) void eval_norm_kernel(const float* __restrict__ X, int32_t n,
                                                        const uint8_t* __restrict__ inv,
                                                        double* __restrict__ part_d) {
  constexpr int G = 64 / L;
  constexpr int LD = 4 * L * C;
  __shared__ double shd[4];
  const int lane = threadIdx.x & 63;
  const int g = lane / L, j = lane % L;
  const int64_t grp = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * G + g;
  const int64_t ngrp = (((int64_t)gridDim.x * blockDim.x) >> 6) * G;
  double acc = 0.0;
  for (int64_t r = grp; r < n; r += ngrp) {
    if (inv[r]) continue;
    const float* x = X + r * LD + 4 * j;
    const float d = row_dot<L, C>(x, x);
    if (j == 0) acc += (double)d;
  }
  const double bd = block_combine<double>(wave_sum_d(acc), shd);
  if (threadIdx.x == 0) part_d[blockIdx.x] = bd;
}

// one 256-thread block; fixed association: thread t sums entries t, t+256, ... then a
// fixed LDS tree -- reproducible for a given grid
__device__ __forceinline__ double finish_sum(const double* __restrict__ p, int n, double* sh) {
  double a = 0;
  for (int k = threadIdx.x; k < n; k += 256) a += p[k];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}
__global__ __launch_bounds__(256) void eval_finish_kernel(const double* __restrict__ p0, int n0,
                                                          const int64_t* __restrict__ pi,
                                                          const double* __restrict__ p1, int n1,
                                                          const double* __restrict__ p2, int n2,
                                                          double* __restrict__ out) {
  __shared__ double sh[256];
  __shared__ long long shi[256];
  const double s = finish_sum(p0, n0, sh);
  const double a = finish_sum(p1, n1, sh);
  const double b = finish_sum(p2, n2, sh);
  long long c = 0;
  for (int k = threadIdx.x; k < n0; k += 256) c += pi[k];
  shi[threadIdx.x] = c;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) shi[threadIdx.x] += shi[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = s;
    out[1] = (double)shi[0];
    out[2] = a;
    out[3] = b;
  }
}

template <int L, int C>
static int eval_lc(mfx_ctx* ctx, const DevCSR& m, const float* U, const float* V, int with_norms, int nb,
                   int nbu, int nbi) {
  ProfScope ps(ctx, MFX_K_EVAL);
  double* pd = ctx->red_d;
  // the mask bytes are needed when something is invalid, or while mfx_eval_filtered has its own masks in place
  const bool masks = ctx->n_invalid != 0 || ctx->force_masks;
#define MFX_EVAL_SSE(TMFV, MASKV)                                                                                              \
  hipLaunchKernelGGL((eval_sse_kernel<L, C, TMFV, MASKV>), dim3(nb), dim3(256), 0, ctx->stream, m.rowid, m.rowind, m.rowval, m.nnz, \
                     U, V, ctx->invU, ctx->invI, ctx->nU, ctx->nI, pd, ctx->red_i, ctx->tmf_u, ctx->tmf_i)
  if (ctx->tmf_u) {
    if (masks) MFX_EVAL_SSE(true, true); else MFX_EVAL_SSE(true, false);
  } else {
    if (masks) MFX_EVAL_SSE(false, true); else MFX_EVAL_SSE(false, false);
  }
#undef MFX_EVAL_SSE
  if (with_norms) {
    hipLaunchKernelGGL((eval_norm_kernel<L, C>), dim3(nbu), dim3(256), 0, ctx->stream, U, ctx->nU, ctx->invU,
                       pd + nb);
    hipLaunchKernelGGL((eval_norm_kernel<L, C>), dim3(nbi), dim3(256), 0, ctx->stream, V, ctx->nI, ctx->invI,
                       pd + nb + nbu);
  }
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

// block counts of one evaluation
static void eval_blocks(const mfx_ctx* ctx, const DevCSR& m, int* nb, int* nbu, int* nbi) {
  const int G = 64 / ctx->L;
  *nb = (int)std::max<int64_t>(1, std::min<int64_t>(((m.nnz + 63) / 64 + 3) / 4, 2048));
  *nbu = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ctx->nU + 4 * G - 1) / (4 * G), 1024));
  *nbi = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ctx->nI + 4 * G - 1) / (4 * G), 1024));
}

// launches the kernels of one evaluation into the partial-sum region starting at `base` and its 4 results into
// dout[0..3]; nothing is copied back yet
static int eval_enqueue(mfx_ctx* ctx, const DevCSR& m, const float* U, const float* V, int with_norms, int base, double* dout) {
  const int L = ctx->L, C = ctx->C;
  int nb, nbu, nbi;
  eval_blocks(ctx, m, &nb, &nbu, &nbi);
  double* save_d = ctx->red_d;
  int64_t* save_i = ctx->red_i;
  ctx->red_d += base;          // eval_lc addresses the region through the context
  ctx->red_i += base;
  int rc = MFX_E_ARG;
  if (L == 4) rc = eval_lc<4, 1>(ctx, m, U, V, with_norms, nb, nbu, nbi);
  else if (L == 8) rc = eval_lc<8, 1>(ctx, m, U, V, with_norms, nb, nbu, nbi);
  else switch (C) {
    case 1: rc = eval_lc<16, 1>(ctx, m, U, V, with_norms, nb, nbu, nbi); break;
    case 2: rc = eval_lc<16, 2>(ctx, m, U, V, with_norms, nb, nbu, nbi); break;
    case 3: rc = eval_lc<16, 3>(ctx, m, U, V, with_norms, nb, nbu, nbi); break;
    case 4: rc = eval_lc<16, 4>(ctx, m, U, V, with_norms, nb, nbu, nbi); break;
    case 5: rc = eval_lc<16, 5>(ctx, m, U, V, with_norms, nb, nbu, nbi); break;
    case 6: rc = eval_lc<16, 6>(ctx, m, U, V, with_norms, nb, nbu, nbi); break;
    case 7: rc = eval_lc<16, 7>(ctx, m, U, V, with_norms, nb, nbu, nbi); break;
    case 8: rc = eval_lc<16, 8>(ctx, m, U, V, with_norms, nb, nbu, nbi); break;
  }
  if (!rc) {
    hipLaunchKernelGGL(eval_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->red_d, nb, ctx->red_i, ctx->red_d + nb,
                       with_norms ? nbu : 0, ctx->red_d + nb + nbu, with_norms ? nbi : 0, dout);
    if (hipGetLastError() != hipSuccess) rc = mfx_fail(ctx, MFX_E_HIP, "eval: launch failed");
  }
  ctx->red_d = save_d;
  ctx->red_i = save_i;
  return rc;
}

// room for `count` evaluations side by side (+ 8 result doubles)
static int eval_reserve(mfx_ctx* ctx, int need_total) {
  if (ctx->red_blocks >= need_total) return MFX_OK;
  dev_free(ctx->red_d);
  dev_free(ctx->red_i);
  ctx->red_blocks = 0;
  int rc;
  if ((rc = dev_alloc(ctx, &ctx->red_d, (size_t)need_total + 8))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->red_i, (size_t)need_total))) return rc;
  ctx->red_blocks = need_total;
  return MFX_OK;
}
static void eval_unpack(const double* r, mfx_eval_out* out) {
  out->sse = r[0]; out->n = (int64_t)r[1]; out->unorm2 = r[2]; out->inorm2 = r[3];
}

int mfx_launch_eval(mfx_ctx* ctx, const DevCSR& m, const float* U, const float* V, int with_norms, mfx_eval_out* out) {
  int nb, nbu, nbi;
  eval_blocks(ctx, m, &nb, &nbu, &nbi);
  int rc = eval_reserve(ctx, nb + nbu + nbi);
  if (rc) return rc;
  double* dout = ctx->red_d + ctx->red_blocks;
  if ((rc = eval_enqueue(ctx, m, U, V, with_norms, 0, dout))) return rc;
  HIPCHK(hipMemcpyAsync(ctx->red_out, dout, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  eval_unpack(ctx->red_out, out);
  if (int rc = mfx_slots_check_abort(ctx)) return rc;      // an evaluation of factors an aborted drain left behind is an error
  return MFX_OK;
}

// two evaluations, one copy back, one synchronisation (the objective + validation RMSE of every iteration)
int mfx_launch_eval2(mfx_ctx* ctx, const DevCSR& ma, int norms_a, const DevCSR& mb, int norms_b, const float* U, const float* V,
                     mfx_eval_out* out_a, mfx_eval_out* out_b) {
  int nb, nbu, nbi, nb2, nbu2, nbi2;
  eval_blocks(ctx, ma, &nb, &nbu, &nbi);
  eval_blocks(ctx, mb, &nb2, &nbu2, &nbi2);
  const int need_a = nb + nbu + nbi;
  int rc = eval_reserve(ctx, need_a + nb2 + nbu2 + nbi2);
  if (rc) return rc;
  double* dout = ctx->red_d + ctx->red_blocks;
  if ((rc = eval_enqueue(ctx, ma, U, V, norms_a, 0, dout))) return rc;
  if ((rc = eval_enqueue(ctx, mb, U, V, norms_b, need_a, dout + 4))) return rc;
  HIPCHK(hipMemcpyAsync(ctx->red_out, dout, 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  eval_unpack(ctx->red_out, out_a);
  eval_unpack(ctx->red_out + 4, out_b);
  if (int rc = mfx_slots_check_abort(ctx)) return rc;
  return MFX_OK;
}
