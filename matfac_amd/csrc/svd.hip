// svd.hip -- what ModelMF::trainSGDParSVD (modelMF.cpp:353-557) needs beyond the shared SGD pieces:
//   * the rank-K truncated SVD of the train matrix that initialises the factors (the reference calls SVDLIBC's
//     svdLAS2A through svdFrmSvdlibCSREig, svdFrmsvdlib.cpp:69-133: uFac <- left, iFac <- right singular
//     vectors, singular values kept).  SVDLIBC is not in the tree; this is a randomized block subspace iteration
//     (range finder with CholeskyQR2 re-orthonormalisation, then Rayleigh-Ritz on the (K+p)-dimensional
//     subspace): sparse x tall-skinny products on the device, the small dense factorizations in double on the
//     host.  Singular vectors are defined up to sign (and rotation inside clusters of equal singular values);
//     the product U S V^T, which is all the SGD that follows sees, is not.
//   * the SGD update with a per-dimension regulariser 2*((sing_a+1)/(sing_b+sigma_k)) (modelMF.cpp:498-505)
//   * the norms weighted by sigma_k for Model::objectiveSing (model.cpp:1818-1865)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "mfx_internal.h"

#include "sgd_common.h"
#include "sgd_variants.h"

// ---------------------------------------------------------------------------
// SGD with per-dimension regularisation
// ---------------------------------------------------------------------------
template <int L, int C, bool SERIAL>
__global__ __launch_bounds__(256) void sgd_dimreg_kernel(const int32_t* __restrict__ eu, const int32_t* __restrict__ ei,
                                                         const float* __restrict__ er, int64_t first, int64_t count, float* U,
                                                         float* V, uint32_t ubytes, uint32_t vbytes, float lr,
                                                         const float* __restrict__ regk) {
  constexpr int G = 64 / L;
  constexpr int LD = 4 * L * C;
  const int lane = threadIdx.x & 63, g = lane / L, j = lane % L;
  float4v rk[C];
#pragma unroll
  for (int c = 0; c < C; c++) rk[c] = *(const float4v*)(regk + c * 4 * L + 4 * j);
  if (SERIAL) {      // one group, list order: the sequential loop
    if (blockIdx.x != 0 || threadIdx.x >= L) return;
    const Rows<0> Um(U, 0), Vm(V, 0);
    for (int64_t t = 0; t < count; t++)
      visit_dimreg<L, C, 0>(Um, Vm, (int64_t)eu[first + t] * LD + 4 * j, (int64_t)ei[first + t] * LD + 4 * j, er[first + t], lr, rk);
    return;
  }
  const Rows<1> Um(U, ubytes), Vm(V, vbytes);   // rows are shared by all XCDs: coherent accesses
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t base = wave * 64; base < count; base += nwaves * 64) {
    const int nvalid = (int)(count - base < 64 ? count - base : 64);
    const bool ok = lane < nvalid;
    const int mu = ok ? eu[first + base + lane] : 0;
    const int mi = ok ? ei[first + base + lane] : 0;
    const float mr = ok ? er[first + base + lane] : 0.0f;
#pragma unroll 1
    for (int s = 0; s < L; s++) {
      const int e = s * G + g;
      const int u = __shfl(mu, e, 64);
      const int it = __shfl(mi, e, 64);
      const float r = __shfl(mr, e, 64);
      if (e < nvalid) visit_dimreg<L, C, 1>(Um, Vm, (int64_t)u * LD + 4 * j, (int64_t)it * LD + 4 * j, r, lr, rk);
    }
  }
}

template <int L, int C>
static int launch_dimreg(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count) {
  ProfScope ps(ctx, MFX_K_SGD);
  const uint64_t ub = (uint64_t)ctx->nU * ctx->ld * 4, vb = (uint64_t)ctx->nI * ctx->ld * 4;
  NEED(ub < (1ull << 32) && vb < (1ull << 32), MFX_E_ARG, "sgd with per-dimension regularisation: factor matrices must be < 4 GiB");
  if (o->mode == MFX_SGD_SERIAL) {
    hipLaunchKernelGGL((sgd_dimreg_kernel<L, C, true>), dim3(1), dim3(64), 0, ctx->stream, ctx->eu, ctx->ei, ctx->er, first, count,
                       ctx->U, ctx->V, (uint32_t)ub, (uint32_t)vb, o->learnRate, ctx->dimreg);
  } else {
    const int64_t waves = (count + 63) / 64;
    const int cap = o->blocks > 0 ? std::min(o->blocks, 8192) : std::max(8, std::min(2048, std::min(ctx->nU, ctx->nI) / 64));
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((waves + 3) / 4, cap));
    hipLaunchKernelGGL((sgd_dimreg_kernel<L, C, false>), dim3(blocks), dim3(256), 0, ctx->stream, ctx->eu, ctx->ei, ctx->er, first,
                       count, ctx->U, ctx->V, (uint32_t)ub, (uint32_t)vb, o->learnRate, ctx->dimreg);
  }
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

int mfx_launch_sgd_dimreg(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count) {
  NEED(o->mode == MFX_SGD_HOGWILD || o->mode == MFX_SGD_SERIAL, MFX_E_ARG,
       "per-dimension regularisation runs on MFX_SGD_HOGWILD / MFX_SGD_SERIAL (mode=%d)", o->mode);
  const int L = ctx->L, C = ctx->C;
  if (L == 4) return launch_dimreg<4, 1>(ctx, o, first, count);
  if (L == 8) return launch_dimreg<8, 1>(ctx, o, first, count);
  switch (C) {
    case 1: return launch_dimreg<16, 1>(ctx, o, first, count);
    case 2: return launch_dimreg<16, 2>(ctx, o, first, count);
    case 3: return launch_dimreg<16, 3>(ctx, o, first, count);
    case 4: return launch_dimreg<16, 4>(ctx, o, first, count);
  }
  return mfx_fail(ctx, MFX_E_ARG, "per-dimension regularisation: K <= 256");
}

extern "C" int mfx_sgd_set_dim_reg(mfx_ctx* ctx, const float* reg) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->U, MFX_E_STATE, "mfx_sgd_set_dim_reg: no model");
  HIPCHK(hipSetDevice(ctx->device));
  if (!reg) { dev_free(ctx->dimreg); return MFX_OK; }
  int rc;
  if (!ctx->dimreg && (rc = dev_alloc(ctx, &ctx->dimreg, (size_t)ctx->ld))) return rc;
  std::vector<float> h((size_t)ctx->ld, 0.0f);
  for (int k = 0; k < ctx->K; k++) h[k] = reg[k];
  HIPCHK(hipMemcpy(ctx->dimreg, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
  return MFX_OK;
}

// ---------------------------------------------------------------------------
// weighted norms (objectiveSing)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weighted_norm_kernel(const float* __restrict__ X, int32_t n, int ld, int K,
                                                            const uint8_t* __restrict__ inv, const float* __restrict__ w,
                                                            double* __restrict__ part) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
    if (inv[r]) continue;
    const float* x = X + r * ld;
    for (int k = 0; k < K; k++) acc += (double)(x[k] * x[k] * w[k]);   // float product, double sum (model.cpp:1844)
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

extern "C" int mfx_eval_weighted(mfx_ctx* ctx, int which, int snapshot, const float* w, mfx_eval_out* out) {
  if (!ctx) return MFX_E_ARG;
  NEED(w && out, MFX_E_ARG, "mfx_eval_weighted: NULL argument");
  int rc = mfx_eval(ctx, which, snapshot, 0, out);
  if (rc) return rc;
  const int nb = 256;
  float* dw = nullptr;
  double* part = nullptr;
  if ((rc = dev_alloc(ctx, &dw, (size_t)ctx->K)) || (rc = dev_alloc(ctx, &part, (size_t)2 * nb))) { dev_free(dw); return rc; }
  std::vector<double> h((size_t)2 * nb);
  hipError_t e = hipMemcpyAsync(dw, w, sizeof(float) * (size_t)ctx->K, hipMemcpyHostToDevice, ctx->stream);
  hipLaunchKernelGGL(weighted_norm_kernel, dim3(nb), dim3(256), 0, ctx->stream, snapshot ? ctx->Ubest : ctx->U, ctx->nU, ctx->ld,
                     ctx->K, ctx->invU, dw, part);
  hipLaunchKernelGGL(weighted_norm_kernel, dim3(nb), dim3(256), 0, ctx->stream, snapshot ? ctx->Vbest : ctx->V, ctx->nI, ctx->ld,
                     ctx->K, ctx->invI, dw, part + nb);
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), part, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  dev_free(dw); dev_free(part);
  NEED(e == hipSuccess, MFX_E_HIP, "mfx_eval_weighted: %s", hipGetErrorString(e));
  double a = 0, b = 0;
  for (int k = 0; k < nb; k++) { a += h[k]; b += h[nb + k]; }
  out->unorm2 = a;
  out->inorm2 = b;
  return MFX_OK;
}

// ---------------------------------------------------------------------------
// truncated SVD
// ---------------------------------------------------------------------------
namespace {
__global__ void svd_rand_kernel(float* __restrict__ Q, int64_t n, uint32_t seed) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t h = mfx_mix32((uint32_t)t * 0x9e3779b1U + seed) ^ mfx_mix32((uint32_t)(t >> 32) + seed * 0x85ebca6bU + 0x1234567U);
    Q[t] = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
  }
}
// out[row][0..r) (+)= sum over the segment of val * Q[ind][0..r); one wavefront per row segment
__global__ __launch_bounds__(256) void svd_spmm_kernel(const int32_t* __restrict__ seg_row, const int64_t* __restrict__ seg_beg,
                                                       const int64_t* __restrict__ seg_end, const int32_t* __restrict__ seg_slab,
                                                       int64_t nseg, const int32_t* __restrict__ ind, const float* __restrict__ val,
                                                       const float* __restrict__ Q, int rs, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t s = wave; s < nseg; s += nwaves) {
    float acc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const int64_t b = seg_beg[s], e = seg_end[s];
    for (int64_t x = b; x < e; x++) {
      const float v = val[x];
      const float* q = Q + (int64_t)ind[x] * rs;
#pragma unroll
      for (int t = 0; t < 5; t++)
        if (lane + 64 * t < rs) acc[t] = __builtin_fmaf(v, q[lane + 64 * t], acc[t]);
    }
    float* o = out + (int64_t)seg_row[s] * rs;
#pragma unroll
    for (int t = 0; t < 5; t++)
      if (lane + 64 * t < rs) {
        if (seg_slab[s] < 0) o[lane + 64 * t] = acc[t];
        else atomicAdd(o + lane + 64 * t, acc[t]);      // rows cut into several segments (out is zeroed first)
      }
  }
}
// G (double, zeroed) += Y^T Y over a chunk of rows; 64x64 output tile per workgroup, lower tiles only
__global__ __launch_bounds__(256) void svd_gram_kernel(const float* __restrict__ Y, int64_t n, int rs, int ntile, int64_t chunk,
                                                       double* __restrict__ G) {
  __shared__ float sa[16][64], sb[16][64];
  int ti = 0;
  { int p = blockIdx.y; while ((ti + 1) * (ti + 2) / 2 <= p) ti++; }
  const int tj = blockIdx.y - ti * (ti + 1) / 2;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float acc[4][4] = {};
  const int64_t r0 = (int64_t)blockIdx.x * chunk, r1 = min(n, r0 + chunk);
  for (int64_t base = r0; base < r1; base += 16) {
    for (int q = threadIdx.x; q < 16 * 64; q += 256) {
      const int rr = q >> 6, cc = q & 63;
      const int64_t row = base + rr;
      const int ca = ti * 64 + cc, cb = tj * 64 + cc;
      sa[rr][cc] = (row < r1 && ca < rs) ? Y[row * rs + ca] : 0.0f;
      sb[rr][cc] = (row < r1 && cb < rs) ? Y[row * rs + cb] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 16; rr++)
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = __builtin_fmaf(sa[rr][ty * 4 + a], sb[rr][tx * 4 + b], acc[a][b]);
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int gi = ti * 64 + ty * 4 + a, gj = tj * 64 + tx * 4 + b;
      if (gi < rs && gj < rs) atomicAdd(G + (int64_t)gi * rs + gj, (double)acc[a][b]);
    }
  (void)ntile;
}
// out = Y M (M [rs][rs] row-major); 16 rows per workgroup
__global__ __launch_bounds__(256) void svd_rmul_kernel(const float* __restrict__ Y, int64_t n, int rs, const float* __restrict__ M,
                                                       float* __restrict__ out) {
  extern __shared__ float sy[];   // [16][rs]
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  for (int q = threadIdx.x; q < 16 * rs; q += 256) {
    const int64_t row = r0 + q / rs;
    sy[q] = row < n ? Y[row * rs + q % rs] : 0.0f;
  }
  __syncthreads();
  const int rr = threadIdx.x >> 4, c0 = threadIdx.x & 15;
  if (r0 + rr >= n) return;
  for (int c = c0; c < rs; c += 16) {
    float a = 0.0f;
    for (int k = 0; k < rs; k++) a = __builtin_fmaf(sy[rr * rs + k], M[k * rs + c], a);
    out[(r0 + rr) * rs + c] = a;
  }
}
__global__ void svd_store_kernel(const float* __restrict__ src, int64_t n, int rs, int K, int ld, float* __restrict__ dst) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n * ld; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / ld;
    const int k = (int)(t - row * ld);
    dst[t] = k < K ? src[row * rs + k] : 0.0f;
  }
}

// ---- small dense helpers on the host, double ----
// lower Cholesky of the symmetric r x r matrix G (row-major); false when a pivot is not safely positive
bool chol(std::vector<double>& G, int r) {
  double dmax = 0;
  for (int i = 0; i < r; i++) dmax = std::max(dmax, G[(size_t)i * r + i]);
  for (int j = 0; j < r; j++) {
    double d = G[(size_t)j * r + j];
    for (int k = 0; k < j; k++) d -= G[(size_t)j * r + k] * G[(size_t)j * r + k];
    if (!(d > 1e-10 * dmax)) return false;
    d = std::sqrt(d);
    G[(size_t)j * r + j] = d;
    for (int i = j + 1; i < r; i++) {
      double s = G[(size_t)i * r + j];
      for (int k = 0; k < j; k++) s -= G[(size_t)i * r + k] * G[(size_t)j * r + k];
      G[(size_t)i * r + j] = s / d;
    }
  }
  return true;
}
// cyclic Jacobi: A = W diag(lam) W^T, eigenvalues descending, W column k = k-th eigenvector
void jacobi_eig(std::vector<double> A, int r, std::vector<double>& lam, std::vector<double>& W) {
  W.assign((size_t)r * r, 0.0);
  for (int i = 0; i < r; i++) W[(size_t)i * r + i] = 1.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0, dia = 0;
    for (int i = 0; i < r; i++) { dia += A[(size_t)i * r + i] * A[(size_t)i * r + i]; for (int j = 0; j < i; j++) off += A[(size_t)i * r + j] * A[(size_t)i * r + j]; }
    if (off <= 1e-30 * dia) break;
    for (int p = 0; p < r - 1; p++)
      for (int q = p + 1; q < r; q++) {
        const double apq = A[(size_t)p * r + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double theta = (A[(size_t)q * r + q] - A[(size_t)p * r + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < r; k++) {
          const double akp = A[(size_t)k * r + p], akq = A[(size_t)k * r + q];
          A[(size_t)k * r + p] = c * akp - s * akq;
          A[(size_t)k * r + q] = s * akp + c * akq;
        }
        for (int k = 0; k < r; k++) {
          const double apk = A[(size_t)p * r + k], aqk = A[(size_t)q * r + k];
          A[(size_t)p * r + k] = c * apk - s * aqk;
          A[(size_t)q * r + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < r; k++) {
          const double wkp = W[(size_t)k * r + p], wkq = W[(size_t)k * r + q];
          W[(size_t)k * r + p] = c * wkp - s * wkq;
          W[(size_t)k * r + q] = s * wkp + c * wkq;
        }
      }
  }
  std::vector<int> ord(r);
  for (int i = 0; i < r; i++) ord[i] = i;
  std::sort(ord.begin(), ord.end(), [&](int a, int b) { return A[(size_t)a * r + a] > A[(size_t)b * r + b]; });
  lam.resize(r);
  std::vector<double> W2((size_t)r * r);
  for (int k = 0; k < r; k++) {
    lam[k] = A[(size_t)ord[k] * r + ord[k]];
    for (int i = 0; i < r; i++) W2[(size_t)i * r + k] = W[(size_t)i * r + ord[k]];
  }
  W.swap(W2);
}

struct SvdWork {
  mfx_ctx* ctx;
  int rs;
  double* dG = nullptr;
  float* dM = nullptr;
  ~SvdWork() { dev_free(dG); dev_free(dM); }
  // G = Y^T Y on the host (double, symmetric)
  int gram(const float* Y, int64_t n, std::vector<double>& G) {
    HIPCHK(hipMemsetAsync(dG, 0, sizeof(double) * (size_t)rs * rs, ctx->stream));
    const int nt = (rs + 63) / 64;
    const int64_t chunk = 2048;
    hipLaunchKernelGGL(svd_gram_kernel, dim3((unsigned)((n + chunk - 1) / chunk), nt * (nt + 1) / 2), dim3(256), 0, ctx->stream, Y, n, rs, nt,
                       chunk, dG);
    HIPCHK(hipGetLastError());
    G.resize((size_t)rs * rs);
    HIPCHK(hipMemcpyAsync(G.data(), dG, sizeof(double) * G.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int ntile = nt;
    for (int i = 0; i < rs; i++)          // mirror: only tiles tj <= ti were computed
      for (int j = 0; j < rs; j++)
        if (j / 64 > i / 64) G[(size_t)i * rs + j] = G[(size_t)j * rs + i];
    (void)ntile;
    return MFX_OK;
  }
  // out = Y M
  int rmul(const float* Y, int64_t n, const std::vector<double>& M, float* out) {
    std::vector<float> f(M.size());
    for (size_t i = 0; i < M.size(); i++) f[i] = (float)M[i];
    HIPCHK(hipMemcpyAsync(dM, f.data(), sizeof(float) * f.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const size_t lds = sizeof(float) * 16 * (size_t)rs;
    hipLaunchKernelGGL(svd_rmul_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), lds, ctx->stream, Y, n, rs, dM, out);
    HIPCHK(hipGetLastError());
    return MFX_OK;
  }
  // Y <- orthonormal basis of range(Y) (CholeskyQR, twice; eigen-based when the Gramian is numerically singular)
  int orth(float*& Y, float*& tmp, int64_t n) {
    std::vector<double> G, M((size_t)rs * rs);
    for (int pass = 0; pass < 2; pass++) {
      int rc = gram(Y, n, G);
      if (rc) return rc;
      std::vector<double> Lc = G;
      std::fill(M.begin(), M.end(), 0.0);
      if (chol(Lc, rs)) {
        // M = (L^T)^-1: column c solves L^T m = e_c  (upper triangular)
        for (int c = 0; c < rs; c++) {
          for (int i = c; i >= 0; i--) {
            double s = i == c ? 1.0 : 0.0;
            for (int k = i + 1; k <= c; k++) s -= Lc[(size_t)k * rs + i] * M[(size_t)k * rs + c];
            M[(size_t)i * rs + c] = s / Lc[(size_t)i * rs + i];
          }
        }
      } else {
        std::vector<double> lam, W;
        jacobi_eig(G, rs, lam, W);
        for (int k = 0; k < rs; k++) {
          const double sc = lam[k] > 1e-10 * lam[0] ? 1.0 / std::sqrt(lam[k]) : 0.0;     // directions outside the range: dropped
          for (int i = 0; i < rs; i++) M[(size_t)i * rs + k] = W[(size_t)i * rs + k] * sc;
        }
      }
      if ((rc = rmul(Y, n, M, tmp))) return rc;
      std::swap(Y, tmp);
    }
    return MFX_OK;
  }
};
}  // namespace

extern "C" int mfx_svd_init(mfx_ctx* ctx, int32_t power_iters, int32_t oversample, uint32_t seed, float* singular) {
  if (!ctx) return MFX_E_ARG;
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  NEED(m.present && m.has_col, MFX_E_STATE, "mfx_svd_init: train matrix with column view needed");
  NEED(ctx->U, MFX_E_STATE, "mfx_svd_init: no model");
  NEED(m.nrows <= ctx->nU && m.ncols <= ctx->nI, MFX_E_ARG, "mfx_svd_init: matrix exceeds model");
  NEED(power_iters >= 0 && power_iters <= 1000 && oversample >= 0, MFX_E_ARG, "mfx_svd_init: power_iters=%d oversample=%d", power_iters, oversample);
  HIPCHK(hipSetDevice(ctx->device));
  const int K = ctx->K;
  const int r = std::max(1, std::min(std::min(K + oversample, 320), std::min(m.nrows, m.ncols)));
  const int rs = r;
  RowSegs *su, *si;
  int rc;
  if ((rc = mfx_get_segments(ctx, 0, &su)) || (rc = mfx_get_segments(ctx, 1, &si))) return rc;
  float *Yu = nullptr, *Yv = nullptr, *tmp = nullptr;
  SvdWork w;
  w.ctx = ctx; w.rs = rs;
  const int64_t nmax = std::max(m.nrows, m.ncols);
  auto cleanup = [&]() { dev_free(Yu); dev_free(Yv); dev_free(tmp); };
  if ((rc = dev_alloc(ctx, &Yu, (size_t)m.nrows * rs)) || (rc = dev_alloc(ctx, &Yv, (size_t)m.ncols * rs)) ||
      (rc = dev_alloc(ctx, &tmp, (size_t)nmax * rs)) || (rc = dev_alloc(ctx, &w.dG, (size_t)rs * rs)) ||
      (rc = dev_alloc(ctx, &w.dM, (size_t)rs * rs))) { cleanup(); return rc; }
  auto spmm = [&](int side, const float* Q, float* out) -> int {      // side 0: out[users] = R Q ; 1: out[items] = R^T Q
    RowSegs* sg = side == 0 ? su : si;
    const int64_t n = side == 0 ? m.nrows : m.ncols;
    HIPCHK(hipMemsetAsync(out, 0, sizeof(float) * (size_t)n * rs, ctx->stream));
    if (sg->nseg > 0) {
      const int blocks = (int)std::min<int64_t>((sg->nseg + 3) / 4, 256 * 16);
      hipLaunchKernelGGL(svd_spmm_kernel, dim3(blocks), dim3(256), 0, ctx->stream, sg->seg_row, sg->seg_beg, sg->seg_end, sg->seg_slab,
                         sg->nseg, side == 0 ? m.rowind : m.colind, side == 0 ? m.rowval : m.colval, Q, rs, out);
      HIPCHK(hipGetLastError());
    }
    return MFX_OK;
  };
  hipLaunchKernelGGL(svd_rand_kernel, dim3(1024), dim3(256), 0, ctx->stream, Yv, (int64_t)m.ncols * rs, seed * 2654435761U + 12345U);
  rc = w.orth(Yv, tmp, m.ncols);
  for (int it = 0; !rc && it < power_iters; it++) {
    if ((rc = spmm(0, Yv, Yu))) break;
    if ((rc = w.orth(Yu, tmp, m.nrows))) break;
    if ((rc = spmm(1, Yu, Yv))) break;
    rc = w.orth(Yv, tmp, m.ncols);
  }
  // Rayleigh-Ritz: B = R Qv, B^T B = W diag(sigma^2) W^T; left = B W / sigma, right = Qv W
  std::vector<double> G, lam, W;
  if (!rc) rc = spmm(0, Yv, Yu);
  if (!rc) rc = w.gram(Yu, m.nrows, G);
  if (rc) { cleanup(); return rc; }
  jacobi_eig(G, rs, lam, W);
  std::vector<double> Ml((size_t)rs * rs, 0.0), Mr((size_t)rs * rs, 0.0);
  std::vector<float> sig((size_t)K, 0.0f);
  for (int k = 0; k < std::min(K, rs); k++) {
    const double s = lam[k] > 0 ? std::sqrt(lam[k]) : 0.0;
    sig[k] = (float)s;
    const bool keep = s > 1e-7 * std::sqrt(std::max(lam[0], 0.0));
    for (int i = 0; i < rs; i++) {
      Ml[(size_t)i * rs + k] = keep ? W[(size_t)i * rs + k] / s : 0.0;
      Mr[(size_t)i * rs + k] = keep ? W[(size_t)i * rs + k] : 0.0;
    }
  }
  float* tmp2 = nullptr;
  if ((rc = dev_alloc(ctx, &tmp2, (size_t)nmax * rs))) { cleanup(); return rc; }
  if (!(rc = w.rmul(Yu, m.nrows, Ml, tmp))) {
    hipLaunchKernelGGL(svd_store_kernel, dim3(2048), dim3(256), 0, ctx->stream, tmp, (int64_t)m.nrows, rs, std::min(K, rs), ctx->ld, ctx->U);
    if (!(rc = w.rmul(Yv, m.ncols, Mr, tmp2)))
      hipLaunchKernelGGL(svd_store_kernel, dim3(2048), dim3(256), 0, ctx->stream, tmp2, (int64_t)m.ncols, rs, std::min(K, rs), ctx->ld, ctx->V);
  }
  hipError_t e = hipStreamSynchronize(ctx->stream);
  dev_free(tmp2);
  cleanup();
  if (rc) return rc;
  NEED(e == hipSuccess, MFX_E_HIP, "mfx_svd_init: %s", hipGetErrorString(e));
  if (singular) for (int k = 0; k < K; k++) singular[k] = sig[k];
  return MFX_OK;
}
