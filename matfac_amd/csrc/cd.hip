// cd.hip -- cyclic coordinate descent: the two loops of ModelMF::trainCCD (modelMF.cpp:1528-1565 users,
// :1567-1605 items).
//
// Per row (a user, then an item) and per factor k in that row's own shuffled order:
//   num   = sum_e (res_e + x_k * y_ek) * y_ek      float products, double accumulation   (:1546, :1585)
//   denom = reg + sum_e y_ek * y_ek                                                       (:1547, :1586)
//   new   = num / denom ;  res_e -= (new - x_k) * y_ek  in double, stored float ;  x_k = new
// The k-steps of a row depend on each other through its residuals, rows are independent: one team of
// lanes per row.  Short rows (<= 256 ratings) get a wavefront, rows up to 2048 ratings a 256-thread
// workgroup, longer ones a 1024-thread workgroup; up to 16384 ratings the residuals, indices and the k-th
// elements of the gathered rows live in registers for the whole sweep, beyond that they stream through L2.  The gathered side is read from a transposed copy
// Yt[k][n] (refreshed once per sweep): a row's gathers for one k then fall into one dense vector, which
// is what makes the 50k-rating items coalesce.
// The reference keeps the other view's residuals current through a binary search per rating per k
// (:1556-1561, :1595-1600).  Both views receive the same subtractions in the same order, so they stay
// bit-identical entry by entry; here the swept view is copied into the other one through the
// CSR<->CSC position map once per sweep instead.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "mfx_internal.h"

namespace {
constexpr int SHORT_E = 4, SHORT_MAX = 64 * SHORT_E;      // wavefront rows
constexpr int MID_NT = 256, MID_E = 8, MID_MAX = MID_NT * MID_E;           // 256-thread workgroup rows
constexpr int LONG_NT = 1024, LONG_E = 16;                                // registers up to 16384, streaming beyond

struct CdState {
  float *res_row = nullptr, *res_col = nullptr, *Yt = nullptr;
  uint32_t* c2r = nullptr;             // column-view position -> CSR position
  int32_t* rows[2] = {nullptr, nullptr};   // per side: long rows (longest first), middle rows, short rows
  int32_t nlong[2] = {0, 0}, nmid[2] = {0, 0}, nshort[2] = {0, 0};
  uint16_t* order = nullptr;
  int64_t order_cap = 0;
  hipStream_t s2 = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
};
CdState* cd_state(mfx_ctx* ctx) { return (CdState*)ctx->cd; }
}  // namespace

void mfx_cd_free_internal(mfx_ctx* ctx) {
  CdState* s = cd_state(ctx);
  if (!s) return;
  dev_free(s->res_row); dev_free(s->res_col); dev_free(s->Yt); dev_free(s->c2r);
  dev_free(s->rows[0]); dev_free(s->rows[1]); dev_free(s->order);
  if (s->ev_a) (void)hipEventDestroy(s->ev_a);
  if (s->ev_b) (void)hipEventDestroy(s->ev_b);
  if (s->s2) (void)hipStreamDestroy(s->s2);
  delete s;
  ctx->cd = nullptr;
}

// Yt[k][n] = Y[n][ld] (k < K)
__global__ __launch_bounds__(256) void cd_transpose_kernel(const float* __restrict__ Y, int n, int ld, int K, float* __restrict__ Yt) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    const int r = r0 + j, k = k0 + tx;
    tile[j][tx] = (r < n && k < K) ? Y[(int64_t)r * ld + k] : 0.0f;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int k = k0 + j, r = r0 + tx;
    if (k < K && r < n) Yt[(int64_t)k * n + r] = tile[tx][j];
  }
}

__global__ void cd_gather_kernel(const float* __restrict__ src, const uint32_t* __restrict__ c2r, int64_t nnz, float* __restrict__ dst_col) {
  for (int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; d < nnz; d += (int64_t)gridDim.x * blockDim.x) dst_col[d] = src[c2r[d]];
}
__global__ void cd_scatter_kernel(const float* __restrict__ src_col, const uint32_t* __restrict__ c2r, int64_t nnz, float* __restrict__ dst) {
  for (int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; d < nnz; d += (int64_t)gridDim.x * blockDim.x) dst[c2r[d]] = src_col[d];
}
// the given column view must be the stable one the map describes
__global__ void cd_check_map_kernel(const uint32_t* __restrict__ c2r, const int32_t* __restrict__ rowid, const float* __restrict__ rowval,
                                    const int32_t* __restrict__ colind, const float* __restrict__ colval, int64_t nnz, int* bad) {
  for (int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; d < nnz; d += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t e = c2r[d];
    if (colind[d] != rowid[e] || __float_as_uint(colval[d]) != __float_as_uint(rowval[e])) *bad = 1;
  }
}

__device__ __forceinline__ int cd_step_k(const uint16_t* __restrict__ order, int row, int K, int step, int bits, uint32_t k0, uint32_t k1) {
  if (order) return (int)order[(int64_t)row * K + step];
  if (K == 1) return 0;
  // device order: one permutation per sweep, shared by all rows -- the rows in flight then gather from the
  // same few columns of Yt, which stay in L2
  const int ab = bits / 2;
  return (int)mfx_perm_index(step, K, ab, bits - ab, k0, k1);
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

struct CdArgs {
  const int64_t* ptr; const int32_t* ind; float* res; float* X; const float* Yt;
  int n_other, ld, K, bits; float reg; const int32_t* rows; int nrows;
  const uint16_t* order; uint32_t k0, k1;
};

// one wavefront per row of <= SHORT_MAX ratings
__global__ __launch_bounds__(256) void cd_short_kernel(CdArgs a) {
  const int lane = threadIdx.x & 63;
  const int w = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (w >= a.nrows) return;
  const int row = a.rows[w];
  const int64_t b = a.ptr[row];
  const int d = (int)(a.ptr[row + 1] - b);
  float r[SHORT_E], y[SHORT_E];
  int idx[SHORT_E];
#pragma unroll
  for (int t = 0; t < SHORT_E; t++) {
    const int e = lane + 64 * t;
    idx[t] = e < d ? a.ind[b + e] : -1;
    r[t] = e < d ? a.res[b + e] : 0.0f;
  }
  float* xrow = a.X + (int64_t)row * a.ld;
  for (int step = 0; step < a.K; step++) {
    const int k = cd_step_k(a.order, row, a.K, step, a.bits, a.k0, a.k1);
    const float xk = xrow[k];
    const float* yk = a.Yt + (int64_t)k * a.n_other;
    double num = 0.0, den = 0.0;
#pragma unroll
    for (int t = 0; t < SHORT_E; t++) {
      y[t] = idx[t] >= 0 ? yk[idx[t]] : 0.0f;
      const float s = r[t] + xk * y[t];
      num += (double)(s * y[t]);
      den += (double)(y[t] * y[t]);
    }
    num = wave_sum(num);
    den = wave_sum(den);
    const double nv = num / ((double)a.reg + den);
    const double dx = nv - (double)xk;
#pragma unroll
    for (int t = 0; t < SHORT_E; t++) r[t] = (float)((double)r[t] - dx * (double)y[t]);
    if (lane == 0) xrow[k] = (float)nv;
  }
#pragma unroll
  for (int t = 0; t < SHORT_E; t++) {
    const int e = lane + 64 * t;
    if (e < d) a.res[b + e] = r[t];
  }
}

// one NT-thread workgroup per longer row
template <int NT, int E>
__global__ __launch_bounds__(NT) void cd_team_kernel(CdArgs a) {
  constexpr int LONG_NT = NT, LONG_E = E, LONG_REG_MAX = NT * E;
  __shared__ double red[2][LONG_NT / 64][2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row = a.rows[blockIdx.x];
  const int64_t b = a.ptr[row];
  const int64_t d = a.ptr[row + 1] - b;
  const bool in_regs = d <= LONG_REG_MAX;
  float r[LONG_E], y[LONG_E];
  int idx[LONG_E];
  if (in_regs) {
#pragma unroll
    for (int t = 0; t < LONG_E; t++) {
      const int64_t e = tid + (int64_t)LONG_NT * t;
      idx[t] = e < d ? a.ind[b + e] : -1;
      r[t] = e < d ? a.res[b + e] : 0.0f;
    }
  }
  float* xrow = a.X + (int64_t)row * a.ld;
  const int32_t* ind = a.ind + b;
  float* res = a.res + b;
  for (int step = 0; step < a.K; step++) {
    const int k = cd_step_k(a.order, row, a.K, step, a.bits, a.k0, a.k1);
    const float xk = xrow[k];
    const float* yk = a.Yt + (int64_t)k * a.n_other;
    double num = 0.0, den = 0.0;
    if (in_regs) {
#pragma unroll
      for (int t = 0; t < LONG_E; t++) {
        y[t] = idx[t] >= 0 ? yk[idx[t]] : 0.0f;
        const float s = r[t] + xk * y[t];
        num += (double)(s * y[t]);
        den += (double)(y[t] * y[t]);
      }
    } else {
#pragma unroll 4
      for (int64_t e = tid; e < d; e += LONG_NT) {
        const float yv = yk[ind[e]];
        const float s = res[e] + xk * yv;
        num += (double)(s * yv);
        den += (double)(yv * yv);
      }
    }
    num = wave_sum(num);
    den = wave_sum(den);
    const int p = step & 1;
    if (lane == 0) { red[p][wv][0] = num; red[p][wv][1] = den; }
    __syncthreads();
    num = 0.0; den = 0.0;
#pragma unroll
    for (int q = 0; q < LONG_NT / 64; q++) { num += red[p][q][0]; den += red[p][q][1]; }
    const double nv = num / ((double)a.reg + den);
    const double dx = nv - (double)xk;
    if (in_regs) {
#pragma unroll
      for (int t = 0; t < LONG_E; t++) r[t] = (float)((double)r[t] - dx * (double)y[t]);
    } else {
#pragma unroll 4
      for (int64_t e = tid; e < d; e += LONG_NT) res[e] = (float)((double)res[e] - dx * (double)yk[ind[e]]);
    }
    if (tid == 0) xrow[k] = (float)nv;
  }
  if (in_regs) {
#pragma unroll
    for (int t = 0; t < LONG_E; t++) {
      const int64_t e = tid + (int64_t)LONG_NT * t;
      if (e < d) res[e] = r[t];
    }
  }
}

// rows with ratings, long ones first and longest first
static int build_row_list(mfx_ctx* ctx, CdState* s, int side, const int64_t* dev_ptr, int32_t n) {
  std::vector<int64_t> ptr((size_t)n + 1);
  HIPCHK(hipMemcpy(ptr.data(), dev_ptr, sizeof(int64_t) * ptr.size(), hipMemcpyDeviceToHost));
  std::vector<int32_t> lng, mid, sht;
  for (int32_t r = 0; r < n; r++) {
    const int64_t d = ptr[r + 1] - ptr[r];
    if (d > MID_MAX) lng.push_back(r);
    else if (d > SHORT_MAX) mid.push_back(r);
    else if (d > 0) sht.push_back(r);
  }
  auto longer = [&](int32_t x, int32_t y) { return ptr[x + 1] - ptr[x] > ptr[y + 1] - ptr[y]; };
  std::stable_sort(lng.begin(), lng.end(), longer);
  std::stable_sort(mid.begin(), mid.end(), longer);
  s->nlong[side] = (int32_t)lng.size();
  s->nmid[side] = (int32_t)mid.size();
  s->nshort[side] = (int32_t)sht.size();
  lng.insert(lng.end(), mid.begin(), mid.end());
  lng.insert(lng.end(), sht.begin(), sht.end());
  int rc = dev_alloc(ctx, &s->rows[side], lng.size());
  if (rc) return rc;
  if (!lng.empty()) HIPCHK(hipMemcpy(s->rows[side], lng.data(), sizeof(int32_t) * lng.size(), hipMemcpyHostToDevice));
  return MFX_OK;
}

extern "C" int mfx_ccd_begin(mfx_ctx* ctx) {
  if (!ctx) return MFX_E_ARG;
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  NEED(m.present && m.has_col, MFX_E_STATE, "mfx_ccd_begin: train matrix with column view needed");
  NEED(ctx->U, MFX_E_STATE, "mfx_ccd_begin: no model");
  NEED(m.nrows <= ctx->nU && m.ncols <= ctx->nI, MFX_E_ARG, "mfx_ccd_begin: matrix exceeds model");
  NEED(m.nnz < ((int64_t)1 << 31), MFX_E_ARG, "mfx_ccd_begin: nnz must be < 2^31");
  HIPCHK(hipSetDevice(ctx->device));
  mfx_cd_free_internal(ctx);
  CdState* s = new CdState;
  ctx->cd = s;
  int rc;
  const size_t nnz = (size_t)m.nnz;
  if ((rc = dev_alloc(ctx, &s->res_row, nnz)) || (rc = dev_alloc(ctx, &s->res_col, nnz)) ||
      (rc = dev_alloc(ctx, &s->Yt, (size_t)ctx->K * (size_t)std::max(ctx->nU, ctx->nI))))
    return rc;
  if ((rc = mfx_build_c2r_map_device(ctx, m, &s->c2r))) return rc;
  if (nnz) {
    int* bad;
    if ((rc = dev_alloc(ctx, &bad, 1))) return rc;
    hipError_t e1 = hipMemsetAsync(bad, 0, sizeof(int), ctx->stream);
    hipLaunchKernelGGL(cd_check_map_kernel, dim3(2048), dim3(256), 0, ctx->stream, s->c2r, m.rowid, m.rowval, m.colind, m.colval, m.nnz, bad);
    int hbad = 0;
    hipError_t e2 = hipMemcpy(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost);
    dev_free(bad);
    NEED(e1 == hipSuccess && e2 == hipSuccess, MFX_E_HIP, "mfx_ccd_begin: map check failed to run");
    NEED(!hbad, MFX_E_ARG, "mfx_ccd_begin: the column view is not the stable transpose of the row view (gk_csr_CreateIndex order)");
    // res = gk_csr_Dup(trainMat) (modelMF.cpp:1509)
    HIPCHK(hipMemcpyAsync(s->res_row, m.rowval, sizeof(float) * nnz, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(s->res_col, m.colval, sizeof(float) * nnz, hipMemcpyDeviceToDevice, ctx->stream));
  }
  if ((rc = build_row_list(ctx, s, 0, m.rowptr, m.nrows))) return rc;
  if ((rc = build_row_list(ctx, s, 1, m.colptr, m.ncols))) return rc;
  HIPCHK(hipStreamCreateWithFlags(&s->s2, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&s->ev_a, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&s->ev_b, hipEventDisableTiming));
  // uFac(u, k) = 0 (modelMF.cpp:1516-1522)
  HIPCHK(hipMemsetAsync(ctx->U, 0, sizeof(float) * (size_t)ctx->nU * ctx->ld, ctx->stream));
  return MFX_OK;
}

extern "C" int mfx_ccd_sweep(mfx_ctx* ctx, int32_t side, float reg, const uint16_t* order, uint32_t seed, int32_t iter) {
  if (!ctx) return MFX_E_ARG;
  CdState* s = cd_state(ctx);
  NEED(s, MFX_E_STATE, "mfx_ccd_sweep: call mfx_ccd_begin first");
  NEED(side == MFX_SIDE_USERS || side == MFX_SIDE_ITEMS, MFX_E_ARG, "mfx_ccd_sweep: side=%d", side);
  HIPCHK(hipSetDevice(ctx->device));
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const int K = ctx->K;
  const int32_t nmine = side == MFX_SIDE_USERS ? m.nrows : m.ncols;
  const int32_t nother = side == MFX_SIDE_USERS ? ctx->nI : ctx->nU;
  const float* Y = side == MFX_SIDE_USERS ? ctx->V : ctx->U;
  if (order) {
    const int64_t need = (int64_t)nmine * K;
    if (need > s->order_cap) {
      dev_free(s->order);
      int rc = dev_alloc(ctx, &s->order, (size_t)need);
      if (rc) return rc;
      s->order_cap = need;
    }
    HIPCHK(hipMemcpyAsync(s->order, order, sizeof(uint16_t) * (size_t)need, hipMemcpyHostToDevice, ctx->stream));
  }
  ProfScope ps(ctx, MFX_K_CD);
  hipLaunchKernelGGL(cd_transpose_kernel, dim3((nother + 63) / 64, (K + 63) / 64), dim3(256), 0, ctx->stream, Y, nother, ctx->ld, K, s->Yt);
  CdArgs a;
  a.ptr = side == MFX_SIDE_USERS ? m.rowptr : m.colptr;
  a.ind = side == MFX_SIDE_USERS ? m.rowind : m.colind;
  a.res = side == MFX_SIDE_USERS ? s->res_row : s->res_col;
  a.X = side == MFX_SIDE_USERS ? ctx->U : ctx->V;
  a.Yt = s->Yt;
  a.n_other = nother; a.ld = ctx->ld; a.K = K; a.reg = reg;
  a.bits = 2;
  while ((1 << a.bits) < K) a.bits++;
  a.order = order ? s->order : nullptr;
  a.k0 = mfx_mix32(seed ^ 0x51ed270bU) + (uint32_t)iter * 0x9e3779b9U + (uint32_t)side * 0x7f4a7c15U;
  a.k1 = mfx_mix32(seed * 0x85ebca6bU + 0xc2b2ae35U) ^ mfx_mix32((uint32_t)iter * 2U + (uint32_t)side + 0x165667b1U);
  const int nl = s->nlong[side], nm = s->nmid[side], ns = s->nshort[side];
  // the short rows run on a second stream next to the longer ones (rows are independent)
  HIPCHK(hipEventRecord(s->ev_a, ctx->stream));
  HIPCHK(hipStreamWaitEvent(s->s2, s->ev_a, 0));
  if (nl > 0) {
    a.rows = s->rows[side]; a.nrows = nl;
    hipLaunchKernelGGL((cd_team_kernel<LONG_NT, LONG_E>), dim3(nl), dim3(LONG_NT), 0, ctx->stream, a);
  }
  if (nm > 0) {
    a.rows = s->rows[side] + nl; a.nrows = nm;
    hipLaunchKernelGGL((cd_team_kernel<MID_NT, MID_E>), dim3(nm), dim3(MID_NT), 0, ctx->stream, a);
  }
  if (ns > 0) {
    a.rows = s->rows[side] + nl + nm; a.nrows = ns;
    hipLaunchKernelGGL(cd_short_kernel, dim3((ns + 3) / 4), dim3(256), 0, s->s2, a);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(s->ev_b, s->s2));
  HIPCHK(hipStreamWaitEvent(ctx->stream, s->ev_b, 0));
  // bring the other view up to date (the reference's binSearch updates, :1556-1561 / :1595-1600)
  if (m.nnz > 0) {
    if (side == MFX_SIDE_USERS)
      hipLaunchKernelGGL(cd_gather_kernel, dim3(4096), dim3(256), 0, ctx->stream, s->res_row, s->c2r, m.nnz, s->res_col);
    else
      hipLaunchKernelGGL(cd_scatter_kernel, dim3(4096), dim3(256), 0, ctx->stream, s->res_col, s->c2r, m.nnz, s->res_row);
    HIPCHK(hipGetLastError());
  }
  return MFX_OK;
}

extern "C" int mfx_ccd_end(mfx_ctx* ctx) {
  if (!ctx) return MFX_E_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  mfx_cd_free_internal(ctx);
  return MFX_OK;
}

extern "C" int mfx_debug_ccd_residuals(mfx_ctx* ctx, float* res_row, float* res_col) {
  if (!ctx) return MFX_E_ARG;
  CdState* s = cd_state(ctx);
  NEED(s, MFX_E_STATE, "mfx_debug_ccd_residuals: call mfx_ccd_begin first");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const size_t nnz = (size_t)ctx->mat[MFX_MAT_TRAIN].nnz;
  if (nnz && res_row) HIPCHK(hipMemcpy(res_row, s->res_row, sizeof(float) * nnz, hipMemcpyDeviceToHost));
  if (nnz && res_col) HIPCHK(hipMemcpy(res_col, s->res_col, sizeof(float) * nnz, hipMemcpyDeviceToHost));
  return MFX_OK;
}
