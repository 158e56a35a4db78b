// ccd.hip -- CCD++ rank-one sweeps for gfx950 (row view + driver; the column view is ccd_cols.hip).
//
// Replaces ModelMF::trainCCDPP (modelMF.cpp:1013-1121) and trainCCDPPFreqAdap
// (:1258-1360).  State: the residual R - U V^T on BOTH views (CSR values res_row,
// CSC values res_col), exactly as the reference keeps gk_csr_Dup(trainMat)'s rowval
// and colval in lock-step.  All passes are HBM-streaming:
//   row pass   u_k[u] = sum_i res_ui v_k[i] / (uReg + sum_i v_k[i]^2)   (:1062-1074)
//   col pass   v_k[i] = sum_u res_ui u_k[u] / (iReg + sum_u u_k[u]^2)   (:1078-1090)
//   residual   res_ui +-= u_k[u] v_k[i] on both views                   (:1032-1056, :1095-1116)
// Arithmetic as in the reference: the products are float*float, num/denom accumulate
// in double, the quotient is rounded to float once.  Both views are PADDED (ccd_blocks.h:
// every row / (strip, column) piece starts on a multiple of 8 entries, the view is whole
// 128-entry trips); a pass is a segmented reduction over the trips -- a 16-lane group per
// trip, the sums of the pieces inside it to consecutive slots -- and the finishing kernels
// add a row's slots in order and take the quotient, so every sum has a fixed association
// (reproducible).  Item ids are 16 bits wherever v_k is staged in LDS.
#include <algorithm>

#include "ccd_blocks.h"

void mfx_ccd_free_internal(mfx_ctx* ctx) {
  dev_free(ctx->res_row); dev_free(ctx->res_col); dev_free(ctx->uk); dev_free(ctx->vk);
  dev_free(ctx->uk_pend); dev_free(ctx->vk_pend); dev_free(ctx->uk_init); dev_free(ctx->ccd_rpair);
  ctx->ccd_pending = false;
  dev_free(ctx->ccd_part); dev_free(ctx->colid); dev_free(ctx->ccd_ind16); dev_free(ctx->ccd_ind32); dev_free(ctx->ccd_rowid); dev_free(ctx->ccd_rpos);
  dev_free(ctx->ccd_rfirst); dev_free(ctx->ccd_rcnt); dev_free(ctx->ccd_lrow);
  mfx_blocks_free(ctx->ccd_blocks);
  ctx->ccd_nnzp = 0;
  ctx->ccd_nlrow = 0;
  mfx_ccd_cols_free(ctx);
  ctx->ccd_part_cap = 0;
  ctx->ccd_active = false;
}

__global__ void expand_colid_kernel(const int64_t* __restrict__ colptr, int32_t ncols,
                                    int32_t* __restrict__ colid) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t c = wave; c < ncols; c += nwaves) {
    const int64_t b = colptr[c], e = colptr[c + 1];
    for (int64_t t = b + lane; t < e; t += 64) colid[t] = (int32_t)c;
  }
}

// column k of a factor matrix <-> dense vector
// (both factor matrices in ONE launch: a rank-one step is ~25 launches of 5..250 us, and every launch costs ~3 us of gap)
__global__ void extract_col_kernel(const float* __restrict__ X, int32_t n, const float* __restrict__ Y, int32_t m, int ld, int k,
                                   float* __restrict__ xk, float* __restrict__ yk) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) xk[t] = X[t * ld + k];
  else if (t - n < m) yk[t - n] = Y[(t - n) * ld + k];
}
__global__ void store_col_kernel(float* __restrict__ X, int32_t n, float* __restrict__ Y, int32_t m, int ld, int k,
                                 const float* __restrict__ xk, const float* __restrict__ yk) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) X[t * ld + k] = xk[t];
  else if (t - n < m) Y[(t - n) * ld + k] = yk[t - n];
}

// res[e] (+/-)= a[rowOf[e]] * b[colOf[e]]   -- float product, then float add/sub.  ia: the row of every EIGHT entries of the padded view.
// Streaming: 16 bytes per lane per array (4 ratings per thread), the two factor vectors are L2 resident.
// LDSB: the vector b (v_k in the row view) is first staged in LDS -- a 4-byte gather from L2 moves a whole
// 128-byte line, which made these kernels L2-bandwidth-bound; from LDS the gather is free.
extern __shared__ __attribute__((aligned(16))) float ccd_lds[];

// (slot n holds +0.0: the pass loop's masked entries gather it)
__device__ __forceinline__ void stage_vector(const float* __restrict__ v, int n) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int n4 = n >> 2;
  for (int q = threadIdx.x; q < n4; q += blockDim.x) ((f4*)ccd_lds)[q] = ((const f4*)v)[q];
  for (int q = (n4 << 2) + threadIdx.x; q < n; q += blockDim.x) ccd_lds[q] = v[q];
  if (threadIdx.x == 0) ccd_lds[n] = 0.0f;
  __syncthreads();
}

// The item ids of the row view: 16-bit wherever the gathered item vector is staged in LDS (it fits for at most 38 400 items), so
// the streaming kernels move 2 instead of 4 bytes per entry for them (ctx->ccd_ind16, made once by mfx_ccdpp_begin).
template <bool LDS> struct ItemIdx { typedef int32_t type; };
template <> struct ItemIdx<true> { typedef uint16_t type; };
// The padded row view (ccd_blocks.h): row r copied to [rpos[r], rpos[r+1]), the padding entries behind it with residual 0 and the id
// `zero_id` (the +0.0 slot behind v_k); then the tail of the last trip and the slack the pass loop may read.  rowid holds the row of
// every EIGHT entries (rows start on multiples of 8): the residual kernels read 0.5 instead of 4 bytes of it per entry.
template <typename IdxT>
__global__ __launch_bounds__(256) void rowview_build_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ rowind,
                                                            const float* __restrict__ rowval, const int64_t* __restrict__ rpos,
                                                            int32_t nrows, int zero_id, int64_t nnzp, int64_t nalloc,
                                                            float* __restrict__ res, IdxT* __restrict__ ids, int32_t* __restrict__ rowid) {
  const int j = threadIdx.x & 15;
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t r = grp; r < nrows; r += ngrp) {
    const int64_t src = rowptr[r], n = rowptr[r + 1] - src, d = rpos[r], np = rpos[r + 1] - d;
    for (int64_t t = j; t < np; t += 16) {
      const bool real = t < n;
      res[mfx_blk_mem_of(d + t)] = real ? rowval[src + t] : 0.0f;
      ids[d + t] = (IdxT)(real ? rowind[src + t] : zero_id);
      if ((t & 7) == 0) rowid[(d + t) >> 3] = (int32_t)r;
    }
  }
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = rpos[nrows] + tid; t < nalloc; t += nth) {      // (nalloc is a whole number of trips: the image of [.., nalloc) is inside it)
    res[mfx_blk_mem_of(t)] = 0.0f;
    ids[t] = (IdxT)zero_id;
    if ((t & 7) == 0) rowid[t >> 3] = nrows > 0 ? nrows - 1 : 0;      // (the slack too: the fused sweep gathers with it)
  }
}
// test hook: the padded residuals back in CSR order
__global__ __launch_bounds__(256) void rowview_export_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ rpos, int32_t nrows,
                                                             const float* __restrict__ res, float* __restrict__ out) {
  const int j = threadIdx.x & 15;
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t r = grp; r < nrows; r += ngrp) {
    const int64_t src = rowptr[r], n = rowptr[r + 1] - src, d = rpos[r];
    for (int64_t t = j; t < n; t += 16) out[src + t] = res[mfx_blk_mem_of(d + t)];
  }
}

template <int SIGN, bool LDSB>
__global__ __launch_bounds__(1024) void resid_update_kernel(float* __restrict__ res, const int32_t* __restrict__ ia,
                                                            const typename ItemIdx<LDSB>::type* __restrict__ ib,
                                                            const float* __restrict__ a, const float* __restrict__ bg,
                                                            int nb, int64_t n) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef int i4 __attribute__((ext_vector_type(4)));
  if (LDSB) stage_vector(bg, nb);
  const float* b = LDSB ? ccd_lds : bg;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {     // (n is a multiple of 128)
    const int64_t ql = (q & ~(int64_t)31) | ((q & 15) << 1) | ((q >> 4) & 1);     // residual quad q (memory order) holds entry quad ql (mfx_blk_entry_of)
    const i4 y = MfxCcdTrip::load4(ib + 4 * ql);
    f4 r = ((const f4*)res)[q];
    const float au = a[ia[ql >> 1]];         // the four entries of a quad lie in ONE row
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float prod = au * b[y[e]];
      r[e] = SIGN > 0 ? r[e] + prod : r[e] - prod;
    }
    ((f4*)res)[q] = r;
  }
}

// res = (res - a0*b0) + a1*b1 in one sweep: the subtract of factor k (modelMF.cpp:1095-1116) and the add-back
// of the next factor (:1032-1056) touch the same residual entries back to back; the two roundings are kept.
template <bool LDSB>
__global__ __launch_bounds__(1024) void resid_fused_kernel(float* __restrict__ res, const int32_t* __restrict__ ia,
                                                           const typename ItemIdx<LDSB>::type* __restrict__ ib,
                                                           const float* __restrict__ a0, const float* __restrict__ b0g,
                                                           const float* __restrict__ a1, const float* __restrict__ b1g,
                                                           int nb, int64_t n) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef int i4 __attribute__((ext_vector_type(4)));
  const int nbs = (nb + 4) & ~3;            // stride of the two staged vectors: nb values and the +0.0 the padding entries gather
  if (LDSB) {
    const int n4 = nb >> 2;
    for (int q = threadIdx.x; q < n4; q += blockDim.x) {
      ((f4*)ccd_lds)[q] = ((const f4*)b0g)[q];
      ((f4*)(ccd_lds + nbs))[q] = ((const f4*)b1g)[q];
    }
    for (int q = (n4 << 2) + threadIdx.x; q < nb; q += blockDim.x) {
      ccd_lds[q] = b0g[q];
      ccd_lds[nbs + q] = b1g[q];
    }
    if (threadIdx.x == 0) { ccd_lds[nb] = 0.0f; ccd_lds[nbs + nb] = 0.0f; }
    __syncthreads();
  }
  const float* b0 = LDSB ? ccd_lds : b0g;
  const float* b1 = LDSB ? ccd_lds + nbs : b1g;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {     // (n is a multiple of 128)
    const int64_t ql = (q & ~(int64_t)31) | ((q & 15) << 1) | ((q >> 4) & 1);     // residual quad q (memory order) holds entry quad ql (mfx_blk_entry_of)
    const i4 y = MfxCcdTrip::load4(ib + 4 * ql);
    f4 r = ((const f4*)res)[q];
    const int row = ia[ql >> 1];             // the four entries of a quad lie in ONE row
    const float au0 = a0[row], au1 = a1[row];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float p0 = au0 * b0[y[e]];
      const float p1 = au1 * b1[y[e]];
      r[e] = (r[e] - p0) + p1;
    }
    ((f4*)res)[q] = r;
  }
}

__device__ __forceinline__ double group16_sum(double v) { return mfx_row16_sum(v); }

// Row pass: workgroup w works through window w of the padded row view (ccd_blocks.h), v_k staged in LDS when it fits.  The quotient
// is taken by the finishing kernel: a double division is ~35 instructions.  (trainCCDPPFreqAdap's rule applies to items only:
// modelMF.cpp:1336-1342 -- the row pass never sees a threshold.)
template <bool LDSO>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void ccd_pass_kernel(const int2* __restrict__ rec, const int32_t* __restrict__ wg_t0,
                                                       const int32_t* __restrict__ wg_n, const int64_t* __restrict__ wg_rec,
                                                       const int32_t* __restrict__ wg_stride, const float* __restrict__ res,
                                                       const typename ItemIdx<LDSO>::type* __restrict__ ind,
                                                       const float* __restrict__ otherg, int nother, double* __restrict__ part,
                                                       uint32_t part_bytes) {
  if (LDSO) stage_vector(otherg, nother);
  const float* other = LDSO ? ccd_lds : otherg;   // (the global vectors of a CCD++ session carry one more element, +0.0: mfx_ccdpp_begin)
  const int j = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int w = blockIdx.x, wn = wg_n[w];
  const int64_t e0 = (int64_t)wg_t0[w] * MFX_BLK_E;
  mfx_ccd_block_loop<typename ItemIdx<LDSO>::type>(rec + wg_rec[w], wn, wg_stride[w], g, res + e0, ind + e0, other, j, part, part_bytes);
}

// The FIRST row sweep of a factor with the residual update on the way (ccd_blocks.h, FUSE): both item vectors of the update in LDS
// (the finished factor's v and the new one's, which is also what the sums gather), the users' pair per eight entries from rpair.
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void ccd_pass_fused_kernel(
    const int2* __restrict__ rec, const int32_t* __restrict__ wg_t0, const int32_t* __restrict__ wg_n, const int64_t* __restrict__ wg_rec,
    const int32_t* __restrict__ wg_stride, float* __restrict__ res, const uint16_t* __restrict__ ind, const float* __restrict__ v0g,
    const float* __restrict__ v1g, int nother, const int32_t* __restrict__ rowid8, const float2* __restrict__ rpair, double* __restrict__ part,
    uint32_t part_bytes) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int nbs = (nother + 4) & ~3;
  {
    const int n4 = nother >> 2;
    for (int q = threadIdx.x; q < n4; q += blockDim.x) {
      ((f4*)ccd_lds)[q] = ((const f4*)v0g)[q];
      ((f4*)(ccd_lds + nbs))[q] = ((const f4*)v1g)[q];
    }
    for (int q = (n4 << 2) + threadIdx.x; q < nother; q += blockDim.x) {
      ccd_lds[q] = v0g[q];
      ccd_lds[nbs + q] = v1g[q];
    }
    if (threadIdx.x == 0) { ccd_lds[nother] = 0.0f; ccd_lds[nbs + nother] = 0.0f; }
    __syncthreads();
  }
  const float *b0 = ccd_lds, *b1 = ccd_lds + nbs;
  const int j = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int w = blockIdx.x, wn = wg_n[w];
  const int64_t e0 = (int64_t)wg_t0[w] * MFX_BLK_E;
  mfx_ccd_block_loop<uint16_t, true, int32_t>(rec + wg_rec[w], wn, wg_stride[w], g, res + e0, ind + e0, b1, j, part, part_bytes, b0, b1,
                                              rowid8 + e0 / MFX_BLK_EPL, rpair);
}
// (a0[r], a1[r]) side by side: what a lane of the fused sweep gathers for its row / column with ONE 8-byte load
__global__ void ccd_pairs_kernel(int64_t n, const float* __restrict__ a0, const float* __restrict__ a1, float2* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) out[r] = make_float2(a0[r], a1[r]);
}
int mfx_ccd_pairs(mfx_ctx* ctx, int64_t n, const float* a0, const float* a1, float2* out) {
  hipLaunchKernelGGL(ccd_pairs_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, ctx->stream, n, a0, a1, out);
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

// The quotients of a row pass, ONE launch: the first `dblocks` workgroups one thread per row -- u_k[row] = num / (reg + den) over the
// row's pieces in order (at most 32: 4 096 entries) --, the others one 16-lane group per LONG row: pieces lane-strided in order, then
// a fixed butterfly (a single thread walking ~400 partials of a 50 k-rating row was 13 % of a factor in round 1).
__global__ __launch_bounds__(256) void ccd_finish_kernel(const int32_t* __restrict__ rfirst, const int32_t* __restrict__ rcnt, int32_t nrows,
                                                         int dblocks, const int32_t* __restrict__ lrow, int64_t nlrow,
                                                         const double* __restrict__ part, float reg, float* __restrict__ mine) {
  if ((int)blockIdx.x < dblocks) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const int n = rcnt[r];
    if (n == 0 || n > 32) return;           // a row without ratings keeps uFac(r, k) (modelMF.cpp:1063-1065)
    const double* p = part + 2 * (int64_t)rfirst[r];
    double num = p[0], den = p[1];
    for (int s = 1; s < n; s++) { num += p[2 * s]; den += p[2 * s + 1]; }
    mine[r] = (float)(num / ((double)reg + den));
    return;
  }
  const int j = threadIdx.x & 15;
  const int64_t m = ((int64_t)((int)blockIdx.x - dblocks) * blockDim.x + threadIdx.x) >> 4;
  if (m >= nlrow) return;
  const int row = lrow[m];
  double num = 0.0, den = 0.0;
  const int first = rfirst[row], n = rcnt[row];
  for (int s = j; s < n; s += 16) {
    num += part[2 * (int64_t)(first + s)];
    den += part[2 * (int64_t)(first + s) + 1];
  }
  num = group16_sum(num);
  den = group16_sum(den);
  if (j == 0) mine[row] = (float)(num / ((double)reg + den));
}

constexpr size_t LDS_BUDGET = 160 * 1024;
static bool lds_fits(size_t bytes) { return bytes > 0 && bytes <= 150 * 1024; }
static hipError_t set_lds(mfx_ctx*, const void* fn, size_t bytes) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename T>
static int up_vec(mfx_ctx* ctx, T** dst, const std::vector<T>& v) {
  int rc = dev_alloc(ctx, dst, v.size());
  if (rc) return rc;
  if (!v.empty()) HIPCHK(hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
  return MFX_OK;
}

extern "C" int mfx_ccdpp_begin(mfx_ctx* ctx) {
  if (!ctx) return MFX_E_ARG;
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  NEED(m.present && m.has_col, MFX_E_STATE, "mfx_ccdpp_begin: train matrix with column view needed");
  NEED(ctx->U, MFX_E_STATE, "mfx_ccdpp_begin: no model");
  NEED(m.nrows <= ctx->nU && m.ncols <= ctx->nI, MFX_E_ARG, "mfx_ccdpp_begin: matrix exceeds model");
  HIPCHK(hipSetDevice(ctx->device));
  mfx_ccd_free_internal(ctx);
  int rc;
  // u_k, v_k and the pending pair: one more element each, +0.0, which masked entries of the pass loop gather
  if ((rc = dev_alloc(ctx, &ctx->uk, (size_t)ctx->nU + 1))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->vk, (size_t)ctx->nI + 1))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->uk_pend, (size_t)ctx->nU + 1))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->vk_pend, (size_t)ctx->nI + 1))) return rc;
  HIPCHK(hipMemsetAsync(ctx->uk + ctx->nU, 0, sizeof(float), ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->vk + ctx->nI, 0, sizeof(float), ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->uk_pend + ctx->nU, 0, sizeof(float), ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->vk_pend + ctx->nI, 0, sizeof(float), ctx->stream));
  // res = gk_csr_Dup(trainMat) (modelMF.cpp:1013): both value arrays.  The row view is PADDED (ccd_blocks.h): every row starts on
  // a multiple of 8 entries, the view ends on a multiple of 128 and has MFX_BLK_SLACK readable entries behind it.
  std::vector<int64_t> rowptr((size_t)m.nrows + 1, 0), rpos((size_t)m.nrows + 1, 0);
  if (m.nrows > 0) HIPCHK(hipMemcpy(rowptr.data(), m.rowptr, sizeof(int64_t) * rowptr.size(), hipMemcpyDeviceToHost));
  std::vector<MfxPiece> pieces;
  std::vector<int32_t> prow;
  pieces.reserve((size_t)m.nrows);
  prow.reserve((size_t)m.nrows);
  for (int32_t r = 0; r < m.nrows; r++) {
    const int64_t n = rowptr[(size_t)r + 1] - rowptr[(size_t)r];
    rpos[(size_t)r + 1] = rpos[(size_t)r] + ((n + MFX_BLK_EPL - 1) & ~(int64_t)(MFX_BLK_EPL - 1));
    if (n > 0) { pieces.push_back(MfxPiece{rpos[(size_t)r], rpos[(size_t)r + 1]}); prow.push_back(r); }
  }
  const int64_t nnzp = (rpos[(size_t)m.nrows] + MFX_BLK_E - 1) / MFX_BLK_E * MFX_BLK_E, nalloc = nnzp + MFX_BLK_SLACK;
  ctx->ccd_nnzp = nnzp;
  NEED(nnzp / MFX_BLK_E < ((int64_t)1 << 31), MFX_E_ARG, "mfx_ccdpp_begin: too many trips");
  const bool ids16 = lds_fits(((size_t)ctx->nI + 1) * sizeof(float)) && ctx->nI < 65536;
  if ((rc = dev_alloc(ctx, &ctx->res_row, (size_t)nalloc))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->ccd_rowid, (size_t)(nalloc / MFX_BLK_EPL)))) return rc;
  if ((rc = up_vec(ctx, &ctx->ccd_rpos, rpos))) return rc;
  if (ids16) { if ((rc = dev_alloc(ctx, &ctx->ccd_ind16, (size_t)nalloc))) return rc; }
  else if ((rc = dev_alloc(ctx, &ctx->ccd_ind32, (size_t)nalloc))) return rc;
  if (ids16)
    hipLaunchKernelGGL(rowview_build_kernel<uint16_t>, dim3(2048), dim3(256), 0, ctx->stream, m.rowptr, m.rowind, m.rowval, ctx->ccd_rpos, m.nrows,
                       ctx->nI, nnzp, nalloc, ctx->res_row, ctx->ccd_ind16, ctx->ccd_rowid);
  else
    hipLaunchKernelGGL(rowview_build_kernel<int32_t>, dim3(2048), dim3(256), 0, ctx->stream, m.rowptr, m.rowind, m.rowval, ctx->ccd_rpos, m.nrows,
                       ctx->nI, nnzp, nalloc, ctx->res_row, ctx->ccd_ind32, ctx->ccd_rowid);
  HIPCHK(hipGetLastError());
  // the column view (res->colval) lives in user-strip-major order: ccd_cols.hip
  if ((rc = mfx_ccd_cols_build(ctx))) return rc;
  // uFac.fill(0) (modelMF.cpp:1020)
  HIPCHK(hipMemsetAsync(ctx->U, 0, sizeof(float) * (size_t)ctx->nU * ctx->ld, ctx->stream));
  // the row pass: windows of the padded view, at least eight steps per group
  {
    MfxBlockPlan plan;
    std::vector<int32_t> first(pieces.size()), cnt(pieces.size());
    const int64_t ntr = nnzp / MFX_BLK_E;
    const char* we = getenv("MFX_CCD_ROW_WGS");
    const int max_wg = we && atoi(we) > 0 ? atoi(we) : 256;      // one per CU: as fast as two for the plain sweep, one round for the fused one (142 KB of LDS)
    const int nwg = (int)std::max<int64_t>(1, std::min<int64_t>((ntr + 8 * MFX_BLK_GPW - 1) / (8 * MFX_BLK_GPW), max_wg));
    NEED(mfx_blocks_region(pieces.data(), pieces.size(), 0, nnzp, nwg, 0, plan, first.data(), cnt.data()), MFX_E_STATE,
         "mfx_ccdpp_begin: the padded row view does not lay out (%lld entries, %zu rows with ratings)", (long long)nnzp, pieces.size());
    std::vector<int32_t> rfirst((size_t)m.nrows, 0), rcnt((size_t)m.nrows, 0), lrow;
    for (size_t k = 0; k < pieces.size(); k++) {
      rfirst[(size_t)prow[k]] = first[k];
      rcnt[(size_t)prow[k]] = cnt[k];
      if (cnt[k] > 32) lrow.push_back(prow[k]);
    }
    if ((rc = mfx_blocks_upload(ctx, plan, &ctx->ccd_blocks))) return rc;
    if ((rc = up_vec(ctx, &ctx->ccd_rfirst, rfirst)) || (rc = up_vec(ctx, &ctx->ccd_rcnt, rcnt)) || (rc = up_vec(ctx, &ctx->ccd_lrow, lrow))) return rc;
    ctx->ccd_nlrow = (int64_t)lrow.size();
    if (plan.nslots > ctx->ccd_part_cap) {
      dev_free(ctx->ccd_part);
      if ((rc = dev_alloc(ctx, &ctx->ccd_part, (size_t)plan.nslots * 2))) return rc;
      ctx->ccd_part_cap = plan.nslots;
    }
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->ccd_active = true;
  return MFX_OK;
}


static int run_pass(mfx_ctx* ctx, int side, float reg, float freq_thresh, int k) {
  if (side == 1) return mfx_ccd_cols_pass(ctx, ctx->uk, ctx->vk, reg, freq_thresh, k);
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const MfxBlocks& B = ctx->ccd_blocks;
  if (B.nwg > 0) {
    ProfScope ps(ctx, MFX_K_CCD_ROW);
    const size_t lds = ((size_t)ctx->nI + 1) * sizeof(float);
    const uint32_t pbytes = (uint32_t)(B.nslots * 16);
    if (ctx->ccd_ind16) {   // the gathered vector fits in LDS (items: C2 107 KB, C4 71 KB)
      HIPCHK(set_lds(ctx, (const void*)ccd_pass_kernel<true>, lds));
      hipLaunchKernelGGL(ccd_pass_kernel<true>, dim3(B.nwg), dim3(1024), lds, ctx->stream, B.rec, B.wg_t0, B.wg_n, B.wg_rec, B.wg_stride, ctx->res_row,
                         (const uint16_t*)ctx->ccd_ind16, ctx->vk, ctx->nI, ctx->ccd_part, pbytes);
    } else {
      hipLaunchKernelGGL(ccd_pass_kernel<false>, dim3(B.nwg), dim3(1024), 0, ctx->stream, B.rec, B.wg_t0, B.wg_n, B.wg_rec, B.wg_stride, ctx->res_row,
                         (const int32_t*)ctx->ccd_ind32, ctx->vk, ctx->nI, ctx->ccd_part, pbytes);
    }
    HIPCHK(hipGetLastError());
    const int dblocks = (m.nrows + 255) / 256;
    const int fblocks = (int)((ctx->ccd_nlrow * 16 + 255) / 256);
    hipLaunchKernelGGL(ccd_finish_kernel, dim3((unsigned)(dblocks + fblocks)), dim3(256), 0, ctx->stream, ctx->ccd_rfirst, ctx->ccd_rcnt, m.nrows,
                       dblocks, ctx->ccd_lrow, ctx->ccd_nlrow, ctx->ccd_part, reg, ctx->uk);
    HIPCHK(hipGetLastError());
  }
  (void)freq_thresh; (void)k;
  return MFX_OK;
}

template <int SIGN>
static int run_resid_rows(mfx_ctx* ctx, const float* uk, const float* vk) {
  const int64_t nnzp = ctx->ccd_nnzp;                     // the padded row view: its padding entries gather the +0.0 behind v_k and stay 0
  const int blocks = (int)std::min<int64_t>((nnzp / 4 + 1023) / 1024 + 1, 256 * 2);
  // row view: res_row[e] +-= u_k[rowid[e]] * v_k[rowind[e]] (v_k gathered: staged in LDS when it fits);
  // column view: res_col[e] +-= u_k[colind[e]] * v_k[colid[e]] (u_k gathered from L2)
  const size_t lds = ((size_t)ctx->nI + 1) * sizeof(float);
  if (ctx->ccd_ind16) {
    const int per_cu = (int)std::min<size_t>(2, LDS_BUDGET / lds);
    HIPCHK(set_lds(ctx, (const void*)resid_update_kernel<SIGN, true>, lds));
    hipLaunchKernelGGL((resid_update_kernel<SIGN, true>), dim3(std::min(blocks, 256 * per_cu)), dim3(1024), lds,
                       ctx->stream, ctx->res_row, ctx->ccd_rowid, (const uint16_t*)ctx->ccd_ind16, uk, vk, ctx->nI, nnzp);
  } else {
    hipLaunchKernelGGL((resid_update_kernel<SIGN, false>), dim3(blocks), dim3(1024), 0, ctx->stream, ctx->res_row,
                       ctx->ccd_rowid, ctx->ccd_ind32, uk, vk, ctx->nI, nnzp);
  }
  HIPCHK(hipGetLastError());
  return MFX_OK;
}
template <int SIGN>
static int run_resid(mfx_ctx* ctx, const float* uk, const float* vk) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  if (m.nnz == 0) return MFX_OK;
  ProfScope ps(ctx, MFX_K_CCD_RESID);
  int rc = run_resid_rows<SIGN>(ctx, uk, vk);
  if (rc) return rc;
  return mfx_ccd_cols_resid(ctx, SIGN, uk, vk, nullptr, nullptr);
}

// subtract of the pending factor fused with the add-back of the new one
static int run_resid_fused(mfx_ctx* ctx, const float* uk0, const float* vk0, const float* uk1, const float* vk1) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  if (m.nnz == 0) return MFX_OK;
  ProfScope ps(ctx, MFX_K_CCD_RESID);
  const int64_t nnzp = ctx->ccd_nnzp;
  const int blocks = (int)std::min<int64_t>((nnzp / 4 + 1023) / 1024 + 1, 256 * 2);
  const size_t lds = 2 * (((size_t)ctx->nI + 4) & ~(size_t)3) * sizeof(float);     // two vectors, each with its +0.0 slot
  if (ctx->ccd_ind16 && lds_fits(lds)) {
    HIPCHK(set_lds(ctx, (const void*)resid_fused_kernel<true>, lds));
    hipLaunchKernelGGL(resid_fused_kernel<true>, dim3(std::min(blocks, 256)), dim3(1024), lds, ctx->stream,
                       ctx->res_row, ctx->ccd_rowid, (const uint16_t*)ctx->ccd_ind16, uk0, vk0, uk1, vk1, ctx->nI, nnzp);
  } else if (ctx->ccd_ind16) {                                // two item vectors do not fit in LDS (C2: 2 x 107 KB): two sweeps, same roundings
    int rc = run_resid_rows<-1>(ctx, uk0, vk0);
    if (rc) return rc;
    if ((rc = run_resid_rows<+1>(ctx, uk1, vk1))) return rc;
  } else {
    hipLaunchKernelGGL(resid_fused_kernel<false>, dim3(blocks), dim3(1024), 0, ctx->stream, ctx->res_row, ctx->ccd_rowid,
                       ctx->ccd_ind32, uk0, vk0, uk1, vk1, ctx->nI, nnzp);
  }
  HIPCHK(hipGetLastError());
  return mfx_ccd_cols_resid(ctx, 2, uk0, vk0, uk1, vk1);
}

// The fused first sweep needs: 16-bit ids and BOTH item vectors in LDS (row view), no light region and no sharding (column view:
// ccd_cols.hip).  MFX_CCD_FUSE=0 keeps the separate update (the cross-check of the tests).
static bool can_fuse_first_sweep(mfx_ctx* ctx) {
  const char* e = getenv("MFX_CCD_FUSE");
  if (e && e[0] == '0') return false;
  const size_t lds = 2 * (((size_t)ctx->nI + 4) & ~(size_t)3) * sizeof(float);
  return ctx->ccd_ind16 && lds_fits(lds) && ctx->ccd_blocks.nwg > 0 && !mfx_sharded(ctx) && mfx_ccd_cols_can_fuse(ctx);
}
static int run_first_sweep_fused(mfx_ctx* ctx, float uReg, float iReg, float freq_thresh, int k) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const MfxBlocks& B = ctx->ccd_blocks;
  int rc;
  if (!ctx->uk_init && (rc = dev_alloc(ctx, &ctx->uk_init, (size_t)ctx->nU + 1))) return rc;
  if (!ctx->ccd_rpair && (rc = dev_alloc(ctx, &ctx->ccd_rpair, (size_t)ctx->nU + 1))) return rc;
  HIPCHK(hipMemcpyAsync(ctx->uk_init, ctx->uk, sizeof(float) * ((size_t)ctx->nU + 1), hipMemcpyDeviceToDevice, ctx->stream));
  {
    ProfScope ps(ctx, MFX_K_CCD_ROW);
    if ((rc = mfx_ccd_pairs(ctx, (int64_t)ctx->nU, ctx->uk_pend, ctx->uk, ctx->ccd_rpair))) return rc;
    const size_t lds = 2 * (((size_t)ctx->nI + 4) & ~(size_t)3) * sizeof(float);
    HIPCHK(set_lds(ctx, (const void*)ccd_pass_fused_kernel, lds));
    hipLaunchKernelGGL(ccd_pass_fused_kernel, dim3(B.nwg), dim3(1024), lds, ctx->stream, B.rec, B.wg_t0, B.wg_n, B.wg_rec, B.wg_stride, ctx->res_row,
                       (const uint16_t*)ctx->ccd_ind16, ctx->vk_pend, ctx->vk, ctx->nI, (const int32_t*)ctx->ccd_rowid, (const float2*)ctx->ccd_rpair, ctx->ccd_part,
                       (uint32_t)(B.nslots * 16));
    HIPCHK(hipGetLastError());
    const int dblocks = (m.nrows + 255) / 256;
    const int fblocks = (int)((ctx->ccd_nlrow * 16 + 255) / 256);
    hipLaunchKernelGGL(ccd_finish_kernel, dim3((unsigned)(dblocks + fblocks)), dim3(256), 0, ctx->stream, ctx->ccd_rfirst, ctx->ccd_rcnt, m.nrows,
                       dblocks, ctx->ccd_lrow, ctx->ccd_nlrow, ctx->ccd_part, uReg, ctx->uk);
    HIPCHK(hipGetLastError());
  }
  return mfx_ccd_cols_pass_fused(ctx, ctx->uk_pend, ctx->vk_pend, ctx->uk_init, ctx->vk, ctx->uk, ctx->vk, iReg, freq_thresh, k);
}

// The subtract of a finished factor is deferred so that it can share a sweep with the next add-back;
// anything that looks at the residuals (or ends the session) flushes it first.
static int ccd_flush(mfx_ctx* ctx) {
  if (!ctx->ccd_pending) return MFX_OK;
  ctx->ccd_pending = false;
  return run_resid<-1>(ctx, ctx->uk_pend, ctx->vk_pend);
}

extern "C" int mfx_ccdpp_rank1(mfx_ctx* ctx, int32_t k, int32_t inner, float uReg, float iReg, int32_t add_back,
                               float freq_thresh) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->ccd_active, MFX_E_STATE, "mfx_ccdpp_rank1: call mfx_ccdpp_begin first");
  NEED(k >= 0 && k < ctx->K, MFX_E_ARG, "mfx_ccdpp_rank1: k=%d outside [0,%d)", k, ctx->K);
  NEED(inner >= 0, MFX_E_ARG, "mfx_ccdpp_rank1: inner=%d", inner);
  HIPCHK(hipSetDevice(ctx->device));
  const int bui = (int)(((int64_t)ctx->nU + ctx->nI + 255) / 256);
  // u_k = uFac.col(k); v_k = iFac.col(k)  (modelMF.cpp:1028-1029)
  hipLaunchKernelGGL(extract_col_kernel, dim3(bui), dim3(256), 0, ctx->stream, ctx->U, ctx->nU, ctx->V, ctx->nI, ctx->ld, k, ctx->uk, ctx->vk);
  HIPCHK(hipGetLastError());
  int rc;
  int it0 = 0;
  if (ctx->ccd_pending && add_back && inner >= 1 && can_fuse_first_sweep(ctx)) {
    // previous factor's :1095-1116 + this one's :1032-1056 applied ON THE WAY of the first sweep (:1058-1092, it = 0): each view is
    // read once instead of twice.  The add-back takes u_k, v_k as extracted -- the row pass moves u_k on before the column view is updated.
    ctx->ccd_pending = false;
    if ((rc = run_first_sweep_fused(ctx, uReg, iReg, freq_thresh, k))) return rc;
    it0 = 1;
  } else if (ctx->ccd_pending && add_back) {                // previous factor's :1095-1116 + this one's :1032-1056
    ctx->ccd_pending = false;
    if ((rc = run_resid_fused(ctx, ctx->uk_pend, ctx->vk_pend, ctx->uk, ctx->vk))) return rc;
  } else {
    if ((rc = ccd_flush(ctx))) return rc;
    if (add_back && (rc = run_resid<+1>(ctx, ctx->uk, ctx->vk))) return rc;   // :1032-1056
  }
  for (int it = it0; it < inner; it++) {                    // :1058-1092
    if ((rc = run_pass(ctx, 0, uReg, -1.0f, k))) return rc;
    if ((rc = run_pass(ctx, 1, iReg, freq_thresh, k))) return rc;
  }
  // uFac.col(k) = u_k; iFac.col(k) = v_k  (:1119-1120)
  hipLaunchKernelGGL(store_col_kernel, dim3(bui), dim3(256), 0, ctx->stream, ctx->U, ctx->nU, ctx->V, ctx->nI, ctx->ld, k, ctx->uk, ctx->vk);
  HIPCHK(hipGetLastError());
  // res -= u_k v_k^T (:1095-1116) is deferred: keep (u_k, v_k) aside
  std::swap(ctx->uk, ctx->uk_pend);
  std::swap(ctx->vk, ctx->vk_pend);
  ctx->ccd_pending = true;
  return MFX_OK;
}

extern "C" int mfx_ccdpp_end(mfx_ctx* ctx) {
  if (!ctx) return MFX_E_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->ccd_active) { int rc = ccd_flush(ctx); if (rc) return rc; }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  mfx_ccd_free_internal(ctx);
  return MFX_OK;
}

extern "C" int mfx_debug_residuals(mfx_ctx* ctx, float* res_row, float* res_col) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->ccd_active, MFX_E_STATE, "mfx_debug_residuals: CCD++ not active");
  HIPCHK(hipSetDevice(ctx->device));
  { int rc = ccd_flush(ctx); if (rc) return rc; }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const size_t nnz = (size_t)m.nnz;
  if (nnz == 0) return MFX_OK;
  if (res_row) {                      // the padded view back in CSR order
    float* tmp = nullptr;
    int rc = dev_alloc(ctx, &tmp, nnz);
    if (rc) return rc;
    hipLaunchKernelGGL(rowview_export_kernel, dim3(2048), dim3(256), 0, ctx->stream, m.rowptr, ctx->ccd_rpos, m.nrows, ctx->res_row, tmp);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(res_row, tmp, sizeof(float) * nnz, hipMemcpyDeviceToHost);
    dev_free(tmp);
    if (e != hipSuccess) return mfx_fail(ctx, MFX_E_HIP, "mfx_debug_residuals: %s", hipGetErrorString(e));
  }
  if (res_col) return mfx_ccd_cols_export(ctx, res_col);
  return MFX_OK;
}
