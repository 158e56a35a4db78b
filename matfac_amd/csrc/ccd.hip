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
// in double, the quotient is rounded to float once.  A 16-lane group reduces one row
// segment (<= MFX_SEG ratings) at a time, working through its range of the TRIP LIST
// (mfx_internal.h: mfx_ccd_trip_loop -- 128-entry trips, records fetched four at a time,
// data prefetched two trips ahead, whole trips loaded and masked); the sums of a segment
// go to a slot and the quotients are taken by the finishing kernels in a fixed order, so
// every sum has a fixed association (reproducible).  Item ids are 16 bits wherever v_k
// is staged in LDS.
#include <algorithm>

#include "mfx_internal.h"

void mfx_ccd_free_internal(mfx_ctx* ctx) {
  dev_free(ctx->res_row); dev_free(ctx->res_col); dev_free(ctx->uk); dev_free(ctx->vk);
  dev_free(ctx->uk_pend); dev_free(ctx->vk_pend);
  ctx->ccd_pending = false;
  dev_free(ctx->ccd_part); dev_free(ctx->colid); dev_free(ctx->ccd_ind16); mfx_trips_free(ctx->ccd_trips); dev_free(ctx->ccd_gptr); dev_free(ctx->ccd_single);
  ctx->ccd_nsingle = 0;
  ctx->ccd_ngroups = 0;
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

// res[e] (+/-)= a[rowOf[e]] * b[colOf[e]]   -- float product, then float add/sub.
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
__global__ void narrow_ids_kernel(const int32_t* __restrict__ src, int64_t n, uint16_t* __restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) dst[t] = (uint16_t)src[t];
}

// a[x[0..3]] for four CONSECUTIVE entries of the row view: their rows ascend, and with ~200 entries per row a quad almost always
// lies in one row or two.  Two gathers (first and last entry) serve it then; an entry whose row is neither (rows of one or two
// entries) takes its own.  (A 4-byte gather occupies the vector memory pipe like a 16-byte load: four of them per quad were
// half of this kernel's memory instructions.)
typedef int mfx_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gather_sorted4(const float* __restrict__ a, mfx_i4 x, float (&o)[4]) {
  o[0] = a[x[0]];
  o[3] = a[x[3]];
  o[1] = x[1] == x[0] ? o[0] : o[3];
  o[2] = x[2] == x[0] ? o[0] : o[3];
  if (x[1] != x[0] && x[1] != x[3]) o[1] = a[x[1]];
  if (x[2] != x[0] && x[2] != x[3]) o[2] = a[x[2]];
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
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
    const i4 x = ((const i4*)ia)[q], y = MfxCcdTrip::load4(ib + 4 * q);
    f4 r = ((const f4*)res)[q];
    float au[4];
    gather_sorted4(a, x, au);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float prod = au[e] * b[y[e]];
      r[e] = SIGN > 0 ? r[e] + prod : r[e] - prod;
    }
    ((f4*)res)[q] = r;
  }
  for (int64_t e = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
    const float prod = a[ia[e]] * b[ib[e]];
    res[e] = SIGN > 0 ? res[e] + prod : res[e] - prod;
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
  if (LDSB) {
    const int n4 = nb >> 2;
    for (int q = threadIdx.x; q < n4; q += blockDim.x) {
      ((f4*)ccd_lds)[q] = ((const f4*)b0g)[q];
      ((f4*)(ccd_lds + ((nb + 3) & ~3)))[q] = ((const f4*)b1g)[q];
    }
    for (int q = (n4 << 2) + threadIdx.x; q < nb; q += blockDim.x) {
      ccd_lds[q] = b0g[q];
      ccd_lds[((nb + 3) & ~3) + q] = b1g[q];
    }
    __syncthreads();
  }
  const float* b0 = LDSB ? ccd_lds : b0g;
  const float* b1 = LDSB ? ccd_lds + ((nb + 3) & ~3) : b1g;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
    const i4 x = ((const i4*)ia)[q], y = MfxCcdTrip::load4(ib + 4 * q);
    f4 r = ((const f4*)res)[q];
    float au0[4], au1[4];
    gather_sorted4(a0, x, au0);
    gather_sorted4(a1, x, au1);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float p0 = au0[e] * b0[y[e]];
      const float p1 = au1[e] * b1[y[e]];
      r[e] = (r[e] - p0) + p1;
    }
    ((f4*)res)[q] = r;
  }
  for (int64_t e = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride)
    res[e] = (res[e] - a0[ia[e]] * b0[ib[e]]) + a1[ia[e]] * b1[ib[e]];
}

__device__ __forceinline__ double group16_sum(double v) { return mfx_row16_sum(v); }

// Row pass: one 16-lane group per range of the trip list (mfx_ccd_trip_loop).  meta of a trip = the slot of the segment's
// (num, den) pair: rows with several segments first (RowSegs::seg_slab), then one slot per single-segment row.  The quotient
// is taken by the finishing kernels: a double division is ~35 instructions, and with four groups per wavefront three steps out
// of four end a segment somewhere in the wave -- it was a quarter of the pass.  (trainCCDPPFreqAdap's rule applies to items
// only: modelMF.cpp:1336-1342 -- the row pass never sees a threshold.)
constexpr int ROW_TRIP_E = 128;      // entries per trip of the row view: 8 per lane (mfx_internal.h)
template <bool LDSO, bool BUF>
__global__ __launch_bounds__(1024) void ccd_pass_kernel(const MfxTrips trips, const int32_t* __restrict__ gptr, int ngroups,
                                                       const float* __restrict__ res,
                                                       const typename ItemIdx<LDSO>::type* __restrict__ ind, uint32_t res_bytes,
                                                       const float* __restrict__ otherg, int nother, double* __restrict__ part) {
  if (LDSO) stage_vector(otherg, nother);
  const float* other = LDSO ? ccd_lds : otherg;
  const int j = threadIdx.x & 15;
  const int grp = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  if (grp >= ngroups) return;
  // (the global vectors of a CCD++ session carry one more element, +0.0: mfx_ccdpp_begin)
  mfx_ccd_trip_loop<BUF, ROW_TRIP_E / 16>(trips, gptr[grp], gptr[grp + 1], res, ind, res_bytes, other, nother, j, part);
}

// The quotients of a row pass, ONE launch: the first `dblocks` workgroups take the single-segment rows (u_k[row] = num / (reg +
// den) from the row's slot, one thread each); the others the rows with several segments -- partials summed by a 16-lane
// group, lane-strided in segment order then a fixed butterfly (a single thread walking ~100 partials made this 57 us launch
// 13 % of a factor).
__global__ __launch_bounds__(256) void ccd_finish_kernel(const int32_t* __restrict__ single, int64_t nsingle, const double* __restrict__ spart,
                                                         int dblocks, const int32_t* __restrict__ mrow,
                                                         const int32_t* __restrict__ mrow_first,
                                                         const int32_t* __restrict__ mrow_n, int64_t nmrow,
                                                         const double* __restrict__ part, float reg,
                                                         float* __restrict__ mine, const int64_t* __restrict__ ptr,
                                                         float freq_thresh, int k) {
  if ((int)blockIdx.x < dblocks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nsingle) mine[single[i]] = (float)(spart[2 * i] / ((double)reg + spart[2 * i + 1]));
    return;
  }
  const int j = threadIdx.x & 15;
  const int64_t m = ((int64_t)((int)blockIdx.x - dblocks) * blockDim.x + threadIdx.x) >> 4;
  if (m >= nmrow) return;
  double num = 0.0, den = 0.0;
  const int first = mrow_first[m], n = mrow_n[m];
  for (int s = j; s < n; s += 16) {
    num += part[2 * (int64_t)(first + s)];
    den += part[2 * (int64_t)(first + s) + 1];
  }
  num = group16_sum(num);
  den = group16_sum(den);
  if (j != 0) return;
  const int row = mrow[m];
  float v = (float)(num / ((double)reg + den));
  if (freq_thresh >= 0.0f) {
    const double freq = (double)(ptr[row + 1] - ptr[row]);
    if (freq < (double)freq_thresh && k > 0) v = 0.0f;
  }
  mine[row] = v;
}

constexpr size_t LDS_BUDGET = 160 * 1024;
static bool lds_fits(size_t bytes) { return bytes > 0 && bytes <= 150 * 1024; }
static hipError_t set_lds(mfx_ctx*, const void* fn, size_t bytes) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
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
  const size_t nnz = (size_t)m.nnz;
  if ((rc = dev_alloc(ctx, &ctx->res_row, nnz))) return rc;
  // u_k, v_k and the pending pair: one more element each, +0.0, which masked entries of the pass loop gather
  if ((rc = dev_alloc(ctx, &ctx->uk, (size_t)ctx->nU + 1))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->vk, (size_t)ctx->nI + 1))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->uk_pend, (size_t)ctx->nU + 1))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->vk_pend, (size_t)ctx->nI + 1))) return rc;
  HIPCHK(hipMemsetAsync(ctx->uk + ctx->nU, 0, sizeof(float), ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->vk + ctx->nI, 0, sizeof(float), ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->uk_pend + ctx->nU, 0, sizeof(float), ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->vk_pend + ctx->nI, 0, sizeof(float), ctx->stream));
  // res = gk_csr_Dup(trainMat) (modelMF.cpp:1013): both value arrays
  if (nnz) {
    HIPCHK(hipMemcpyAsync(ctx->res_row, m.rowval, sizeof(float) * nnz, hipMemcpyDeviceToDevice, ctx->stream));
  }
  HIPCHK(hipMemsetAsync(ctx->res_row + nnz, 0, MFX_ALLOC_PAD, ctx->stream));   // the pad behind the residuals is read (masked) by the pass loop: finite
  if (nnz && lds_fits(((size_t)ctx->nI + 1) * sizeof(float))) {
    if ((rc = dev_alloc(ctx, &ctx->ccd_ind16, nnz))) return rc;
    hipLaunchKernelGGL(narrow_ids_kernel, dim3(2048), dim3(256), 0, ctx->stream, m.rowind, (int64_t)nnz, ctx->ccd_ind16);
    HIPCHK(hipGetLastError());
  }
  // the column view (res->colval) lives in user-strip-major order: ccd_cols.hip
  if ((rc = mfx_ccd_cols_build(ctx))) return rc;
  // uFac.fill(0) (modelMF.cpp:1020)
  HIPCHK(hipMemsetAsync(ctx->U, 0, sizeof(float) * (size_t)ctx->nU * ctx->ld, ctx->stream));
  RowSegs* sg;
  if ((rc = mfx_get_segments(ctx, 0, &sg))) return rc;
  // the row view's trip list: segments in memory order (a trip's aligned 128 entries overlap its neighbours' -- adjacent in
  // time they hit in L2), cut into one range per group of the launch
  if (sg->nseg > 0) {
    std::vector<int64_t> sb((size_t)sg->nseg), se((size_t)sg->nseg);
    std::vector<int32_t> srow((size_t)sg->nseg), sslab((size_t)sg->nseg);
    HIPCHK(hipMemcpyAsync(sb.data(), sg->seg_beg, sizeof(int64_t) * sb.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(se.data(), sg->seg_end, sizeof(int64_t) * se.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(srow.data(), sg->seg_row, sizeof(int32_t) * srow.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(sslab.data(), sg->seg_slab, sizeof(int32_t) * sslab.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<int32_t> order((size_t)sg->nseg);
    for (size_t q = 0; q < order.size(); q++) order[q] = (int32_t)q;
    std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return sb[x] < sb[y]; });
    std::vector<int4> trips;
    std::vector<int32_t> gptr, single;
    std::vector<MfxSeg> segs;
    segs.reserve(order.size());
    int64_t ntrips = 0;
    for (int32_t q : order) {
      NEED(se[q] - sb[q] <= MFX_SEG, MFX_E_STATE, "mfx_ccdpp_begin: a segment longer than %d", MFX_SEG);
      int32_t slot = sslab[q];
      if (slot < 0) { slot = (int32_t)(sg->nslab + (int64_t)single.size()); single.push_back(srow[q]); }
      segs.push_back(MfxSeg{sb[q], se[q], slot});
      ntrips += mfx_seg_trips(segs.back(), ROW_TRIP_E);
    }
    NEED(sg->nslab + (int64_t)single.size() < ((int64_t)1 << 31), MFX_E_ARG, "mfx_ccdpp_begin: too many segments");
    if ((rc = dev_alloc(ctx, &ctx->ccd_single, single.size()))) return rc;
    if (!single.empty()) HIPCHK(hipMemcpy(ctx->ccd_single, single.data(), sizeof(int32_t) * single.size(), hipMemcpyHostToDevice));
    ctx->ccd_nsingle = (int64_t)single.size();
    if (sg->nslab + ctx->ccd_nsingle > ctx->ccd_part_cap) {
      dev_free(ctx->ccd_part);
      if ((rc = dev_alloc(ctx, &ctx->ccd_part, (size_t)(sg->nslab + ctx->ccd_nsingle) * 2))) return rc;
      ctx->ccd_part_cap = sg->nslab + ctx->ccd_nsingle;
    }
    NEED(ntrips < ((int64_t)1 << 31), MFX_E_ARG, "mfx_ccdpp_begin: too many trips");
    const char* we = getenv("MFX_CCD_ROW_WGS");
    const int max_wg = we && atoi(we) > 0 ? atoi(we) : 512;
    const int nwg = (int)std::max<int64_t>(1, std::min<int64_t>((ntrips + 8 * 64 - 1) / (8 * 64), max_wg));   // at least eight trips per group
    const int ng = nwg * 64;
    trips.reserve((size_t)ntrips);
    int64_t max_end = 0;
    mfx_trips_layout(segs, 0, segs.size(), nwg, 64, ROW_TRIP_E, trips, gptr, &max_end);
    gptr.push_back((int32_t)trips.size());
    // whole trips are loaded from the residuals (4 bytes per entry) and from the ids (2 or 4): they must lie inside the allocations
    static_assert((ROW_TRIP_E - 1) * sizeof(float) <= MFX_ALLOC_PAD, "a row-view trip must fit the allocation pad");
    NEED(mfx_trips_fit(max_end, m.nnz, sizeof(float)) && mfx_trips_fit(max_end, m.nnz, sizeof(int32_t)), MFX_E_STATE,
         "mfx_ccdpp_begin: a trip of the row view reads %lld entries behind the %lld of its arrays (allocation pad %zu bytes)",
         (long long)(max_end - m.nnz), (long long)m.nnz, MFX_ALLOC_PAD);
    if ((rc = mfx_trips_upload(ctx, trips, &ctx->ccd_trips))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->ccd_gptr, gptr.size()))) return rc;
    HIPCHK(hipMemcpy(ctx->ccd_gptr, gptr.data(), sizeof(int32_t) * gptr.size(), hipMemcpyHostToDevice));
    ctx->ccd_ngroups = ng;
  }
  ctx->ccd_active = true;
  return MFX_OK;
}


static int run_pass(mfx_ctx* ctx, int side, float reg, float freq_thresh, int k) {
  if (side == 1) return mfx_ccd_cols_pass(ctx, ctx->uk, ctx->vk, reg, freq_thresh, k);
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  RowSegs* sg;
  int rc = mfx_get_segments(ctx, side, &sg);
  if (rc) return rc;
  const float* res = side == 0 ? ctx->res_row : ctx->res_col;
  const int32_t* ind = side == 0 ? m.rowind : m.colind;
  const float* other = side == 0 ? ctx->vk : ctx->uk;
  float* mine = side == 0 ? ctx->uk : ctx->vk;
  const int64_t* ptr = side == 0 ? m.rowptr : m.colptr;
  const int nother = side == 0 ? ctx->nI : ctx->nU;
  if (sg->nseg > 0) {
    ProfScope ps(ctx, MFX_K_CCD_ROW);
    const size_t lds = ((size_t)nother + 1) * sizeof(float);
    const int blocks = (ctx->ccd_ngroups + 63) / 64;
    const uint64_t rbytes = ((uint64_t)m.nnz + 64) * 4;             // buffer loads (lanes outside a segment skip their access) below 4 GB
    const bool buf = rbytes < ((uint64_t)1 << 32) && !getenv("MFX_CCD_NOBUF");
#define MFX_ROWPASS(LD, BF, LDSB, IND)                                                                                          \
  hipLaunchKernelGGL((ccd_pass_kernel<LD, BF>), dim3(blocks), dim3(1024), LDSB, ctx->stream, ctx->ccd_trips, ctx->ccd_gptr, ctx->ccd_ngroups, \
                     res, IND, (uint32_t)rbytes, other, nother, ctx->ccd_part)
    if (lds_fits(lds)) {   // the gathered vector fits in LDS (items: C2 107 KB, C4 71 KB)
      HIPCHK(set_lds(ctx, (const void*)ccd_pass_kernel<true, true>, lds));
      HIPCHK(set_lds(ctx, (const void*)ccd_pass_kernel<true, false>, lds));
      if (buf) MFX_ROWPASS(true, true, lds, (const uint16_t*)ctx->ccd_ind16);
      else MFX_ROWPASS(true, false, lds, (const uint16_t*)ctx->ccd_ind16);
    } else {
      if (buf) MFX_ROWPASS(false, true, 0, ind);
      else MFX_ROWPASS(false, false, 0, ind);
    }
#undef MFX_ROWPASS
    HIPCHK(hipGetLastError());
  }
  if (ctx->ccd_nsingle > 0 || sg->nmrow > 0) {
    const int dblocks = (int)((ctx->ccd_nsingle + 255) / 256);
    const int fblocks = (int)((sg->nmrow * 16 + 255) / 256);
    hipLaunchKernelGGL(ccd_finish_kernel, dim3((unsigned)(dblocks + fblocks)), dim3(256), 0, ctx->stream, ctx->ccd_single, ctx->ccd_nsingle,
                       ctx->ccd_part + 2 * sg->nslab, dblocks, sg->mrow, sg->mrow_first, sg->mrow_n, sg->nmrow, ctx->ccd_part, reg, mine, ptr,
                       freq_thresh, k);
    HIPCHK(hipGetLastError());
  }
  return MFX_OK;
}

template <int SIGN>
static int run_resid(mfx_ctx* ctx, const float* uk, const float* vk) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  if (m.nnz == 0) return MFX_OK;
  ProfScope ps(ctx, MFX_K_CCD_RESID);
  const int blocks = (int)std::min<int64_t>((m.nnz / 4 + 1023) / 1024 + 1, 256 * 2);
  // row view: res_row[e] +-= u_k[rowid[e]] * v_k[rowind[e]] (v_k gathered: staged in LDS when it fits);
  // column view: res_col[e] +-= u_k[colind[e]] * v_k[colid[e]] (u_k gathered from L2)
  const size_t lds = ((size_t)ctx->nI + 1) * sizeof(float);
  if (lds_fits(lds)) {
    const int per_cu = (int)std::min<size_t>(2, LDS_BUDGET / lds);
    HIPCHK(set_lds(ctx, (const void*)resid_update_kernel<SIGN, true>, lds));
    hipLaunchKernelGGL((resid_update_kernel<SIGN, true>), dim3(std::min(blocks, 256 * per_cu)), dim3(1024), lds,
                       ctx->stream, ctx->res_row, m.rowid, (const uint16_t*)ctx->ccd_ind16, uk, vk, ctx->nI, m.nnz);
  } else {
    hipLaunchKernelGGL((resid_update_kernel<SIGN, false>), dim3(blocks), dim3(1024), 0, ctx->stream, ctx->res_row,
                       m.rowid, m.rowind, uk, vk, ctx->nI, m.nnz);
  }
  HIPCHK(hipGetLastError());
  return mfx_ccd_cols_resid(ctx, SIGN, uk, vk, nullptr, nullptr);
}

// subtract of the pending factor fused with the add-back of the new one
static int run_resid_fused(mfx_ctx* ctx, const float* uk0, const float* vk0, const float* uk1, const float* vk1) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  if (m.nnz == 0) return MFX_OK;
  ProfScope ps(ctx, MFX_K_CCD_RESID);
  const int blocks = (int)std::min<int64_t>((m.nnz / 4 + 1023) / 1024 + 1, 256 * 2);
  const size_t lds = 2 * (((size_t)ctx->nI + 3) & ~(size_t)3) * sizeof(float);
  if (lds_fits(lds)) {
    HIPCHK(set_lds(ctx, (const void*)resid_fused_kernel<true>, lds));
    hipLaunchKernelGGL(resid_fused_kernel<true>, dim3(std::min(blocks, 256)), dim3(1024), lds, ctx->stream,
                       ctx->res_row, m.rowid, (const uint16_t*)ctx->ccd_ind16, uk0, vk0, uk1, vk1, ctx->nI, m.nnz);
  } else {
    hipLaunchKernelGGL(resid_fused_kernel<false>, dim3(blocks), dim3(1024), 0, ctx->stream, ctx->res_row, m.rowid,
                       m.rowind, uk0, vk0, uk1, vk1, ctx->nI, m.nnz);
  }
  HIPCHK(hipGetLastError());
  return mfx_ccd_cols_resid(ctx, 2, uk0, vk0, uk1, vk1);
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
  if (ctx->ccd_pending && add_back) {                       // previous factor's :1095-1116 + this one's :1032-1056
    ctx->ccd_pending = false;
    if ((rc = run_resid_fused(ctx, ctx->uk_pend, ctx->vk_pend, ctx->uk, ctx->vk))) return rc;
  } else {
    if ((rc = ccd_flush(ctx))) return rc;
    if (add_back && (rc = run_resid<+1>(ctx, ctx->uk, ctx->vk))) return rc;   // :1032-1056
  }
  for (int it = 0; it < inner; it++) {                      // :1058-1092
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
  const size_t nnz = (size_t)ctx->mat[MFX_MAT_TRAIN].nnz;
  if (nnz == 0) return MFX_OK;
  if (res_row) HIPCHK(hipMemcpy(res_row, ctx->res_row, sizeof(float) * nnz, hipMemcpyDeviceToHost));
  if (res_col) return mfx_ccd_cols_export(ctx, res_col);
  return MFX_OK;
}
