// ccd_cols.hip -- the column view of CCD++ with the user vector staged in LDS.
//
// The column sweep v_k[i] = sum_u res_ui u_k[u] / (iReg + sum_u u_k[u]^2) (modelMF.cpp:1078-1090) and the
// column-view residual updates (:1045-1055, :1106-1116) gather u_k[u] for the users of a column.  From L2
// every 4-byte gather moves a 128-byte line, which made these kernels L2-bandwidth-bound (measured: the
// same residual kernel runs 2.7x faster on the row view, where the gathered v_k sits in LDS).  u_k does not fit in
// LDS (C4: 1.9 MB), so the column view is stored in USER-BLOCK-major order: block b holds, column by
// column, the entries whose user lies in [b*UB, (b+1)*UB), users ascending -- i.e. the reference's
// colind/colval (gk_csr_CreateIndex order) cut into UB-user strips.  A workgroup stages u_k[b*UB ...] (32 KB)
// in LDS and works on entries of that strip only.  Column sums are formed per (strip, column) segment of <= 1024
// entries and finished per column in a fixed order (strip-major), so every sum has a fixed association.
//
// LIGHT columns (at most LIGHT entries in the whole column: at the Netflix shape two thirds of the items, 4 % of the
// entries, and two thirds of the 1 M (strip, column) segments, each 1..16 entries long -- the pass spent its time on
// per-segment latency, not on bytes) stay OUT of the strip scheme: their entries are kept contiguous behind the strips,
// users ascending (the reference's CSC order), one segment per column, and u_k is gathered from L2 for them.
// (That was round 2's first answer to the short pieces; with the trip lists the strips handle them better, and the default
// threshold is 0 -- every column goes through the strips.  The region stays as the MFX_CCD_LIGHT knob and in the tests.)
// Both regions are worked through as trip lists (mfx_internal.h) by ONE launch: the light workgroups first, then the
// strips' workgroups in proportion to their trips.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "mfx_internal.h"

namespace {
constexpr int UB = 8192;          // users per strip: 32 KB of u_k (two vectors fit for the fused update)
constexpr int CSEG = 1024;        // entries per segment
constexpr int LIGHT = 0;          // columns with at most this many entries are handled whole, outside the strips (MFX_CCD_LIGHT).
                                  // 0 since the trip lists: short (strip, column) pieces no longer cost per-segment latency, and the
                                  // light region's L2 gathers do (C4: 2.29 vs 2.40 ms per factor at 1024, C2 shape 0.86 vs 0.89)
constexpr int GPW = 64;            // 16-lane groups per pass workgroup (1024 threads)
constexpr int COL_TRIP_E = 64;     // entries per trip of the column view: 4 per lane (mfx_internal.h)
constexpr int PASS_WGS = 1024;    // pass workgroups over all strips (two rounds of the 512 resident ones; MFX_CCD_PASS_WGS)
constexpr int64_t ENT_PER_WG = 128 * 1024;

struct ColState {
  int nb = 0;                       // strips
  int64_t nnz = 0;
  int32_t* off = nullptr;           // [nI][nb+1] CSC position where strip b starts inside column i (relative to colptr[i])
  int64_t* dst = nullptr;           // [nb][nI] blocked position of segment (b,i)
  // Index widths (the kernels stream these arrays, so their width is a share of the time): a strip-local user id is below
  // UB = 8192 -> 16 bits, always; the column id is 16 bits when there are at most 65 536 columns; the light region keeps absolute
  // 32-bit user ids in its own array (entry t of the blocked order sits at luser[t - (light0 & ~63)]: 16-byte loads stay aligned).
  uint16_t* buser = nullptr;        // blocked, strips only: user - b*UB
  int32_t* luser = nullptr;         // light region: absolute user id
  uint16_t* bcol16 = nullptr;       // blocked: column id (ncols <= 65536) ...
  int32_t* bcol32 = nullptr;        // ... or 32-bit
  float* res = nullptr;             // blocked residual (the reference's res->colval in strip-major order)
  // pass segments
  int64_t* seg_beg = nullptr; int64_t* seg_end = nullptr; int32_t* seg_col = nullptr;
  int64_t nseg = 0;
  double* part = nullptr;           // [nseg][2]
  double* sums = nullptr;           // sharded runs: [ncols][2] (num, den) for the all-reduce
  int32_t* col_ptr = nullptr;       // [nI+1] column -> its segments (strip-major order)
  int32_t* col_seg = nullptr;
  // pass: trip list (mfx_ccd_trip_loop), one range per group; the groups of the light region first, then the strips'
  // workgroups (GPW groups each, one strip each)
  MfxTrips trips; int32_t* gptr = nullptr;
  int32_t* pw_blk = nullptr; int npw = 0;      // strip of each pass workgroup
  int ngl = 0;                                 // groups of the light region (they come first, then the npw * GPW strip groups)
  int32_t* rw_blk = nullptr; int64_t* rw_e0 = nullptr; int64_t* rw_e1 = nullptr; int nrw = 0;       // residual
  int64_t light0 = 0;               // blocked position where the light columns start (== nnz: none)
  int32_t lseg0 = 0, nlseg = 0;     // their segments (one per column) in the segment tables
};
ColState* st(mfx_ctx* ctx) { return (ColState*)ctx->ccd_cols; }

template <typename T>
int up(mfx_ctx* ctx, T** dst, const std::vector<T>& v) {
  int rc = dev_alloc(ctx, dst, v.size());
  if (rc) return rc;
  if (!v.empty()) HIPCHK(hipMemcpyAsync(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, ctx->stream));
  return MFX_OK;
}
}  // namespace

void mfx_ccd_cols_free(mfx_ctx* ctx) {
  ColState* s = st(ctx);
  if (!s) return;
  dev_free(s->off); dev_free(s->dst); dev_free(s->buser); dev_free(s->luser); dev_free(s->bcol16); dev_free(s->bcol32); dev_free(s->res);
  dev_free(s->seg_beg); dev_free(s->seg_end); dev_free(s->seg_col); dev_free(s->part); dev_free(s->sums);
  dev_free(s->col_ptr); dev_free(s->col_seg);
  dev_free(s->pw_blk); mfx_trips_free(s->trips); dev_free(s->gptr);
  dev_free(s->rw_blk); dev_free(s->rw_e0); dev_free(s->rw_e1);
  delete s;
  ctx->ccd_cols = nullptr;
}

// off[i][b] = first entry of column i whose user is >= b*UB (users ascend inside a column)
__global__ void strip_offsets_kernel(const int64_t* __restrict__ colptr, const int32_t* __restrict__ colind,
                                     int32_t ncols, int nb, int32_t* __restrict__ off) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)ncols * (nb + 1)) return;
  const int i = (int)(t / (nb + 1)), b = (int)(t % (nb + 1));
  const int64_t beg = colptr[i], end = colptr[i + 1];
  int64_t lo = beg, hi = end;
  const int64_t key = (int64_t)b * UB;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (colind[mid] < key) lo = mid + 1; else hi = mid;
  }
  off[t] = (int32_t)(lo - beg);
}

// copy the (strip, column) pieces into strip-major order
template <typename ColT>
__global__ __launch_bounds__(256) void strip_scatter_kernel(const int64_t* __restrict__ colptr,
                                                            const int32_t* __restrict__ colind,
                                                            const float* __restrict__ colval,
                                                            const int32_t* __restrict__ off,
                                                            const int64_t* __restrict__ dst, int32_t ncols, int nb,
                                                            int64_t light0, uint16_t* __restrict__ buser,
                                                            int32_t* __restrict__ luser, ColT* __restrict__ bcol,
                                                            float* __restrict__ res) {
  const int j = threadIdx.x & 15;
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t p = grp; p < (int64_t)ncols * nb; p += ngrp) {
    const int b = (int)(p / ncols), i = (int)(p % ncols);
    const int64_t src = colptr[i] + off[(int64_t)i * (nb + 1) + b];
    const int64_t n = off[(int64_t)i * (nb + 1) + b + 1] - off[(int64_t)i * (nb + 1) + b];
    const int64_t d = dst[(int64_t)b * ncols + i];
    const bool is_light = d >= light0;                // light columns keep the absolute user id (u_k comes from L2)
    for (int64_t t = j; t < n; t += 16) {
      if (is_light) luser[d + t - (light0 & ~(int64_t)63)] = colind[src + t];
      else buser[d + t] = (uint16_t)(colind[src + t] - b * UB);
      bcol[d + t] = (ColT)i;
      res[d + t] = colval[src + t];
    }
  }
}

// inverse of the scatter for the test hook: strip-major residual -> CSC order
__global__ __launch_bounds__(256) void strip_gather_kernel(const int64_t* __restrict__ colptr,
                                                           const int32_t* __restrict__ off,
                                                           const int64_t* __restrict__ dst, int32_t ncols, int nb,
                                                           const float* __restrict__ res, float* __restrict__ out) {
  const int j = threadIdx.x & 15;
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t p = grp; p < (int64_t)ncols * nb; p += ngrp) {
    const int b = (int)(p / ncols), i = (int)(p % ncols);
    const int64_t src = colptr[i] + off[(int64_t)i * (nb + 1) + b];
    const int64_t n = off[(int64_t)i * (nb + 1) + b + 1] - off[(int64_t)i * (nb + 1) + b];
    const int64_t d = dst[(int64_t)b * ncols + i];
    for (int64_t t = j; t < n; t += 16) out[src + t] = res[d + t];
  }
}

int mfx_ccd_cols_build(mfx_ctx* ctx) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  mfx_ccd_cols_free(ctx);
  ColState* s = new ColState;
  ctx->ccd_cols = s;
  const int32_t nI = m.ncols;
  const int nb = std::max(1, (m.nrows + UB - 1) / UB);
  s->nb = nb;
  s->nnz = m.nnz;
  int rc;
  if ((rc = dev_alloc(ctx, &s->off, (size_t)nI * (nb + 1)))) return rc;
  {
    const int64_t n = (int64_t)nI * (nb + 1);
    hipLaunchKernelGGL(strip_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, m.colptr,
                       m.colind, nI, nb, s->off);
    HIPCHK(hipGetLastError());
  }
  std::vector<int32_t> off((size_t)nI * (nb + 1));
  HIPCHK(hipMemcpyAsync(off.data(), s->off, sizeof(int32_t) * off.size(), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  // strip-major positions, segments, per-column segment lists, workgroup tables
  std::vector<int64_t> dst((size_t)nb * nI), seg_beg, seg_end, rw_e0, rw_e1;
  std::vector<int32_t> seg_col, pw_blk, rw_blk, strip_seg0;
  std::vector<int32_t> col_cnt((size_t)nI, 0);
  std::vector<uint8_t> light((size_t)nI, 0);
  const char* le = getenv("MFX_CCD_LIGHT");        // experiment / test knob: the threshold (0: every column goes through the strips)
  const int light_max = le ? atoi(le) : LIGHT;
  for (int32_t i = 0; i < nI; i++) light[(size_t)i] = off[(size_t)i * (nb + 1) + nb] <= light_max;
  int64_t pos = 0;
  for (int b = 0; b < nb; b++) {
    const int64_t ent0 = pos;
    const int32_t sg0 = (int32_t)seg_col.size();
    for (int32_t i = 0; i < nI; i++) {
      if (light[(size_t)i]) continue;
      const int64_t n = off[(size_t)i * (nb + 1) + b + 1] - off[(size_t)i * (nb + 1) + b];
      dst[(size_t)b * nI + i] = pos;
      for (int64_t c = 0; c < n; c += CSEG) {
        seg_beg.push_back(pos + c);
        seg_end.push_back(pos + std::min<int64_t>(n, c + CSEG));
        seg_col.push_back(i);
        col_cnt[i]++;
      }
      pos += n;
    }
    const int32_t sg1 = (int32_t)seg_col.size();
    strip_seg0.push_back(sg0);
    (void)sg1;
    for (int64_t a = ent0; a < pos; a += ENT_PER_WG) { rw_blk.push_back(b); rw_e0.push_back(a); rw_e1.push_back(std::min(pos, a + ENT_PER_WG)); }
  }
  // the light columns behind the strips: whole columns, one segment each; piece (b, i) sits at its CSC offset
  s->light0 = pos;
  s->lseg0 = (int32_t)seg_col.size();
  for (int32_t i = 0; i < nI; i++) {
    if (!light[(size_t)i]) continue;
    const int64_t n = off[(size_t)i * (nb + 1) + nb];
    for (int b = 0; b < nb; b++) dst[(size_t)b * nI + i] = pos + off[(size_t)i * (nb + 1) + b];
    for (int64_t c = 0; c < n; c += CSEG) {        // (longer than CSEG only under MFX_CCD_LIGHT: a trip record holds 11 bits of length)
      seg_beg.push_back(pos + c);
      seg_end.push_back(pos + std::min<int64_t>(n, c + CSEG));
      seg_col.push_back(i);
      col_cnt[i]++;
    }
    pos += n;
  }
  s->nlseg = (int32_t)seg_col.size() - s->lseg0;
  strip_seg0.push_back(s->lseg0);
  // trips: strip by strip (workgroups of one strip each), then the light columns; meta = the segment
  std::vector<int4> trips;
  std::vector<int32_t> gptr;
  {
    std::vector<MfxSeg> segs(seg_col.size());
    std::vector<int64_t> cum(seg_col.size() + 1, 0);
    for (size_t k = 0; k < seg_col.size(); k++) {
      segs[k] = MfxSeg{seg_beg[k], seg_end[k], (int32_t)k};
      cum[k + 1] = cum[k] + mfx_seg_trips(segs[k], COL_TRIP_E);
    }
    if (cum.back() >= ((int64_t)1 << 31)) return mfx_fail(ctx, MFX_E_ARG, "mfx_ccdpp_begin: too many trips in the column view");
    trips.reserve((size_t)cum.back());
    const char* pe = getenv("MFX_CCD_PASS_WGS");
    const int want_wgs = pe && atoi(pe) > 0 ? atoi(pe) : PASS_WGS;
    const int64_t strip_trips = cum[(size_t)s->lseg0];
    const int64_t per_wg = std::max<int64_t>(8 * GPW, (strip_trips + want_wgs - 1) / want_wgs);     // at least eight trips per group
    const int64_t light_trips = cum.back() - strip_trips;
    const int lwg = (int)std::min<int64_t>((light_trips + 8 * GPW - 1) / (8 * GPW), 512);
    s->ngl = lwg * GPW;
    int64_t strip_end = 0, light_end = 0;      // positions behind the last entry the trips of the two regions load
    if (lwg > 0) mfx_trips_layout(segs, (size_t)s->lseg0, seg_col.size(), lwg, GPW, COL_TRIP_E, trips, gptr, &light_end);
    for (int b = 0; b < nb; b++) {
      const size_t k0 = (size_t)strip_seg0[(size_t)b], k1 = (size_t)strip_seg0[(size_t)b + 1];
      const int64_t tb = cum[k1] - cum[k0];
      if (tb == 0) continue;
      const int nw = (int)std::max<int64_t>(1, (tb + per_wg / 2) / per_wg);
      mfx_trips_layout(segs, k0, k1, nw, GPW, COL_TRIP_E, trips, gptr, &strip_end);
      for (int w = 0; w < nw; w++) pw_blk.push_back(b);
    }
    gptr.push_back((int32_t)trips.size());
    // whole trips are loaded: strips from res (float[nnz]) and buser (uint16[light0 + 4]), the light region from res and
    // luser (int32, shifted by light0 & ~63): every one of them must lie inside its allocation (mfx_trips_fit)
    static_assert((COL_TRIP_E - 1) * sizeof(float) <= MFX_ALLOC_PAD, "a column-view trip must fit the allocation pad");
    if (!(mfx_trips_fit(std::max(strip_end, light_end), m.nnz, sizeof(float)) && mfx_trips_fit(strip_end, s->light0 + 4, sizeof(uint16_t)) &&
          mfx_trips_fit(light_end, m.nnz + 4, sizeof(int32_t))))
      return mfx_fail(ctx, MFX_E_STATE, "mfx_ccdpp_begin: a trip of the column view reads behind its arrays (strips to %lld of %lld, light "
                      "columns to %lld of %lld; allocation pad %zu bytes)", (long long)strip_end, (long long)s->light0, (long long)light_end,
                      (long long)m.nnz, MFX_ALLOC_PAD);
  }
  std::vector<int32_t> col_ptr((size_t)nI + 1, 0), col_seg(seg_col.size());
  for (int32_t i = 0; i < nI; i++) col_ptr[i + 1] = col_ptr[i] + col_cnt[i];
  {
    std::vector<int32_t> w(col_ptr.begin(), col_ptr.end() - 1);
    for (size_t k = 0; k < seg_col.size(); k++) col_seg[w[seg_col[k]]++] = (int32_t)k;   // strip-major inside a column
  }
  if ((rc = up(ctx, &s->dst, dst))) return rc;
  if ((rc = up(ctx, &s->seg_beg, seg_beg))) return rc;
  if ((rc = up(ctx, &s->seg_end, seg_end))) return rc;
  if ((rc = up(ctx, &s->seg_col, seg_col))) return rc;
  if ((rc = up(ctx, &s->col_ptr, col_ptr))) return rc;
  if ((rc = up(ctx, &s->col_seg, col_seg))) return rc;
  if ((rc = up(ctx, &s->pw_blk, pw_blk))) return rc;
  if ((rc = mfx_trips_upload(ctx, trips, &s->trips))) return rc;
  if ((rc = up(ctx, &s->gptr, gptr))) return rc;
  if ((rc = up(ctx, &s->rw_blk, rw_blk))) return rc;
  if ((rc = up(ctx, &s->rw_e0, rw_e0))) return rc;
  if ((rc = up(ctx, &s->rw_e1, rw_e1))) return rc;
  s->nseg = (int64_t)seg_col.size();
  s->npw = (int)pw_blk.size();
  s->nrw = (int)rw_blk.size();
  if ((rc = dev_alloc(ctx, &s->part, (size_t)s->nseg * 2))) return rc;
  const bool col16 = nI <= 65536;
  if ((rc = dev_alloc(ctx, &s->buser, (size_t)s->light0 + 4))) return rc;      // (+ MFX_ALLOC_PAD: the last strip trips run past light0)
  if ((rc = dev_alloc(ctx, &s->luser, (size_t)(m.nnz - (s->light0 & ~(int64_t)63)) + 4))) return rc;
  if (col16) { if ((rc = dev_alloc(ctx, &s->bcol16, (size_t)m.nnz + 4))) return rc; }
  else if ((rc = dev_alloc(ctx, &s->bcol32, (size_t)m.nnz + 4))) return rc;
  if ((rc = dev_alloc(ctx, &s->res, (size_t)m.nnz))) return rc;
  HIPCHK(hipMemsetAsync(s->res + m.nnz, 0, MFX_ALLOC_PAD, ctx->stream));    // read (masked) by the pass loop: finite
  if (m.nnz > 0) {
    if (col16)
      hipLaunchKernelGGL(strip_scatter_kernel<uint16_t>, dim3(2048), dim3(256), 0, ctx->stream, m.colptr, m.colind, m.colval,
                         s->off, s->dst, nI, nb, s->light0, s->buser, s->luser, s->bcol16, s->res);
    else
      hipLaunchKernelGGL(strip_scatter_kernel<int32_t>, dim3(2048), dim3(256), 0, ctx->stream, m.colptr, m.colind, m.colval,
                         s->off, s->dst, nI, nb, s->light0, s->buser, s->luser, s->bcol32, s->res);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MFX_OK;
}

__device__ __forceinline__ void stage_strip(float* lds, const float* __restrict__ v, int first, int n) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  // first is a multiple of UB, so the source is 16-byte aligned
  const int n4 = n >> 2;
  for (int q = threadIdx.x; q < n4; q += blockDim.x) ((f4*)lds)[q] = ((const f4*)(v + first))[q];
  for (int q = (n4 << 2) + threadIdx.x; q < n; q += blockDim.x) lds[q] = v[first + q];
}

__device__ __forceinline__ double g16_sum(double v) { return mfx_row16_sum(v); }

// column pass: one 16-lane group per range of the trip list.  Workgroups [0, nlw): the light columns, u_k gathered from L2 by
// the absolute user id; the others: one strip each, the strip of u_k in LDS and strip-local 16-bit user ids.  ONE launch, the
// latency-bound light part (4 % of the entries, but 42 us on its own at C4) first, so that it runs under the strips.
template <bool BUF>
__global__ __launch_bounds__(1024) void colpass_kernel(const MfxTrips trips, const int32_t* __restrict__ gptr,
                                                       const int32_t* __restrict__ pw_blk, int nlw, int g_count,
                                                       const float* __restrict__ res, uint32_t res_bytes, const uint16_t* __restrict__ buser,
                                                       const int32_t* __restrict__ luser, const float* __restrict__ uk, int nU_strips,
                                                       int nU, double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float su[UB + 4];
  const bool strip = (int)blockIdx.x >= nlw;       // the light workgroups come first: they start first and run under the strips
  if (strip) {
    const int b = pw_blk[(int)blockIdx.x - nlw];
    stage_strip(su, uk, b * UB, min(UB, nU_strips - b * UB));
    if (threadIdx.x == 0) su[UB] = 0.0f;          // what masked entries gather
    __syncthreads();
  }
  const int j = threadIdx.x & 15;
  const int g = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  if (g >= g_count) return;
  if (strip) mfx_ccd_trip_loop<BUF, COL_TRIP_E / 16>(trips, gptr[g], gptr[g + 1], res, buser, res_bytes, su, UB, j, part);
  else mfx_ccd_trip_loop<BUF, COL_TRIP_E / 16>(trips, gptr[g], gptr[g + 1], res, luser, res_bytes, uk, nU, j, part);      // uk[nU] = +0.0 (mfx_ccdpp_begin)
}

// residual update of the light region (MODE as colresid_kernel): u_k from L2
template <int MODE, typename ColT>
__global__ __launch_bounds__(256) void colresid_light_kernel(int64_t e0, int64_t e1, float* __restrict__ res, const int32_t* __restrict__ luser,
                                                             int64_t lshift, const ColT* __restrict__ bcol, const float* __restrict__ uk0,
                                                             const float* __restrict__ vk0, const float* __restrict__ uk1,
                                                             const float* __restrict__ vk1) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < e1; t += stride) {
    const int u = luser[t - lshift], c = (int)bcol[t];
    const float p0 = uk0[u] * vk0[c];
    float r = res[t];
    if (MODE == 1) r = r + p0;
    else r = r - p0;
    if (MODE == 2) r = r + uk1[u] * vk1[c];
    res[t] = r;
  }
}

// v_k[i] from the column's segment partials in strip-major order; one 16-lane group per column.
// SUMS (sharded runs): only emit (num, den) of this rank's users; coldivide_kernel finishes after the all-reduce.
template <bool SUMS>
__global__ __launch_bounds__(256) void colfinish_kernel(const int32_t* __restrict__ col_ptr,
                                                        const int32_t* __restrict__ col_seg,
                                                        const double* __restrict__ part, int32_t ncols, float reg,
                                                        float* __restrict__ vk, const int64_t* __restrict__ colptr,
                                                        float freq_thresh, int k, double* __restrict__ sums) {
  const int j = threadIdx.x & 15;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  if (i >= ncols) return;
  const int a = col_ptr[i], e = col_ptr[i + 1];
  if (a == e) {         // an item without train ratings is invalid: v_k keeps iFac(i,k) (modelMF.cpp:1079-1081)
    if (SUMS && j == 0) { sums[2 * i] = 0.0; sums[2 * i + 1] = 0.0; }
    return;
  }
  double num = 0.0, den = 0.0;
  for (int t = a + j; t < e; t += 16) {   // lane-strided in list order, then a fixed butterfly
    num += part[2 * (int64_t)col_seg[t]];
    den += part[2 * (int64_t)col_seg[t] + 1];
  }
  num = g16_sum(num);
  den = g16_sum(den);
  if (j == 0) {
    if (SUMS) { sums[2 * i] = num; sums[2 * i + 1] = den; return; }
    float v = (float)(num / ((double)reg + den));
    if (freq_thresh >= 0.0f) {  // modelMF.cpp:1336-1342
      const double freq = (double)(colptr[i + 1] - colptr[i]);
      if (freq < (double)freq_thresh && k > 0) v = 0.0f;
    }
    vk[i] = v;
  }
}
// sharded runs: v_k[i] from the sums over all ranks; gcol = ratings per item over all ranks
__global__ void coldivide_kernel(const double* __restrict__ sums, const double* __restrict__ gcol, int32_t ncols, float reg,
                                 float* __restrict__ vk, float freq_thresh, int k) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncols || gcol[i] == 0.0) return;
  float v = (float)(sums[2 * i] / ((double)reg + sums[2 * i + 1]));
  if (freq_thresh >= 0.0f && gcol[i] < (double)freq_thresh && k > 0) v = 0.0f;
  vk[i] = v;
}

// residual update on the strip-major column view.  MODE +1: res += u0*v0, -1: res -= u0*v0,
// 2: res = (res - u0*v0) + u1*v1 (deferred subtract fused with the next add-back)
template <int MODE, typename ColT>
__global__ __launch_bounds__(1024) void colresid_kernel(const int32_t* __restrict__ rw_blk,
                                                        const int64_t* __restrict__ rw_e0,
                                                        const int64_t* __restrict__ rw_e1, float* __restrict__ res,
                                                        const uint16_t* __restrict__ buser,
                                                        const ColT* __restrict__ bcol,
                                                        const float* __restrict__ uk0, const float* __restrict__ vk0,
                                                        const float* __restrict__ uk1, const float* __restrict__ vk1,
                                                        int nU) {
  __shared__ __attribute__((aligned(16))) float su[(MODE == 2 ? 2 : 1) * UB];
  const int b = rw_blk[blockIdx.x];
  const int n = min(UB, nU - b * UB);
  stage_strip(su, uk0, b * UB, n);
  if (MODE == 2) stage_strip(su + UB, uk1, b * UB, n);
  __syncthreads();
  // 16 aligned bytes per lane and array (4 entries); the entries of a neighbouring workgroup's range are left alone
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef int i4 __attribute__((ext_vector_type(4)));
  const int64_t e0 = rw_e0[blockIdx.x], e1 = rw_e1[blockIdx.x];
  for (int64_t t = (e0 & ~(int64_t)3) + 4 * (int64_t)threadIdx.x; t < e1; t += 4 * (int64_t)blockDim.x) {
    if (t >= e0 && t + 4 <= e1) {
      const i4 lu = MfxCcdTrip::load4(buser + t), c = MfxCcdTrip::load4(bcol + t);
      f4 r = *(const f4*)(res + t);
      // the column ids of four consecutive entries ascend (column by column inside a strip): two gathers serve a quad that
      // lies in one column or two, an entry in a third column takes its own (see gather_sorted4 in ccd.hip)
      float v0[4], v1[4];
      v0[0] = vk0[c[0]]; v0[3] = vk0[c[3]];
      v0[1] = c[1] == c[0] ? v0[0] : v0[3]; v0[2] = c[2] == c[0] ? v0[0] : v0[3];
      if (c[1] != c[0] && c[1] != c[3]) v0[1] = vk0[c[1]];
      if (c[2] != c[0] && c[2] != c[3]) v0[2] = vk0[c[2]];
      if (MODE == 2) {
        v1[0] = vk1[c[0]]; v1[3] = vk1[c[3]];
        v1[1] = c[1] == c[0] ? v1[0] : v1[3]; v1[2] = c[2] == c[0] ? v1[0] : v1[3];
        if (c[1] != c[0] && c[1] != c[3]) v1[1] = vk1[c[1]];
        if (c[2] != c[0] && c[2] != c[3]) v1[2] = vk1[c[2]];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float p0 = su[lu[q]] * v0[q];   // u_k(u)*v_k(item): float product (modelMF.cpp:1053, :1114)
        if (MODE == 1) r[q] = r[q] + p0;
        else r[q] = r[q] - p0;
        if (MODE == 2) r[q] = r[q] + su[UB + lu[q]] * v1[q];
      }
      *(f4*)(res + t) = r;
    } else {
      for (int q = 0; q < 4; q++) {
        const int64_t tt = t + q;
        if (tt < e0 || tt >= e1) continue;
        const int lu = (int)buser[tt], c = (int)bcol[tt];
        const float p0 = su[lu] * vk0[c];
        float r = res[tt];
        if (MODE == 1) r = r + p0;
        else r = r - p0;
        if (MODE == 2) r = r + su[UB + lu] * vk1[c];
        res[tt] = r;
      }
    }
  }
}

int mfx_ccd_cols_pass(mfx_ctx* ctx, const float* uk, float* vk, float reg, float freq_thresh, int k) {
  ColState* s = st(ctx);
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  if (s->npw > 0 || s->ngl > 0) {
    ProfScope ps(ctx, MFX_K_CCD_COL);
    // entry t of the blocked order sits at luser[t - (light0 & ~63)]
    const uint64_t rbytes = ((uint64_t)s->nnz + 64) * 4;
#define MFX_COLPASS(BF)                                                                                                              \
  hipLaunchKernelGGL(colpass_kernel<BF>, dim3(s->npw + s->ngl / GPW), dim3(16 * GPW), 0, ctx->stream, s->trips, s->gptr, s->pw_blk, s->ngl / GPW, \
                     s->npw * GPW + s->ngl, s->res, (uint32_t)rbytes, (const uint16_t*)s->buser,                                      \
                     (const int32_t*)(s->luser - (s->light0 & ~(int64_t)63)), uk, m.nrows, ctx->nU, s->part)
    if (rbytes < ((uint64_t)1 << 32) && !getenv("MFX_CCD_NOBUF")) MFX_COLPASS(true); else MFX_COLPASS(false);
#undef MFX_COLPASS
    const unsigned fb = (unsigned)(((int64_t)m.ncols * 16 + 255) / 256);
    if (!mfx_sharded(ctx)) {
      hipLaunchKernelGGL(colfinish_kernel<false>, dim3(fb), dim3(256), 0, ctx->stream, s->col_ptr, s->col_seg, s->part,
                         m.ncols, reg, vk, m.colptr, freq_thresh, k, (double*)nullptr);
      HIPCHK(hipGetLastError());
      return MFX_OK;
    }
  }
  if (mfx_sharded(ctx) && m.ncols > 0) {
    // the users of one item live on several ranks: sum (num, den) over the ranks, then divide everywhere
    int rc;
    if (!s->sums && (rc = dev_alloc(ctx, &s->sums, (size_t)m.ncols * 2))) return rc;
    const double* gcol;
    if ((rc = mfx_comm_global_col_counts(ctx, &gcol))) return rc;
    if (s->npw > 0 || s->nlseg > 0) {
      hipLaunchKernelGGL(colfinish_kernel<true>, dim3((unsigned)(((int64_t)m.ncols * 16 + 255) / 256)), dim3(256), 0, ctx->stream,
                         s->col_ptr, s->col_seg, s->part, m.ncols, reg, vk, m.colptr, freq_thresh, k, s->sums);
      HIPCHK(hipGetLastError());
    } else {
      HIPCHK(hipMemsetAsync(s->sums, 0, sizeof(double) * 2 * (size_t)m.ncols, ctx->stream));
    }
    if ((rc = mfx_comm_allreduce(ctx, s->sums, (size_t)m.ncols * 2, 1))) return rc;
    hipLaunchKernelGGL(coldivide_kernel, dim3((m.ncols + 255) / 256), dim3(256), 0, ctx->stream, s->sums, gcol, m.ncols, reg, vk,
                       freq_thresh, k);
    HIPCHK(hipGetLastError());
  }
  return MFX_OK;
}

int mfx_ccd_cols_resid(mfx_ctx* ctx, int mode, const float* uk0, const float* vk0, const float* uk1,
                       const float* vk1) {
  ColState* s = st(ctx);
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  if (s->nrw > 0) {
#define MFX_CR(MD)                                                                                                                 \
  do {                                                                                                                             \
    if (s->bcol16)                                                                                                                 \
      hipLaunchKernelGGL((colresid_kernel<MD, uint16_t>), dim3(s->nrw), dim3(1024), 0, ctx->stream, s->rw_blk, s->rw_e0, s->rw_e1, \
                         s->res, s->buser, s->bcol16, uk0, vk0, uk1, vk1, m.nrows);                                                \
    else                                                                                                                           \
      hipLaunchKernelGGL((colresid_kernel<MD, int32_t>), dim3(s->nrw), dim3(1024), 0, ctx->stream, s->rw_blk, s->rw_e0, s->rw_e1,  \
                         s->res, s->buser, s->bcol32, uk0, vk0, uk1, vk1, m.nrows);                                                \
  } while (0)
    if (mode == 1) MFX_CR(1); else if (mode == 2) MFX_CR(2); else MFX_CR(-1);
#undef MFX_CR
  }
  if (s->light0 < s->nnz) {
    const unsigned lb = (unsigned)std::min<int64_t>((s->nnz - s->light0 + 255) / 256, 4096);
    const int64_t lsh = s->light0 & ~(int64_t)63;
#define MFX_CL(MD)                                                                                                                \
  do {                                                                                                                            \
    if (s->bcol16)                                                                                                                \
      hipLaunchKernelGGL((colresid_light_kernel<MD, uint16_t>), dim3(lb), dim3(256), 0, ctx->stream, s->light0, s->nnz, s->res,   \
                         s->luser, lsh, s->bcol16, uk0, vk0, uk1, vk1);                                                           \
    else                                                                                                                          \
      hipLaunchKernelGGL((colresid_light_kernel<MD, int32_t>), dim3(lb), dim3(256), 0, ctx->stream, s->light0, s->nnz, s->res,    \
                         s->luser, lsh, s->bcol32, uk0, vk0, uk1, vk1);                                                           \
  } while (0)
    if (mode == 1) MFX_CL(1); else if (mode == 2) MFX_CL(2); else MFX_CL(-1);
#undef MFX_CL
  }
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

// test hook: the residual in the reference's CSC order
int mfx_ccd_cols_export(mfx_ctx* ctx, float* host_out) {
  ColState* s = st(ctx);
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  if (m.nnz == 0) return MFX_OK;
  float* tmp = nullptr;
  int rc = dev_alloc(ctx, &tmp, (size_t)m.nnz);
  if (rc) return rc;
  hipLaunchKernelGGL(strip_gather_kernel, dim3(2048), dim3(256), 0, ctx->stream, m.colptr, s->off, s->dst, m.ncols,
                     s->nb, s->res, tmp);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipMemcpy(host_out, tmp, sizeof(float) * (size_t)m.nnz, hipMemcpyDeviceToHost);
  dev_free(tmp);
  if (e != hipSuccess) return mfx_fail(ctx, MFX_E_HIP, "mfx_ccd_cols_export: %s", hipGetErrorString(e));
  return MFX_OK;
}
