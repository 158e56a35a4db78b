// ccd_cols.hip -- the column view of CCD++ with the user vector staged in LDS.
//
// The column sweep v_k[i] = sum_u res_ui u_k[u] / (iReg + sum_u u_k[u]^2) (modelMF.cpp:1078-1090) and the
// column-view residual updates (:1045-1055, :1106-1116) gather u_k[u] for the users of a column.  From L2
// every 4-byte gather moves a 128-byte line, which made these kernels L2-bandwidth-bound (measured: the
// same residual kernel runs 2.7x faster on the row view, where the gathered v_k sits in LDS).  u_k does not fit in
// LDS (C4: 1.9 MB), so the column view is stored in USER-BLOCK-major order: block b holds, column by
// column, the entries whose user lies in [b*UB, (b+1)*UB), users ascending -- i.e. the reference's
// colind/colval (gk_csr_CreateIndex order) cut into UB-user strips.  A workgroup stages u_k[b*UB ...] (32 KB)
// in LDS and works on entries of that strip only.  The view is PADDED (ccd_blocks.h: every (strip, column) piece starts on a
// multiple of 8 entries, every strip on a multiple of 128); column sums are formed per piece and 128-entry trip and finished per
// column in a fixed order (strip-major), so every sum has a fixed association.
//
// LIGHT columns (at most LIGHT entries in the whole column: at the Netflix shape two thirds of the items, 4 % of the
// entries, and two thirds of the 1 M (strip, column) segments, each 1..16 entries long -- the pass spent its time on
// per-segment latency, not on bytes) stay OUT of the strip scheme: their entries are kept contiguous behind the strips,
// users ascending (the reference's CSC order), one segment per column, and u_k is gathered from L2 for them.
// (That was round 2's first answer to the short pieces; the strips handle them better since, and the default
// threshold is 0 -- every column goes through the strips.  The region stays as the MFX_CCD_LIGHT knob and in the tests.)
// Both regions are worked through by ONE launch of the block loop (ccd_blocks.h): the light workgroups first, then the
// strips' workgroups in proportion to their trips.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "ccd_blocks.h"

namespace {
constexpr int UB = 8192;          // users per strip: 32 KB of u_k (two vectors fit for the fused update)
constexpr int LIGHT = 0;          // columns with at most this many entries are handled whole, outside the strips (MFX_CCD_LIGHT).
                                  // 0 since the trip lists: short (strip, column) pieces no longer cost per-segment latency, and the
                                  // light region's L2 gathers do (C4: 2.29 vs 2.40 ms per factor at 1024, C2 shape 0.86 vs 0.89)
constexpr int GPW = MFX_BLK_GPW;   // 16-lane groups per pass workgroup (1024 threads)
constexpr int PASS_WGS = 256;     // pass workgroups over all strips (MFX_CCD_PASS_WGS): one per CU -- as fast as 512 for the plain sweep, one round for the fused one

struct ColState {
  int nb = 0;                       // strips
  int64_t nnz = 0;                  // entries of the matrix
  int64_t nnzp = 0;                 // entries of the PADDED strip-major view (ccd_blocks.h): every (strip, column) piece starts on a multiple
                                    // of 8, every strip (and the light region) on a multiple of 128
  int32_t* off = nullptr;           // [nI][nb+1] CSC position where strip b starts inside column i (relative to colptr[i])
  int64_t* dst = nullptr;           // [nb][nI] padded position of piece (b,i)
  int32_t* plen = nullptr;          // [nb][nI] its padded length (light columns: of the whole column, at strip 0; 0 elsewhere)
  // Index widths (the kernels stream these arrays, so their width is a share of the time): a strip-local user id is below
  // UB = 8192 -> 16 bits, always (padding entries: UB, the +0.0 slot behind the staged strip); the column id is 16 bits when there are
  // at most 65 536 columns; the light region keeps absolute 32-bit user ids in its own array (entry t of the padded order sits at
  // luser[t - light0]; padding entries: nU, the +0.0 behind u_k).
  uint16_t* buser = nullptr;        // strips only: user - b*UB
  int32_t* luser = nullptr;         // light region: absolute user id
  uint16_t* bcol16 = nullptr;       // column id of every EIGHT entries (pieces start on multiples of 8), 16-bit for ncols <= 65536 ...
  int32_t* bcol32 = nullptr;        // ... or 32-bit
  float* res = nullptr;             // residual (the reference's res->colval in padded strip-major order)
  double* part = nullptr;           // [slots][2]
  double* sums = nullptr;           // sharded runs: [ncols][2] (num, den) for the all-reduce
  int32_t* col_ptr = nullptr;       // [nI+1] column -> its slots (strip-major order)
  int32_t* col_seg = nullptr;
  float2* cpair = nullptr;          // fused first sweep: (v_pend[i], v_k[i]) per item
  int32_t* fin_order = nullptr;     // the columns by team size of colfinish_kernel: n16 with at most 64 slots, n64 with at most 512, the rest
  int fin_n16 = 0, fin_n64 = 0, fin_n256 = 0;
  MfxBlocks blocks;                 // the pass: the light region's workgroups first (tag -1), then the strips' (tag = strip)
  int nlw = 0;                      // workgroups of the light region
  int32_t* rw_blk = nullptr; int64_t* rw_e0 = nullptr; int64_t* rw_e1 = nullptr; int64_t* rw_stride = nullptr; int nrw = 0;       // residual: strip, first piece, end of the strip, entries to the workgroup's next piece
  int64_t light0 = 0, light1 = 0;   // padded positions of the light region (light0 == light1: none)
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
  dev_free(s->off); dev_free(s->dst); dev_free(s->plen); dev_free(s->buser); dev_free(s->luser); dev_free(s->bcol16); dev_free(s->bcol32); dev_free(s->res);
  dev_free(s->part); dev_free(s->sums);
  dev_free(s->col_ptr); dev_free(s->col_seg); dev_free(s->fin_order); dev_free(s->cpair);
  mfx_blocks_free(s->blocks);
  dev_free(s->rw_blk); dev_free(s->rw_e0); dev_free(s->rw_e1); dev_free(s->rw_stride);
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

// copy the (strip, column) pieces into the padded strip-major order; the padding entries behind a piece: residual 0, the piece's
// column, and the user id that gathers +0.0 (UB in a strip, nU in the light region)
template <typename ColT>
__global__ __launch_bounds__(256) void strip_scatter_kernel(const int64_t* __restrict__ colptr,
                                                            const int32_t* __restrict__ colind,
                                                            const float* __restrict__ colval,
                                                            const int32_t* __restrict__ off,
                                                            const int64_t* __restrict__ dst, const int32_t* __restrict__ plen,
                                                            int32_t ncols, int nb, int nU,
                                                            int64_t light0, int64_t light1, uint16_t* __restrict__ buser,
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
    const bool is_light = d >= light0 && d < light1;      // light columns keep the absolute user id (u_k comes from L2)
    const int64_t np = plen[(int64_t)b * ncols + i];      // (light: the padding of the WHOLE column hangs on its last strip's piece)
    for (int64_t t = j; t < (np > n ? np : n); t += 16) {
      const bool real = t < n;
      if (is_light) luser[d + t - light0] = real ? colind[src + t] : nU;
      else buser[d + t] = (uint16_t)(real ? colind[src + t] - b * UB : UB);
      if (((d + t) & 7) == 0) bcol[(d + t) >> 3] = (ColT)i;
      res[mfx_blk_mem_of(d + t)] = real ? colval[src + t] : 0.0f;
    }
  }
}
// p[a[g] .. b[g]) = value for a list of ranges: what no piece owns (the tails of the regions, the slack behind the view)
template <typename T>
__global__ void fill_ranges_kernel(T* __restrict__ p, const int64_t* __restrict__ a, const int64_t* __restrict__ b, int n, T value) {
  for (int g = blockIdx.y; g < n; g += gridDim.y)
    for (int64_t t = a[g] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < b[g]; t += (int64_t)gridDim.x * blockDim.x) p[t] = value;
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
    for (int64_t t = j; t < n; t += 16) out[src + t] = res[mfx_blk_mem_of(d + t)];
  }
}

namespace {
struct Ranges { std::vector<int64_t> a, b; void add(int64_t x, int64_t y) { if (y > x) { a.push_back(x); b.push_back(y); } } };
template <typename T>
int fill_ranges(mfx_ctx* ctx, T* p, const Ranges& r, T value) {
  if (r.a.empty()) return MFX_OK;
  int64_t *da = nullptr, *db = nullptr;
  int rc;
  if ((rc = up(ctx, &da, r.a)) || (rc = up(ctx, &db, r.b))) { dev_free(da); dev_free(db); return rc; }
  hipLaunchKernelGGL(fill_ranges_kernel<T>, dim3(64, (unsigned)std::min<size_t>(r.a.size(), 1024)), dim3(256), 0, ctx->stream, p, da, db, (int)r.a.size(), value);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  dev_free(da); dev_free(db);
  if (e != hipSuccess) return mfx_fail(ctx, MFX_E_HIP, "mfx_ccdpp_begin: %s", hipGetErrorString(e));
  return MFX_OK;
}
int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
}  // namespace

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
  // padded strip-major positions, the pieces of every region, per-column slot lists, workgroup tables
  std::vector<int64_t> dst((size_t)nb * nI, 0), rw_e0, rw_e1, rw_stride;
  std::vector<int32_t> plen((size_t)nb * nI, 0), rw_blk;
  std::vector<uint8_t> light((size_t)nI, 0);
  const char* le = getenv("MFX_CCD_LIGHT");        // experiment / test knob: the threshold (0: every column goes through the strips)
  const int light_max = le ? atoi(le) : LIGHT;
  for (int32_t i = 0; i < nI; i++) light[(size_t)i] = off[(size_t)i * (nb + 1) + nb] <= light_max;
  // residual update: a workgroup works on ONE strip (it stages that strip of u_k), the workgroups of a strip take its 4 096-entry
  // pieces round-robin.  About ONE workgroup per CU over all strips (MFX_CCD_RESID_WGS, default 256): measured at C4 per launch of
  // colresid_kernel<2> -- 128: 385 us, 192: 250, 256: 215, 320: 275, 512: 234, 1 024 / 2 048: 240; the 813 contiguous 128 K-entry
  // ranges of rounds 2 and 3: 245)
  int64_t ent_per_wg = 128 * 1024;
  {
    const char* re = getenv("MFX_CCD_RESID_WGS");
    const int64_t slots = re && atoi(re) > 0 ? atoi(re) : 256;
    const int64_t est = m.nnz + 8 * (int64_t)nI * nb / 2 + 128 * (int64_t)nb;          // entries with their padding, roughly
    if (slots > nb) ent_per_wg = std::max<int64_t>(ent_per_wg, round_up((est + (slots - nb) - 1) / (slots - nb), 4096));
  }
  struct Region { int64_t r0, r1; int tag; std::vector<MfxPiece> pc; std::vector<int32_t> col; };
  std::vector<Region> regions;
  Ranges tails;                                     // positions inside the view that no piece owns
  int64_t pos = 0;
  for (int b = 0; b < nb; b++) {
    Region R;
    R.r0 = pos;
    R.tag = b;
    for (int32_t i = 0; i < nI; i++) {
      if (light[(size_t)i]) continue;
      const int64_t n = off[(size_t)i * (nb + 1) + b + 1] - off[(size_t)i * (nb + 1) + b];
      const int64_t np = round_up(n, MFX_BLK_EPL);
      dst[(size_t)b * nI + i] = pos;
      plen[(size_t)b * nI + i] = (int32_t)np;
      if (n > 0) { R.pc.push_back(MfxPiece{pos, pos + np}); R.col.push_back(i); }
      pos += np;
    }
    R.r1 = round_up(pos, MFX_BLK_E);
    tails.add(pos, R.r1);
    {   // the strip's workgroups take its 4 096-entry pieces (one trip of a workgroup's loop) round-robin: one front per strip
      const int64_t nwb = (R.r1 - R.r0 + ent_per_wg - 1) / ent_per_wg;
      for (int64_t k = 0; k < nwb; k++) { rw_blk.push_back(b); rw_e0.push_back(R.r0 + 4096 * k); rw_e1.push_back(R.r1); rw_stride.push_back(4096 * nwb); }
    }
    pos = R.r1;
    regions.push_back(std::move(R));
  }
  // the light columns behind the strips: whole columns, one piece each; piece (b, i) sits at its CSC offset inside the column
  s->light0 = pos;
  {
    Region R;
    R.r0 = pos;
    R.tag = -1;
    for (int32_t i = 0; i < nI; i++) {
      if (!light[(size_t)i]) continue;
      const int64_t n = off[(size_t)i * (nb + 1) + nb];
      const int64_t np = round_up(n, MFX_BLK_EPL);
      for (int b = 0; b < nb; b++) {
        dst[(size_t)b * nI + i] = pos + off[(size_t)i * (nb + 1) + b];
        plen[(size_t)b * nI + i] = off[(size_t)i * (nb + 1) + b + 1] - off[(size_t)i * (nb + 1) + b] + (b == nb - 1 ? (int32_t)(np - n) : 0);
      }
      if (n > 0) { R.pc.push_back(MfxPiece{pos, pos + np}); R.col.push_back(i); }
      pos += np;
    }
    R.r1 = round_up(pos, MFX_BLK_E);
    tails.add(pos, R.r1);
    pos = R.r1;
    s->light1 = pos;
    regions.push_back(std::move(R));
  }
  const int64_t nnzp = pos, nalloc = nnzp + MFX_BLK_SLACK;
  s->nnzp = nnzp;
  if (nnzp / MFX_BLK_E >= ((int64_t)1 << 31)) return mfx_fail(ctx, MFX_E_ARG, "mfx_ccdpp_begin: too many trips in the column view");
  // the pass: the light region's workgroups first (latency-bound: u_k from L2), then the strips' in proportion to their trips
  MfxBlockPlan plan;
  std::vector<int32_t> col_cnt((size_t)nI, 0);
  std::vector<std::vector<int32_t>> pfirst(regions.size()), pcnt(regions.size());
  {
    const char* pe = getenv("MFX_CCD_PASS_WGS");
    const int want_wgs = pe && atoi(pe) > 0 ? atoi(pe) : PASS_WGS;
    const int64_t strip_trips = s->light0 / MFX_BLK_E;
    const int64_t per_wg = std::max<int64_t>(8 * GPW, (strip_trips + want_wgs - 1) / want_wgs);     // at least eight trips per group
    auto lay = [&](size_t q, int nw) {
      Region& R = regions[q];
      pfirst[q].resize(R.pc.size());
      pcnt[q].resize(R.pc.size());
      return mfx_blocks_region(R.pc.data(), R.pc.size(), R.r0, R.r1, nw, R.tag, plan, pfirst[q].data(), pcnt[q].data());
    };
    const int64_t light_trips = (s->light1 - s->light0) / MFX_BLK_E;
    if (light_trips > 0) {
      if (!lay(regions.size() - 1, (int)std::min<int64_t>((light_trips + 8 * GPW - 1) / (8 * GPW), 512)))
        return mfx_fail(ctx, MFX_E_STATE, "mfx_ccdpp_begin: the light region of the column view does not lay out");
    }
    s->nlw = (int)plan.wg_t0.size();
    // the strips share want_wgs workgroups in proportion to their trips (largest remainders first), at least eight steps per group:
    // the total is what was asked for -- 59 strips rounded one by one came to 531 workgroups for 512, i.e. a second round of 19
    std::vector<int> nw(regions.size() - 1, 0);
    {
      const int64_t total = std::max<int64_t>(1, strip_trips);
      const int64_t budget = std::max<int64_t>(1, std::min<int64_t>(want_wgs, total / (8 * GPW)));
      std::vector<std::pair<int64_t, size_t>> frac;
      int64_t used = 0;
      for (size_t q = 0; q + 1 < regions.size(); q++) {
        const int64_t tb = (regions[q].r1 - regions[q].r0) / MFX_BLK_E;
        if (tb == 0) continue;
        nw[q] = (int)std::max<int64_t>(1, tb * budget / total);
        used += nw[q];
        frac.push_back({-(tb * budget % total), q});
      }
      std::sort(frac.begin(), frac.end());
      for (size_t k = 0; k < frac.size() && used < budget; k++, used++) nw[frac[k].second]++;
    }
    (void)per_wg;
    for (size_t q = 0; q + 1 < regions.size(); q++) {
      if (nw[q] == 0) continue;
      if (!lay(q, nw[q]))
        return mfx_fail(ctx, MFX_E_STATE, "mfx_ccdpp_begin: strip %zu of the column view does not lay out", q);
    }
  }
  // the slots of a column, strip-major (the light region last: a column lies in one or the other)
  for (size_t q = 0; q < regions.size(); q++)
    for (size_t k = 0; k < regions[q].pc.size(); k++) col_cnt[(size_t)regions[q].col[k]] += pcnt[q][k];
  std::vector<int32_t> col_ptr((size_t)nI + 1, 0);
  for (int32_t i = 0; i < nI; i++) col_ptr[(size_t)i + 1] = col_ptr[(size_t)i] + col_cnt[(size_t)i];
  std::vector<int32_t> col_seg((size_t)col_ptr[(size_t)nI]);
  {
    std::vector<int32_t> w(col_ptr.begin(), col_ptr.end() - 1);
    for (size_t q = 0; q < regions.size(); q++)
      for (size_t k = 0; k < regions[q].pc.size(); k++)
        for (int32_t c = 0; c < pcnt[q][k]; c++) col_seg[(size_t)w[(size_t)regions[q].col[k]]++] = pfirst[q][k] + c;
  }
  std::vector<int32_t> fin_order;
  fin_order.reserve((size_t)nI);
  for (int pass = 0; pass < 3; pass++) {
    for (int32_t i = 0; i < nI; i++) {
      const int c = col_cnt[(size_t)i], cls = c <= 64 ? 0 : c <= 512 ? 1 : 2;
      if (cls == pass) fin_order.push_back(i);
    }
    if (pass == 0) s->fin_n16 = (int)fin_order.size();
    else if (pass == 1) s->fin_n64 = (int)fin_order.size() - s->fin_n16;
    else s->fin_n256 = (int)fin_order.size() - s->fin_n16 - s->fin_n64;
  }
  if ((rc = up(ctx, &s->fin_order, fin_order))) return rc;
  if ((rc = up(ctx, &s->dst, dst))) return rc;
  if ((rc = up(ctx, &s->plen, plen))) return rc;
  if ((rc = up(ctx, &s->col_ptr, col_ptr))) return rc;
  if ((rc = up(ctx, &s->col_seg, col_seg))) return rc;
  if ((rc = mfx_blocks_upload(ctx, plan, &s->blocks))) return rc;
  if ((rc = up(ctx, &s->rw_blk, rw_blk))) return rc;
  if ((rc = up(ctx, &s->rw_e0, rw_e0))) return rc;
  if ((rc = up(ctx, &s->rw_e1, rw_e1))) return rc;
  if ((rc = up(ctx, &s->rw_stride, rw_stride))) return rc;
  s->nrw = (int)rw_blk.size();
  if ((rc = dev_alloc(ctx, &s->part, (size_t)plan.nslots * 2))) return rc;
  const bool col16 = nI <= 65536;
  const bool has_light = s->light1 > s->light0;
  // buser serves the strips (and what their last workgroup reads behind them); luser the light region and the slack behind it
  if ((rc = dev_alloc(ctx, &s->buser, (size_t)(s->light0 + MFX_BLK_SLACK)))) return rc;
  if ((rc = dev_alloc(ctx, &s->luser, (size_t)(has_light ? nalloc - s->light0 : 1)))) return rc;
  if (col16) { if ((rc = dev_alloc(ctx, &s->bcol16, (size_t)(nalloc / MFX_BLK_EPL)))) return rc; }      // (with the slack: the fused sweep gathers with it)
  else if ((rc = dev_alloc(ctx, &s->bcol32, (size_t)(nalloc / MFX_BLK_EPL)))) return rc;
  if ((rc = dev_alloc(ctx, &s->res, (size_t)nalloc))) return rc;
  // (the residuals are stored quad-interleaved inside a trip: what no piece owns is not a contiguous range of them -- all zero first)
  HIPCHK(hipMemsetAsync(s->res, 0, sizeof(float) * (size_t)nalloc, ctx->stream));
  if (m.nnz > 0) {
    if (col16)
      hipLaunchKernelGGL(strip_scatter_kernel<uint16_t>, dim3(2048), dim3(256), 0, ctx->stream, m.colptr, m.colind, m.colval,
                         s->off, s->dst, s->plen, nI, nb, ctx->nU, s->light0, s->light1, s->buser, s->luser, s->bcol16, s->res);
    else
      hipLaunchKernelGGL(strip_scatter_kernel<int32_t>, dim3(2048), dim3(256), 0, ctx->stream, m.colptr, m.colind, m.colval,
                         s->off, s->dst, s->plen, nI, nb, ctx->nU, s->light0, s->light1, s->buser, s->luser, s->bcol32, s->res);
    HIPCHK(hipGetLastError());
  }
  {
    Ranges rbu, rlu, rbc;
    for (size_t q = 0; q < tails.a.size(); q++) rbc.add(tails.a[q] / MFX_BLK_EPL, tails.b[q] / MFX_BLK_EPL);
    rbc.add(nnzp / MFX_BLK_EPL, nalloc / MFX_BLK_EPL);
    for (size_t q = 0; q < tails.a.size(); q++) {
      if (tails.a[q] < s->light0) rbu.add(tails.a[q], tails.b[q]);
      else rlu.add(tails.a[q] - s->light0, tails.b[q] - s->light0);
    }
    rbu.add(s->light0, s->light0 + MFX_BLK_SLACK);
    if (has_light) rlu.add(nnzp - s->light0, nalloc - s->light0);
    if ((rc = fill_ranges(ctx, s->buser, rbu, (uint16_t)UB))) return rc;
    if ((rc = fill_ranges(ctx, s->luser, rlu, (int32_t)ctx->nU))) return rc;
    if (col16) { if ((rc = fill_ranges(ctx, s->bcol16, rbc, (uint16_t)0))) return rc; }
    else if ((rc = fill_ranges(ctx, s->bcol32, rbc, (int32_t)0))) return rc;
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

// column pass: workgroup w works through window w of the padded view (ccd_blocks.h).  Workgroups [0, nlw): the light columns, u_k
// gathered from L2 by the absolute user id; the others: one strip each, the strip of u_k in LDS and strip-local 16-bit user ids.
// ONE launch, the latency-bound light part first, so that it runs under the strips.
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void colpass_kernel(const int2* __restrict__ rec, const int32_t* __restrict__ wg_t0,
                                                       const int32_t* __restrict__ wg_n, const int64_t* __restrict__ wg_rec,
                                                       const int32_t* __restrict__ wg_tag, const int32_t* __restrict__ wg_stride,
                                                       const float* __restrict__ res, const uint16_t* __restrict__ buser,
                                                       const int32_t* __restrict__ luser, const float* __restrict__ uk, int nU_strips,
                                                       double* __restrict__ part, uint32_t part_bytes) {
  __shared__ __attribute__((aligned(16))) float su[UB + 4];
  const int w = blockIdx.x, b = wg_tag[w];
  if (b >= 0) {
    stage_strip(su, uk, b * UB, min(UB, nU_strips - b * UB));
    if (threadIdx.x == 0) su[UB] = 0.0f;          // what the padding entries gather
    __syncthreads();
  }
  const int j = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int wn = wg_n[w];
  const int2* r = rec + wg_rec[w];
  const int64_t e0 = (int64_t)wg_t0[w] * MFX_BLK_E;
  if (b >= 0) mfx_ccd_block_loop<uint16_t>(r, wn, wg_stride[w], g, res + e0, buser + e0, su, j, part, part_bytes);
  else mfx_ccd_block_loop<int32_t>(r, wn, wg_stride[w], g, res + e0, luser + e0, uk, j, part, part_bytes);      // uk[nU] = +0.0 (mfx_ccdpp_begin)
}

// The FIRST column sweep of a factor with the residual update on the way (ccd_blocks.h, FUSE): three strips of user vectors in LDS --
// the finished factor's u, the new factor's u as extracted (the add-back), and u_k after the row pass (what the sums gather) --, the
// columns' pair per eight entries from cpair.  Strips only (no light region when this runs).
extern __shared__ __attribute__((aligned(16))) float colfuse_lds[];
template <typename ColT>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void colpass_fused_kernel(
    const int2* __restrict__ rec, const int32_t* __restrict__ wg_t0, const int32_t* __restrict__ wg_n, const int64_t* __restrict__ wg_rec,
    const int32_t* __restrict__ wg_tag, const int32_t* __restrict__ wg_stride, float* __restrict__ res, const uint16_t* __restrict__ buser,
    const float* __restrict__ uk0, const float* __restrict__ uk1, const float* __restrict__ uk, int nU_strips, const ColT* __restrict__ bcol8,
    const float2* __restrict__ cpair,
    double* __restrict__ part, uint32_t part_bytes) {
  constexpr int UBS = UB + 4;
  float *s0 = colfuse_lds, *s1 = colfuse_lds + UBS, *s2 = colfuse_lds + 2 * UBS;
  const int w = blockIdx.x, b = wg_tag[w];
  const int n = min(UB, nU_strips - b * UB);
  stage_strip(s0, uk0, b * UB, n);
  stage_strip(s1, uk1, b * UB, n);
  stage_strip(s2, uk, b * UB, n);
  if (threadIdx.x == 0) { s0[UB] = 0.0f; s1[UB] = 0.0f; s2[UB] = 0.0f; }          // what the padding entries gather
  __syncthreads();
  const int j = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int wn = wg_n[w];
  const int64_t e0 = (int64_t)wg_t0[w] * MFX_BLK_E;
  mfx_ccd_block_loop<uint16_t, true, ColT>(rec + wg_rec[w], wn, wg_stride[w], g, res + e0, buser + e0, s2, j, part, part_bytes, s0, s1,
                                           bcol8 + e0 / MFX_BLK_EPL, cpair);
}

// residual update of the light region (MODE as colresid_kernel): u_k from L2
template <int MODE, typename ColT>
__global__ __launch_bounds__(256) void colresid_light_kernel(int64_t e0, int64_t e1, float* __restrict__ res, const int32_t* __restrict__ luser,
                                                             int64_t lshift, const ColT* __restrict__ bcol, const float* __restrict__ uk0,
                                                             const float* __restrict__ vk0, const float* __restrict__ uk1,
                                                             const float* __restrict__ vk1) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < e1; t += stride) {
    const int64_t tl = mfx_blk_entry_of(t);          // residual t (memory order) is entry tl
    const int u = luser[tl - lshift], c = (int)bcol[tl >> 3];
    const float p0 = uk0[u] * vk0[c];
    float r = res[t];
    if (MODE == 1) r = r + p0;
    else r = r - p0;
    if (MODE == 2) r = r + uk1[u] * vk1[c];
    res[t] = r;
  }
}

// v_k[i] from the column's slots in strip-major order.  A column is summed by a TEAM of 16, 64 or 256 lanes by the number of its
// slots (C4: 1.9 M slots over 17 770 columns, the most popular one 2 600): lane-strided in list order, a fixed butterfly over the
// rows of 16, the rows in order, the waves in order -- a fixed association.  (One 16-lane group per column, as for the segment lists of
// rounds 2 and 3, made this launch 90 us: the longest column's 160 dependent index -> slot reads in one group.)
// ONE launch: blocks [0, b16) take 16 columns each from order[0, n16), blocks [b16, b16 + b64) four each from order[n16, n16 + n64),
// the others one each.
// SUMS (sharded runs): only emit (num, den) of this rank's users; coldivide_kernel finishes after the all-reduce.
template <bool SUMS>
__global__ __launch_bounds__(256) void colfinish_kernel(const int32_t* __restrict__ order, int n16, int n64, int b16, int b64,
                                                        const int32_t* __restrict__ col_ptr,
                                                        const int32_t* __restrict__ col_seg,
                                                        const double* __restrict__ part, int32_t ncols, float reg,
                                                        float* __restrict__ vk, const int64_t* __restrict__ colptr,
                                                        float freq_thresh, int k, double* __restrict__ sums) {
  __shared__ double sh[2 * 4 * 4];
  const int bx = blockIdx.x;
  int team, slot_in_block, idx;
  if (bx < b16) { team = 16; slot_in_block = threadIdx.x >> 4; idx = bx * 16 + slot_in_block; if (idx >= n16) return; }
  else if (bx < b16 + b64) { team = 64; slot_in_block = threadIdx.x >> 6; idx = n16 + (bx - b16) * 4 + slot_in_block; if (idx >= n16 + n64) return; }
  else { team = 256; slot_in_block = 0; idx = n16 + n64 + (bx - b16 - b64); }
  const int i = order[idx];
  const int lane = threadIdx.x & (team - 1);
  const int a = col_ptr[i], e = col_ptr[i + 1];
  if (a == e) {         // an item without train ratings is invalid: v_k keeps iFac(i,k) (modelMF.cpp:1079-1081)
    if (SUMS && lane == 0) { sums[2 * i] = 0.0; sums[2 * i + 1] = 0.0; }
    return;
  }
  double num = 0.0, den = 0.0;
  for (int t = a + lane; t < e; t += team) {
    num += part[2 * (int64_t)col_seg[t]];
    den += part[2 * (int64_t)col_seg[t] + 1];
  }
  num = g16_sum(num);
  den = g16_sum(den);
  if (team >= 64) {      // the four rows of a wave, in order
    double n4 = 0.0, d4 = 0.0;
    for (int r = 0; r < 4; r++) {
      n4 += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(num), 16 * r), __builtin_amdgcn_readlane(__double2loint(num), 16 * r));
      d4 += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(den), 16 * r), __builtin_amdgcn_readlane(__double2loint(den), 16 * r));
    }
    num = n4;
    den = d4;
  }
  if (team == 256) {     // the four waves of the block, in order (block-uniform branch: such a block holds one column)
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[2 * wv] = num; sh[2 * wv + 1] = den; }
    __syncthreads();
    num = ((sh[0] + sh[2]) + sh[4]) + sh[6];
    den = ((sh[1] + sh[3]) + sh[5]) + sh[7];
  }
  if (lane == 0) {
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
                                                        const int64_t* __restrict__ rw_e1, const int64_t* __restrict__ rw_stride,
                                                        float* __restrict__ res,
                                                        const uint16_t* __restrict__ buser,
                                                        const ColT* __restrict__ bcol,
                                                        const float* __restrict__ uk0, const float* __restrict__ vk0,
                                                        const float* __restrict__ uk1, const float* __restrict__ vk1,
                                                        int nU) {
  constexpr int UBS = UB + 4;                     // a staged strip and the +0.0 the padding entries (user id UB) gather
  __shared__ __attribute__((aligned(16))) float su[(MODE == 2 ? 2 : 1) * UBS];
  const int b = rw_blk[blockIdx.x];
  const int n = min(UB, nU - b * UB);
  stage_strip(su, uk0, b * UB, n);
  if (MODE == 2) stage_strip(su + UBS, uk1, b * UB, n);
  if (threadIdx.x == 0) { su[UB] = 0.0f; if (MODE == 2) su[UBS + UB] = 0.0f; }
  __syncthreads();
  // 16 aligned bytes per lane and array (4 entries); the entries of a neighbouring workgroup's range are left alone
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef int i4 __attribute__((ext_vector_type(4)));
  const int64_t e0 = rw_e0[blockIdx.x], e1 = rw_e1[blockIdx.x], stride = rw_stride[blockIdx.x];
  for (int64_t t = e0 + 4 * (int64_t)threadIdx.x; t < e1; t += stride) {      // (e0, e1: multiples of 128; a piece = 4 x 1 024 entries)
    const int64_t tl = mfx_blk_entry_of(t);   // the residual quad at t (memory order) holds the entries tl .. tl + 3
    const i4 lu = MfxCcdTrip::load4(buser + tl);
    const int c = (int)bcol[tl >> 3];         // the four entries of a quad lie in ONE column
    f4 r = *(const f4*)(res + t);
    const float v0 = vk0[c], v1 = MODE == 2 ? vk1[c] : 0.0f;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float p0 = su[lu[q]] * v0;        // u_k(u)*v_k(item): float product (modelMF.cpp:1053, :1114)
      if (MODE == 1) r[q] = r[q] + p0;
      else r[q] = r[q] - p0;
      if (MODE == 2) r[q] = r[q] + su[UBS + lu[q]] * v1;
    }
    *(f4*)(res + t) = r;
  }
}

int mfx_ccd_cols_pass(mfx_ctx* ctx, const float* uk, float* vk, float reg, float freq_thresh, int k) {
  ColState* s = st(ctx);
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const bool any = s->blocks.nwg > 0;
  if (any) {
    ProfScope ps(ctx, MFX_K_CCD_COL);
    const MfxBlocks& B = s->blocks;
    // entry t of the padded order sits at luser[t - light0]
    hipLaunchKernelGGL(colpass_kernel, dim3(B.nwg), dim3(16 * GPW), 0, ctx->stream, B.rec, B.wg_t0, B.wg_n, B.wg_rec, B.wg_tag, B.wg_stride, s->res,
                       (const uint16_t*)s->buser, (const int32_t*)(s->luser - s->light0), uk, m.nrows, s->part, (uint32_t)(B.nslots * 16));
    HIPCHK(hipGetLastError());
    if (!mfx_sharded(ctx)) {
      const int b16 = (s->fin_n16 + 15) / 16, b64 = (s->fin_n64 + 3) / 4;
      hipLaunchKernelGGL(colfinish_kernel<false>, dim3((unsigned)(b16 + b64 + s->fin_n256)), dim3(256), 0, ctx->stream, s->fin_order, s->fin_n16,
                         s->fin_n64, b16, b64, s->col_ptr, s->col_seg, s->part, m.ncols, reg, vk, m.colptr, freq_thresh, k, (double*)nullptr);
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
    if (any) {
      const int b16 = (s->fin_n16 + 15) / 16, b64 = (s->fin_n64 + 3) / 4;
      hipLaunchKernelGGL(colfinish_kernel<true>, dim3((unsigned)(b16 + b64 + s->fin_n256)), dim3(256), 0, ctx->stream, s->fin_order, s->fin_n16,
                         s->fin_n64, b16, b64, s->col_ptr, s->col_seg, s->part, m.ncols, reg, vk, m.colptr, freq_thresh, k, s->sums);
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

bool mfx_ccd_cols_can_fuse(mfx_ctx* ctx) {
  ColState* s = st(ctx);
  return s && s->blocks.nwg > 0 && s->nlw == 0 && s->light0 == s->light1;
}
int mfx_ccd_cols_pass_fused(mfx_ctx* ctx, const float* uk0, const float* vk0, const float* uk1, const float* vk1, const float* uk, float* vk,
                            float reg, float freq_thresh, int k) {
  ColState* s = st(ctx);
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const MfxBlocks& B = s->blocks;
  int rc;
  if (!s->cpair && (rc = dev_alloc(ctx, &s->cpair, (size_t)ctx->nI + 1))) return rc;
  ProfScope ps(ctx, MFX_K_CCD_COL);
  if ((rc = mfx_ccd_pairs(ctx, (int64_t)ctx->nI, vk0, vk1, s->cpair))) return rc;
  const size_t lds = 3 * (size_t)(UB + 4) * sizeof(float);
#define MFX_COLFUSE(CT, BC)                                                                                                          \
  do {                                                                                                                               \
    HIPCHK(hipFuncSetAttribute((const void*)colpass_fused_kernel<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
    hipLaunchKernelGGL(colpass_fused_kernel<CT>, dim3(B.nwg), dim3(16 * GPW), lds, ctx->stream, B.rec, B.wg_t0, B.wg_n, B.wg_rec, B.wg_tag, \
                       B.wg_stride, s->res, (const uint16_t*)s->buser, uk0, uk1, uk, m.nrows, (const CT*)BC, (const float2*)s->cpair, s->part, \
                       (uint32_t)(B.nslots * 16));                                                                                   \
  } while (0)
  if (s->bcol16) MFX_COLFUSE(uint16_t, s->bcol16); else MFX_COLFUSE(int32_t, s->bcol32);
#undef MFX_COLFUSE
  HIPCHK(hipGetLastError());
  const int b16 = (s->fin_n16 + 15) / 16, b64 = (s->fin_n64 + 3) / 4;
  hipLaunchKernelGGL(colfinish_kernel<false>, dim3((unsigned)(b16 + b64 + s->fin_n256)), dim3(256), 0, ctx->stream, s->fin_order, s->fin_n16,
                     s->fin_n64, b16, b64, s->col_ptr, s->col_seg, s->part, m.ncols, reg, vk, m.colptr, freq_thresh, k, (double*)nullptr);
  HIPCHK(hipGetLastError());
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
      hipLaunchKernelGGL((colresid_kernel<MD, uint16_t>), dim3(s->nrw), dim3(1024), 0, ctx->stream, s->rw_blk, s->rw_e0, s->rw_e1, s->rw_stride, \
                         s->res, s->buser, s->bcol16, uk0, vk0, uk1, vk1, m.nrows);                                                \
    else                                                                                                                           \
      hipLaunchKernelGGL((colresid_kernel<MD, int32_t>), dim3(s->nrw), dim3(1024), 0, ctx->stream, s->rw_blk, s->rw_e0, s->rw_e1, s->rw_stride, \
                         s->res, s->buser, s->bcol32, uk0, vk0, uk1, vk1, m.nrows);                                                \
  } while (0)
    if (mode == 1) MFX_CR(1); else if (mode == 2) MFX_CR(2); else MFX_CR(-1);
#undef MFX_CR
  }
  if (s->light0 < s->light1) {
    const unsigned lb = (unsigned)std::min<int64_t>((s->light1 - s->light0 + 255) / 256, 4096);
    const int64_t lsh = s->light0;
#define MFX_CL(MD)                                                                                                                \
  do {                                                                                                                            \
    if (s->bcol16)                                                                                                                \
      hipLaunchKernelGGL((colresid_light_kernel<MD, uint16_t>), dim3(lb), dim3(256), 0, ctx->stream, s->light0, s->light1, s->res,\
                         s->luser, lsh, s->bcol16, uk0, vk0, uk1, vk1);                                                           \
    else                                                                                                                          \
      hipLaunchKernelGGL((colresid_light_kernel<MD, int32_t>), dim3(lb), dim3(256), 0, ctx->stream, s->light0, s->light1, s->res, \
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
