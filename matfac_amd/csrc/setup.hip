// setup.hip -- one-time preprocessing of the train matrix on the device: the column view
// (gk_csr_CreateIndex(GK_CSR_COL), called from Data::Data, datastruct.cpp:60-62) and the slot lists of
// MFX_SGD_TILED.  Integer work: radix sorts / scans from rocPRIM plus a few hand-written passes; both
// results are bit-identical to the host builders they replace (tests/test_setup_gpu.py).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include "sgd_slots.h"

namespace {
// device temporaries of one build, released together
struct Scratch {
  std::vector<void*> ptrs;
  ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
  template <typename T>
  int get(mfx_ctx* ctx, T** p, size_t n) {
    *p = nullptr;
    HIPCHK(hipMalloc((void**)p, (n ? n : 1) * sizeof(T)));
    ptrs.push_back(*p);
    return MFX_OK;
  }
};
static int bits_for(uint64_t n) {   // bits needed for values in [0, n)
  int b = 1;
  while (b < 63 && ((uint64_t)1 << b) < n) b++;
  return b;
}
constexpr int TB = 256;
static inline int grid_for(int64_t n) { return (int)std::min<int64_t>(std::max<int64_t>((n + TB - 1) / TB, 1), 1 << 16); }
}  // namespace

template <typename K, typename V>
static int sort_pairs(mfx_ctx* ctx, Scratch& sc, const K* kin, K* kout, const V* vin, V* vout, size_t n, int bits) {
  size_t bytes = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, bits, ctx->stream));
  char* tmp;
  int rc = sc.get(ctx, &tmp, bytes);
  if (rc) return rc;
  HIPCHK(rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, 0, bits, ctx->stream));
  return MFX_OK;
}

// ---------------------------------------------------------------------------
// column view
// ---------------------------------------------------------------------------
__global__ void col_keys_kernel(const int32_t* __restrict__ rowind, int64_t nnz, uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    key[e] = (uint32_t)rowind[e];
    val[e] = (uint32_t)e;
  }
}
__global__ void col_gather_kernel(const uint32_t* __restrict__ src, const int32_t* __restrict__ rowid,
                                  const float* __restrict__ rowval, int64_t nnz, int32_t* __restrict__ colind,
                                  float* __restrict__ colval) {
  for (int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; d < nnz; d += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t e = src[d];
    colind[d] = rowid[e];
    colval[d] = rowval[e];
  }
}
// ptr[j] = number of sorted keys < j
__global__ void lower_bounds_kernel(const uint32_t* __restrict__ keys, int64_t n, int64_t nptr, int64_t* __restrict__ ptr) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < nptr; j += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)keys[mid] < j) lo = mid + 1; else hi = mid;
    }
    ptr[j] = lo;
  }
}

// Stable sort of the ratings by column = the reference's counting sort: inside a column the users ascend.
int mfx_build_col_index_device(mfx_ctx* ctx, DevCSR& m) {
  const int64_t nnz = m.nnz;
  NEED(nnz < ((int64_t)1 << 31), MFX_E_ARG, "column view on the device: nnz must be < 2^31");
  int rc;
  if ((rc = dev_alloc(ctx, &m.colptr, (size_t)m.ncols + 1))) return rc;
  if ((rc = dev_alloc(ctx, &m.colind, (size_t)nnz))) return rc;
  if ((rc = dev_alloc(ctx, &m.colval, (size_t)nnz))) return rc;
  Scratch sc;
  uint32_t *k0, *k1, *v0, *v1;
  if ((rc = sc.get(ctx, &k0, (size_t)nnz)) || (rc = sc.get(ctx, &k1, (size_t)nnz)) || (rc = sc.get(ctx, &v0, (size_t)nnz)) ||
      (rc = sc.get(ctx, &v1, (size_t)nnz)))
    return rc;
  if (nnz) {
    hipLaunchKernelGGL(col_keys_kernel, dim3(grid_for(nnz)), dim3(TB), 0, ctx->stream, m.rowind, nnz, k0, v0);
    if ((rc = sort_pairs(ctx, sc, k0, k1, v0, v1, (size_t)nnz, bits_for((uint64_t)std::max(m.ncols, 1))))) return rc;
    hipLaunchKernelGGL(col_gather_kernel, dim3(grid_for(nnz)), dim3(TB), 0, ctx->stream, v1, m.rowid, m.rowval, nnz, m.colind, m.colval);
  }
  hipLaunchKernelGGL(lower_bounds_kernel, dim3(grid_for(m.ncols + 1)), dim3(TB), 0, ctx->stream, k1, nnz, (int64_t)m.ncols + 1, m.colptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  m.has_col = true;
  return MFX_OK;
}

int mfx_build_c2r_map_device(mfx_ctx* ctx, const DevCSR& m, uint32_t** map) {
  const int64_t nnz = m.nnz;
  NEED(nnz < ((int64_t)1 << 31), MFX_E_ARG, "position map on the device: nnz must be < 2^31");
  int rc;
  if ((rc = dev_alloc(ctx, map, (size_t)nnz))) return rc;
  if (!nnz) return MFX_OK;
  Scratch sc;
  uint32_t *k0, *k1, *v0;
  if ((rc = sc.get(ctx, &k0, (size_t)nnz)) || (rc = sc.get(ctx, &k1, (size_t)nnz)) || (rc = sc.get(ctx, &v0, (size_t)nnz))) return rc;
  hipLaunchKernelGGL(col_keys_kernel, dim3(grid_for(nnz)), dim3(TB), 0, ctx->stream, m.rowind, nnz, k0, v0);
  if ((rc = sort_pairs(ctx, sc, k0, k1, v0, *map, (size_t)nnz, bits_for((uint64_t)std::max(m.ncols, 1))))) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MFX_OK;
}

// ---------------------------------------------------------------------------
// slot lists
// ---------------------------------------------------------------------------
__global__ void blk_count_kernel(const int32_t* __restrict__ u, const int32_t* __restrict__ i, int64_t n, int32_t* __restrict__ cu,
                                 int32_t* __restrict__ ci) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    atomicAdd(&cu[u[e]], 1);
    atomicAdd(&ci[i[e]], 1);
  }
}
// rows by descending count (ties: ascending id), each onto the block with the smallest load so far (ties: the lowest block)
// Tiling t > 0 (sgd_slots.h): the same dealing over a JITTERED order -- every count scaled by a factor in [1, 1.25) drawn from a hash
// of (row, tiling) -- so that rows of similar weight change places (ties, i.e. most rows, are shuffled outright) and end up in
// different company, while every block still fills up longest-first.
static void balanced_blocks(const int32_t* cnt, int64_t n, int nb, bool user, bool hash_only, int tiling, std::vector<uint8_t>& blk) {
  blk.resize((size_t)n);
  for (int64_t r = 0; r < n; r++) {
    const int32_t rr = tiling == 0 ? (int32_t)r : (int32_t)mfx_mix32((uint32_t)r + 0x9e3779b9U * (uint32_t)tiling);
    blk[(size_t)r] = (uint8_t)(user ? slot_user_block(rr) : mfx_item_block(rr));
  }
  if (hash_only) return;
  std::vector<int32_t> rows;
  for (int64_t r = 0; r < n; r++)
    if (cnt[r] > 0) rows.push_back((int32_t)r);
  if (tiling == 0) {
    std::sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) { return cnt[a] != cnt[b] ? cnt[a] > cnt[b] : a < b; });
  } else {
    std::vector<double> key((size_t)n, 0.0);
    const uint32_t salt = mfx_mix32(0x51ed270bU * (uint32_t)tiling + (user ? 0x68bc21ebU : 0x02e5be93U));
    for (int32_t r : rows)
      key[(size_t)r] = (double)cnt[r] * (1.0 + 0.25 * ((double)mfx_mix32((uint32_t)r ^ salt) * (1.0 / 4294967296.0)));
    std::sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) { return key[(size_t)a] != key[(size_t)b] ? key[(size_t)a] > key[(size_t)b] : a < b; });
  }
  std::vector<int64_t> load((size_t)nb, 0);
  for (int32_t r : rows) {
    int best = 0;
    for (int b = 1; b < nb; b++)
      if (load[(size_t)b] < load[(size_t)best]) best = b;
    blk[(size_t)r] = (uint8_t)best;
    load[(size_t)best] += cnt[r];
  }
}
int mfx_slots_block_tables(mfx_ctx* ctx, SlotList* S, const RatingView& view) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const int64_t nU = m.nrows, nI = m.ncols;
  static_assert(NUB <= 256, "block ids are bytes");
  int rc;
  Scratch sc;
  int32_t* cnt;
  if ((rc = sc.get(ctx, &cnt, (size_t)(nU + nI)))) return rc;
  const char* e = getenv("MFX_SGD_BLOCKS_HASH");
  const bool hash_only = e && atoi(e) != 0;
  std::vector<int32_t> h((size_t)(nU + nI), 0);
  if (!hash_only && view.n > 0) {
    HIPCHK(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)(nU + nI), ctx->stream));
    hipLaunchKernelGGL(blk_count_kernel, dim3(grid_for(view.n)), dim3(TB), 0, ctx->stream, view.u, view.i, view.n, cnt, cnt + nU);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h.data(), cnt, sizeof(int32_t) * (size_t)(nU + nI), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  balanced_blocks(h.data(), nU, NUB, true, hash_only, S->tiling, S->h_ublk);
  balanced_blocks(h.data() + nU, nI, 8, false, hash_only, S->tiling, S->h_iblk);
  dev_free(S->ublk); dev_free(S->iblk);
  if ((rc = dev_alloc(ctx, &S->ublk, (size_t)std::max<int64_t>(nU, 1))) || (rc = dev_alloc(ctx, &S->iblk, (size_t)std::max<int64_t>(nI, 1)))) return rc;
  if (nU) HIPCHK(hipMemcpyAsync(S->ublk, S->h_ublk.data(), (size_t)nU, hipMemcpyHostToDevice, ctx->stream));
  if (nI) HIPCHK(hipMemcpyAsync(S->iblk, S->h_iblk.data(), (size_t)nI, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MFX_OK;
}

// key = tile << ownbits | owned index; value = CSR position
__global__ void slot_keys_kernel(const int32_t* __restrict__ rowid, const int32_t* __restrict__ rowind, int64_t nnz, int side,
                                 int ownbits, const uint8_t* __restrict__ ublk, const uint8_t* __restrict__ iblk,
                                 uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t u = rowid[e], i = rowind[e];
    const uint64_t tile = (uint64_t)((int)ublk[u] * 8 + (int)iblk[i]);
    key[e] = (tile << ownbits) | (uint64_t)(uint32_t)(side == 0 ? i : u);
    val[e] = (uint32_t)e;
  }
}
// runs (tile, owned row) ordered inside a tile by descending length; the stable sort keeps ascending ids on ties
__global__ void run_keys_kernel(const uint64_t* __restrict__ uniq, const uint32_t* __restrict__ cnt, int64_t R, int ownbits,
                                int cbits, uint32_t cmax, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    key[r] = ((uint64_t)(uniq[r] >> ownbits) << cbits) | (uint64_t)(cmax - cnt[r]);
    val[r] = (uint32_t)r;
  }
}
__global__ void run_gather_kernel(const uint32_t* __restrict__ ord, const uint64_t* __restrict__ uniq,
                                  const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ src, int64_t R, int ownbits,
                                  uint32_t* __restrict__ tile2, uint32_t* __restrict__ cnt2, uint32_t* __restrict__ src2,
                                  int32_t* __restrict__ own2) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t o = ord[r];
    const uint64_t k = uniq[o];
    tile2[r] = (uint32_t)(k >> ownbits);
    own2[r] = (int32_t)(k & (((uint64_t)1 << ownbits) - 1));
    cnt2[r] = cnt[o];
    src2[r] = src[o];
  }
}
// The greedy cut of a tile's runs into slots (the loop of the host builder), one wavefront per tile:
// 64 run lengths are loaded at a time and walked with a uniform state.
__global__ __launch_bounds__(64) void slot_cut_kernel(const int64_t* __restrict__ trun, const uint32_t* __restrict__ cnt2, int rows,
                                                      int32_t* __restrict__ li, uint32_t* __restrict__ head) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const int64_t b = trun[t], e = trun[t + 1];
  int cur_r = 0, cur_i = 0;
  for (int64_t base = b; base < e; base += 64) {
    const int64_t r = base + lane;
    const int n_l = r < e ? (int)min(cnt2[r], (uint32_t)0x7fffffff) : 0;
    const int m = (int)min((int64_t)64, e - base);
    int my_li = 0, my_head = 0;
    for (int j = 0; j < m; j++) {
      const int n = __shfl(n_l, j, 64);
      int h, l;
      if (n > CAP_R / 2) { h = 1; l = 0; cur_r = 0; cur_i = 0; }    // a slot of its own
      else {
        if (cur_r + n > CAP_R || cur_i == rows) { cur_r = 0; cur_i = 0; }
        h = cur_r == 0; l = cur_i;
        cur_r += n; cur_i++;
      }
      if (j == lane) { my_li = l; my_head = h; }
    }
    if (r < e) { li[r] = my_li; head[r] = (uint32_t)my_head; }
  }
}
__global__ void slot_heads_kernel(const uint32_t* __restrict__ head, const uint32_t* __restrict__ hs, const int64_t* __restrict__ dst,
                                  int64_t R, int64_t nslots, int64_t nnz, const int64_t* __restrict__ trun, int ntile,
                                  int64_t* __restrict__ slot_beg, int32_t* __restrict__ slot_ibeg, int32_t* __restrict__ tile_slot) {
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (int64_t r = gid; r < R; r += (int64_t)gridDim.x * blockDim.x)
    if (head[r]) { slot_beg[hs[r] - 1] = dst[r]; slot_ibeg[hs[r] - 1] = (int32_t)r; }
  if (gid == 0) { slot_beg[nslots] = nnz; slot_ibeg[nslots] = (int32_t)R; }
  if (gid <= ntile) tile_slot[gid] = trun[gid] < R ? (int32_t)(hs[trun[gid]] - 1) : (int32_t)nslots;
}
// rating o of the slot-ordered list: its run by binary search over the runs' first positions
__global__ void slot_scatter_kernel(const int64_t* __restrict__ dst, int64_t R, int64_t nnz, const uint32_t* __restrict__ src2,
                                    const uint32_t* __restrict__ sorted_e, const int32_t* __restrict__ li,
                                    const int32_t* __restrict__ own2, const int32_t* __restrict__ rowid,
                                    const int32_t* __restrict__ rowind, const float* __restrict__ rowval, int side,
                                    int4* __restrict__ rec) {
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < nnz; o += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = R;              // last run with dst[run] <= o
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (dst[mid] <= o) lo = mid; else hi = mid;
    }
    const uint32_t e = sorted_e[(int64_t)src2[lo] + (o - dst[lo])];
    rec[o] = make_int4(side == 0 ? rowid[e] : rowind[e], li[lo], __float_as_int(rowval[e]), own2[lo]);
  }
}
__global__ void tile_bounds_kernel(const uint32_t* __restrict__ tile2, int64_t R, int ntile, int64_t* __restrict__ trun) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > ntile) return;
  int64_t lo = 0, hi = R;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int)tile2[mid] < t) lo = mid + 1; else hi = mid;
  }
  trun[t] = lo;
}

// ---- item parts: the train ratings grouped by part = item % nparts, CSR order inside a part (stable radix sort) ----------
__global__ void part_keys_kernel(const int32_t* __restrict__ rowind, int64_t nnz, int nparts, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    key[e] = (uint64_t)((uint32_t)rowind[e] % (uint32_t)nparts);
    val[e] = (uint32_t)e;
  }
}
__global__ void part_gather_kernel(const uint32_t* __restrict__ order, int64_t nnz, const int32_t* __restrict__ rowid, const int32_t* __restrict__ rowind,
                                   const float* __restrict__ rowval, int32_t* __restrict__ pu, int32_t* __restrict__ pi, float* __restrict__ pv) {
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < nnz; o += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t e = order[o];
    pu[o] = rowid[e]; pi[o] = rowind[e]; pv[o] = rowval[e];
  }
}
__global__ void part_bounds_kernel(const uint64_t* __restrict__ keys, int64_t nnz, int nparts, int64_t* __restrict__ poff) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > nparts) return;
  int64_t lo = 0, hi = nnz;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)keys[mid] < p) lo = mid + 1; else hi = mid;
  }
  poff[p] = lo;
}
int mfx_slots_group_by_part(mfx_ctx* ctx, SlotState* S, int nparts) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const int64_t nnz = m.nnz;
  NEED(nnz > 0 && nnz < ((int64_t)1 << 31), MFX_E_ARG, "item parts: need 0 < nnz < 2^31");
  NEED(nparts >= 2 && nparts <= 1024, MFX_E_ARG, "item parts: nparts=%d", nparts);
  Scratch sc;
  int rc;
  hipStream_t st = ctx->stream;
  uint64_t *k0, *k1;
  uint32_t *v0, *v1;
  int64_t* dpoff;
  if ((rc = sc.get(ctx, &k0, (size_t)nnz)) || (rc = sc.get(ctx, &k1, (size_t)nnz)) || (rc = sc.get(ctx, &v0, (size_t)nnz)) ||
      (rc = sc.get(ctx, &v1, (size_t)nnz)) || (rc = sc.get(ctx, &dpoff, (size_t)nparts + 1)))
    return rc;
  hipLaunchKernelGGL(part_keys_kernel, dim3(grid_for(nnz)), dim3(TB), 0, st, m.rowind, nnz, nparts, k0, v0);
  if ((rc = sort_pairs(ctx, sc, k0, k1, v0, v1, (size_t)nnz, bits_for((uint64_t)nparts)))) return rc;
  dev_free(S->pu); dev_free(S->pi); dev_free(S->pv);
  if ((rc = dev_alloc(ctx, &S->pu, (size_t)nnz)) || (rc = dev_alloc(ctx, &S->pi, (size_t)nnz)) || (rc = dev_alloc(ctx, &S->pv, (size_t)nnz))) return rc;
  hipLaunchKernelGGL(part_gather_kernel, dim3(grid_for(nnz)), dim3(TB), 0, st, v1, nnz, m.rowid, m.rowind, m.rowval, S->pu, S->pi, S->pv);
  hipLaunchKernelGGL(part_bounds_kernel, dim3((nparts + TB) / TB), dim3(TB), 0, st, k1, nnz, nparts, dpoff);
  HIPCHK(hipGetLastError());
  S->poff.assign((size_t)nparts + 1, 0);
  HIPCHK(hipMemcpyAsync(S->poff.data(), dpoff, sizeof(int64_t) * ((size_t)nparts + 1), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  S->nparts = nparts;
  return MFX_OK;
}

int mfx_slots_build_device(mfx_ctx* ctx, SlotList* S, int rows, int side, const RatingView& view) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const int64_t nnz = view.n;
  NEED(nnz > 0 && nnz < ((int64_t)1 << 31), MFX_E_ARG, "slot lists on the device: need 0 < nnz < 2^31");
  const int32_t nown = side == 0 ? m.ncols : m.nrows;
  const int ownbits = bits_for((uint64_t)std::max(nown, 1)), tilebits = bits_for(NTILE);
  Scratch sc;
  int rc;
  hipStream_t st = ctx->stream;

  // 1. ratings grouped by (tile, owned row), CSR order inside a group
  uint64_t *k0, *k1;
  uint32_t *v0, *v1;
  if ((rc = sc.get(ctx, &k0, (size_t)nnz)) || (rc = sc.get(ctx, &k1, (size_t)nnz)) || (rc = sc.get(ctx, &v0, (size_t)nnz)) ||
      (rc = sc.get(ctx, &v1, (size_t)nnz)))
    return rc;
  if ((rc = mfx_slots_block_tables(ctx, S, view))) return rc;
  hipLaunchKernelGGL(slot_keys_kernel, dim3(grid_for(nnz)), dim3(TB), 0, st, view.u, view.i, nnz, side, ownbits, S->ublk, S->iblk, k0, v0);
  if ((rc = sort_pairs(ctx, sc, k0, k1, v0, v1, (size_t)nnz, ownbits + tilebits))) return rc;

  // 2. runs: (tile, owned row) -> number of ratings; k0 / v0 are free again and hold the unique keys / counts
  uint64_t* uniq = k0;
  uint32_t* cnt = v0;
  uint64_t* d_R;      // [0] number of runs, [1] longest run
  if ((rc = sc.get(ctx, &d_R, 2))) return rc;
  {
    size_t bytes = 0;
    HIPCHK(rocprim::run_length_encode(nullptr, bytes, k1, (unsigned)nnz, uniq, cnt, d_R, st));
    char* tmp;
    if ((rc = sc.get(ctx, &tmp, bytes))) return rc;
    HIPCHK(rocprim::run_length_encode(tmp, bytes, k1, (unsigned)nnz, uniq, cnt, d_R, st));
  }
  uint64_t hR = 0;
  HIPCHK(hipMemcpyAsync(&hR, d_R, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const int64_t R = (int64_t)hR;
  uint32_t* d_max = (uint32_t*)(d_R + 1);
  {
    size_t bytes = 0;
    HIPCHK(rocprim::reduce(nullptr, bytes, cnt, d_max, 0u, (size_t)R, rocprim::maximum<uint32_t>(), st));
    char* tmp;
    if ((rc = sc.get(ctx, &tmp, bytes))) return rc;
    HIPCHK(rocprim::reduce(tmp, bytes, cnt, d_max, 0u, (size_t)R, rocprim::maximum<uint32_t>(), st));
  }
  uint32_t cmax = 0;
  HIPCHK(hipMemcpyAsync(&cmax, d_max, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  // first position of each run in the sorted order
  uint32_t* src;
  if ((rc = sc.get(ctx, &src, (size_t)R))) return rc;
  {
    size_t bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, bytes, cnt, src, 0u, (size_t)R, rocprim::plus<uint32_t>(), st));
    char* tmp;
    if ((rc = sc.get(ctx, &tmp, bytes))) return rc;
    HIPCHK(rocprim::exclusive_scan(tmp, bytes, cnt, src, 0u, (size_t)R, rocprim::plus<uint32_t>(), st));
  }
  HIPCHK(hipStreamSynchronize(st));

  // 3. runs of a tile by descending length
  const int cbits = bits_for((uint64_t)cmax + 1);
  uint64_t *rk0, *rk1;
  uint32_t *rv0, *ord;
  if ((rc = sc.get(ctx, &rk0, (size_t)R)) || (rc = sc.get(ctx, &rk1, (size_t)R)) || (rc = sc.get(ctx, &rv0, (size_t)R)) ||
      (rc = sc.get(ctx, &ord, (size_t)R)))
    return rc;
  hipLaunchKernelGGL(run_keys_kernel, dim3(grid_for(R)), dim3(TB), 0, st, uniq, cnt, R, ownbits, cbits, cmax, rk0, rv0);
  if ((rc = sort_pairs(ctx, sc, rk0, rk1, rv0, ord, (size_t)R, cbits + tilebits))) return rc;
  uint32_t *tile2, *cnt2, *src2, *head, *hs;
  int32_t *own2, *li;
  int64_t *dst, *trun;
  if ((rc = sc.get(ctx, &tile2, (size_t)R)) || (rc = sc.get(ctx, &cnt2, (size_t)R)) || (rc = sc.get(ctx, &src2, (size_t)R)) ||
      (rc = sc.get(ctx, &head, (size_t)R)) || (rc = sc.get(ctx, &hs, (size_t)R)) || (rc = sc.get(ctx, &li, (size_t)R)) ||
      (rc = sc.get(ctx, &dst, (size_t)R + 1)) || (rc = sc.get(ctx, &trun, (size_t)NTILE + 1)))
    return rc;
  if ((rc = dev_alloc(ctx, &own2, (size_t)R))) return rc;     // becomes S->slot_items
  hipLaunchKernelGGL(run_gather_kernel, dim3(grid_for(R)), dim3(TB), 0, st, ord, uniq, cnt, src, R, ownbits, tile2, cnt2, src2, own2);
  {
    size_t bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, bytes, cnt2, dst, (int64_t)0, (size_t)R, rocprim::plus<int64_t>(), st));
    char* tmp;
    if ((rc = sc.get(ctx, &tmp, bytes))) { dev_free(own2); return rc; }
    HIPCHK(rocprim::exclusive_scan(tmp, bytes, cnt2, dst, (int64_t)0, (size_t)R, rocprim::plus<int64_t>(), st));
  }
  hipLaunchKernelGGL(tile_bounds_kernel, dim3(1), dim3(TB), 0, st, tile2, R, NTILE, trun);

  // 4. cut into slots
  hipLaunchKernelGGL(slot_cut_kernel, dim3(NTILE), dim3(64), 0, st, trun, cnt2, rows, li, head);
  {
    size_t bytes = 0;
    HIPCHK(rocprim::inclusive_scan(nullptr, bytes, head, hs, (size_t)R, rocprim::plus<uint32_t>(), st));
    char* tmp;
    if ((rc = sc.get(ctx, &tmp, bytes))) { dev_free(own2); return rc; }
    HIPCHK(rocprim::inclusive_scan(tmp, bytes, head, hs, (size_t)R, rocprim::plus<uint32_t>(), st));
  }
  uint32_t hns = 0;
  if (hipMemcpyAsync(&hns, hs + (R - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    dev_free(own2);
    return mfx_fail(ctx, MFX_E_HIP, "slot lists on the device: %s", hipGetErrorString(hipGetLastError()));
  }
  const int64_t nslots = (int64_t)hns;

  dev_free(S->slot_need); dev_free(S->rowver); S->rowver_n = 0;      // (of the lists that go)
  dev_free(S->visit); dev_free(S->attr); dev_free(S->rec); dev_free(S->slot_beg); dev_free(S->slot_ibeg); dev_free(S->slot_items); dev_free(S->tile_slot);
  S->slot_items = own2;
  if ((rc = dev_alloc(ctx, &S->rec, (size_t)nnz * 4)) || (rc = dev_alloc(ctx, &S->slot_beg, (size_t)nslots + 1)) ||
      (rc = dev_alloc(ctx, &S->slot_ibeg, (size_t)nslots + 1)) || (rc = dev_alloc(ctx, &S->tile_slot, (size_t)NTILE + 1)))
    return rc;
  if (!S->ctr) {     // [NTILE] slot counters, barrier counter, abort flag, and the STICKY abort flag (zeroed here, cleared only when reported)
    if ((rc = dev_alloc(ctx, &S->ctr, (size_t)CTR_WORDS))) return rc;
    HIPCHK(hipMemsetAsync(S->ctr, 0, CTR_WORDS * sizeof(unsigned), ctx->stream));
  }
  static_assert(NTILE + 1 <= TB, "tile_slot is written by the first workgroup");
  hipLaunchKernelGGL(slot_heads_kernel, dim3(grid_for(R)), dim3(TB), 0, st, head, hs, dst, R, nslots, nnz, trun, NTILE,
                     S->slot_beg, S->slot_ibeg, S->tile_slot);
  // 5. the ratings in slot order
  hipLaunchKernelGGL(slot_scatter_kernel, dim3(grid_for(nnz)), dim3(TB), 0, st, dst, R, nnz, src2, v1, li, own2, view.u, view.i,
                     view.r, side, (int4*)S->rec);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  if (getenv("MFX_DEBUG"))
    fprintf(stderr, "[mfx] slots (%s rows owned, device build): %lld for %lld ratings, %lld row refs, longest run %u\n",
            side == 0 ? "item" : "user", (long long)nslots, (long long)nnz, (long long)R, cmax);
  S->nslots = nslots;
  S->nnz = nnz;
  S->rows = rows;
  S->built = true;
  return MFX_OK;
}


// ---------------------------------------------------------------------------
// std::shuffle's swaps applied on the device (mfx_sgd_apply_swaps32)
// ---------------------------------------------------------------------------
// The list a of mfx_sgd_set_order32 after  for i = 1 .. n-1: swap(a[i], a[pos[i]])  (0 <= pos[i] <= i: libstdc++'s std::shuffle beyond
// 65 536 entries, modelMF.cpp:76-81), without walking the swaps one by one.  Step i moves TWO elements: the one that sat at i since the
// start goes to pos[i], the one that sat at pos[i] comes to i (nothing moves when pos[i] == i).  So an element's way through the list is
//     i  --(time i)-->  pos[i]  --(the next time t > i with pos[t] == pos[i])-->  t  --(the first time t' > t with pos[t'] == t)-->  t' ...
// i.e. after its first hop it follows q -> head[q] := the first step AFTER q that hits position q, a forest whose edges all point
// forward.  With the steps sorted stably by the position they hit (one radix sort of (pos[t], t)):  after[t] = the next step of t's group,
// head[q] = the first step of group q other than q itself;  F[q] = the end of q's chain by pointer jumping (a handful of rounds: the
// chains are O(log n) long);  the element of place i ends up at  after[i] == none ? pos[i] : F[after[i]].
// Bit for bit the sequential swaps (tests/test_setup_gpu.py against a plain loop, and every exact-replay test through the host class).
namespace {
constexpr uint32_t SW_NONE = 0xffffffffu;
__global__ void sw_iota_kernel(uint32_t* __restrict__ v, int64_t n, uint32_t* __restrict__ pos0) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) v[t] = (uint32_t)t;
  if (blockIdx.x == 0 && threadIdx.x == 0 && pos0) pos0[0] = 0u;          // (step 0 does not exist: a[0] "swaps with itself")
}
__global__ void sw_links_kernel(const uint32_t* __restrict__ ks, const uint32_t* __restrict__ vs, int64_t n, uint32_t* __restrict__ after,
                                uint32_t* __restrict__ head) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t q = ks[k], t = vs[k];
    const uint32_t nxt = (k + 1 < n && ks[k + 1] == q) ? vs[k + 1] : SW_NONE;
    after[t] = nxt;
    if (k == 0 || ks[k - 1] != q) head[q] = t != q ? t : nxt;           // t == q: the step that hit its own place; it moved nothing
  }
}
// F <- F o F where F[q] = q marks the end of a chain; *changed is raised when any entry moved
__global__ void sw_jump_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int64_t n, unsigned* __restrict__ changed) {
  bool ch = false;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t a = in[q], b = in[a];
    out[q] = b;
    ch = ch || b != a;
  }
  if (__builtin_amdgcn_ballot_w64(ch) != 0ull && (threadIdx.x & 63) == 0) atomicOr(changed, 1u);
}
__global__ void sw_selfend_kernel(uint32_t* __restrict__ head, int64_t n) {
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x)
    if (head[q] == SW_NONE) head[q] = (uint32_t)q;
}
__global__ void sw_scatter_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ after,
                                  const uint32_t* __restrict__ F, int64_t n, uint32_t* __restrict__ out, uint64_t* __restrict__ wide) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t af = after[i];
    const uint32_t dst = af == SW_NONE ? pos[i] : F[af];
    const uint32_t v = a[i];
    out[dst] = v;
    wide[dst] = (uint64_t)v;
  }
}
// The same without the jumping rounds: every element walks its own chain (head[q] == q ends it).  For positions drawn by a generator the
// chains are a hop or two long -- a place q is hit again with probability 1 - (q + 1) / n -- and six rounds of F o F over 20 M entries
// (two random gathers each: 6.9 ms at the ML-20M size) cost five times the walks.  A chain longer than 64 hops raises *over and
// the caller takes the jumping rounds (any pos[i] <= i is handled, e.g. "every step hits its left neighbour": one chain of n hops).
__global__ void sw_walk_scatter_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ after,
                                       const uint32_t* __restrict__ head, int64_t n, uint32_t* __restrict__ out, uint64_t* __restrict__ wide,
                                       unsigned* __restrict__ over) {
  bool ov = false;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t af = after[i];
    uint32_t dst = pos[i];
    if (af != SW_NONE) {
      uint32_t q = af, h = head[q];
      int hops = 0;
      while (h != q && hops < 64) { q = h; h = head[q]; hops++; }
      ov = ov || h != q;
      dst = q;
    }
    const uint32_t v = a[i];
    out[dst] = v;
    wide[dst] = (uint64_t)v;
  }
  if (__builtin_amdgcn_ballot_w64(ov) != 0ull && (threadIdx.x & 63) == 0) atomicOr(over, 1u);
}
}  // namespace

extern "C" int mfx_sgd_apply_swaps32(mfx_ctx* ctx, const uint32_t* pos, int64_t n) {
  if (!ctx) return MFX_E_ARG;
  NEED(pos && n >= 0 && n < ((int64_t)1 << 32), MFX_E_ARG, "mfx_sgd_apply_swaps32: pos NULL or n outside [0, 2^32)");
  NEED(ctx->order32 && ctx->order32_valid && ctx->order && ctx->order_n == n && ctx->order32_cap >= n, MFX_E_STATE,
       "mfx_sgd_apply_swaps32: the list to shuffle is the one of mfx_sgd_set_order32 with the same length (have %lld, swaps for %lld)",
       (long long)ctx->order_n, (long long)n);
  if (n < 2) return MFX_OK;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rc;
  // eight arrays of n entries (+ one flag word) and the sort's workspace, allocated once per list length: an epoch of an ML-20M run
  // must not pay nine hipMalloc / hipFree pairs of 80 MB (measured: 12 ms per call with them, most of it allocation)
  const size_t per = ((size_t)n + 63) / 64 * 64;
  if (ctx->sw_cap < per || !ctx->sw_buf) {
    dev_free(ctx->sw_buf);
    ctx->sw_cap = 0;
    if ((rc = dev_alloc(ctx, &ctx->sw_buf, 8 * per + 64))) return rc;
    ctx->sw_cap = per;
  }
  uint32_t* B = ctx->sw_buf;
  const size_t cp = ctx->sw_cap;
  uint32_t *dpos = B, *ks = B + cp, *v0 = B + 2 * cp, *vs = B + 3 * cp, *after = B + 4 * cp, *f0 = B + 5 * cp, *f1 = B + 6 * cp, *out = B + 7 * cp;
  unsigned* changed = (unsigned*)(B + 8 * cp);
  HIPCHK(hipMemcpyAsync(dpos, pos, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, st));
  const int grid = grid_for(n);
  hipLaunchKernelGGL(sw_iota_kernel, dim3(grid), dim3(TB), 0, st, v0, n, dpos);
  {
    const int bits = bits_for((uint64_t)n);
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, dpos, ks, v0, vs, (size_t)n, 0, bits, st));
    if (ctx->sw_tmp_bytes < bytes || !ctx->sw_tmp) {
      dev_free(ctx->sw_tmp);
      ctx->sw_tmp_bytes = 0;
      if ((rc = dev_alloc(ctx, &ctx->sw_tmp, bytes))) return rc;
      ctx->sw_tmp_bytes = bytes;
    }
    size_t have = ctx->sw_tmp_bytes;
    HIPCHK(rocprim::radix_sort_pairs(ctx->sw_tmp, have, dpos, ks, v0, vs, (size_t)n, 0, bits, st));
  }
  HIPCHK(hipMemsetAsync(f0, 0xff, sizeof(uint32_t) * (size_t)n, st));
  hipLaunchKernelGGL(sw_links_kernel, dim3(grid), dim3(TB), 0, st, (const uint32_t*)ks, (const uint32_t*)vs, n, after, f0);
  hipLaunchKernelGGL(sw_selfend_kernel, dim3(grid), dim3(TB), 0, st, f0, n);
  HIPCHK(hipGetLastError());
  // first the walks (see sw_walk_scatter_kernel); `out` is a scratch list, so a second attempt simply writes it again
  {
    unsigned h = 0;
    HIPCHK(hipMemsetAsync(changed, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(sw_walk_scatter_kernel, dim3(grid), dim3(TB), 0, st, (const uint32_t*)ctx->order32, (const uint32_t*)dpos, (const uint32_t*)after,
                       (const uint32_t*)f0, n, out, ctx->order, changed);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&h, changed, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));                  // (also: the caller's pos buffer is free from here on)
    if (!h) {
      HIPCHK(hipMemcpyAsync(ctx->order32, out, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
      return MFX_OK;
    }
  }
  uint32_t *fin = f0, *fout = f1;
  for (int round = 0; round < 40; round++) {
    // F <- F o F; the flag is that of the LAST of three rounds (a round that changes nothing is followed by rounds that change nothing)
    unsigned h = 0;
    for (int k = 0; k < 3; k++) {
      HIPCHK(hipMemsetAsync(changed, 0, sizeof(unsigned), st));
      hipLaunchKernelGGL(sw_jump_kernel, dim3(grid), dim3(TB), 0, st, (const uint32_t*)fin, fout, n, changed);
      std::swap(fin, fout);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&h, changed, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));                  // (also: the caller's pos buffer is free from here on)
    if (!h) break;
    NEED(round < 39, MFX_E_STATE, "mfx_sgd_apply_swaps32: the chains of the swaps did not close (pos[i] > i somewhere?)");
  }
  hipLaunchKernelGGL(sw_scatter_kernel, dim3(grid), dim3(TB), 0, st, (const uint32_t*)ctx->order32, (const uint32_t*)dpos, (const uint32_t*)after,
                     (const uint32_t*)fin, n, out, ctx->order);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(ctx->order32, out, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
  return MFX_OK;
}

extern "C" int mfx_debug_order32(mfx_ctx* ctx, uint32_t* out, int64_t cap, int64_t* n) {
  if (!ctx) return MFX_E_ARG;
  NEED(n, MFX_E_ARG, "mfx_debug_order32: n NULL");
  *n = (ctx->order32 && ctx->order32_valid) ? ctx->order_n : 0;
  if (!out) return MFX_OK;
  NEED(cap >= *n, MFX_E_ARG, "mfx_debug_order32: cap too small");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (*n) HIPCHK(hipMemcpy(out, ctx->order32, sizeof(uint32_t) * (size_t)*n, hipMemcpyDeviceToHost));
  return MFX_OK;
}
