// mfx_internal.h -- context and helpers shared by the HIP translation units.
#ifndef MFX_INTERNAL_H_
#define MFX_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "mfx.h"

struct DevCSR {
  int32_t nrows = 0, ncols = 0;
  int64_t nnz = 0;
  int64_t* rowptr = nullptr;
  int32_t* rowind = nullptr;
  float* rowval = nullptr;
  int32_t* rowid = nullptr;   // expanded row index per rating (CSR order)
  int64_t* colptr = nullptr;
  int32_t* colind = nullptr;
  float* colval = nullptr;
  bool present = false, has_col = false;
};

// Rows of the train matrix cut into segments of <= MFX_SEG ratings (rows sorted by
// length, longest first): the unit of work of the ALS and CCD++ row kernels, so that a
// 50k-rating item does not serialise a sweep.  side 0 = row view, 1 = column view.
constexpr int MFX_SEG = 1024;
struct RowSegs {
  int32_t* seg_row = nullptr;     // row of each segment
  int64_t* seg_beg = nullptr;
  int64_t* seg_end = nullptr;
  int32_t* seg_slab = nullptr;    // partial-result slot, or -1 when the row is a single segment
  int32_t* mrow = nullptr;        // rows with several segments
  int32_t* mrow_first = nullptr;  // first slot of such a row
  int32_t* mrow_n = nullptr;      // number of slots
  int64_t nseg = 0, nmrow = 0, nslab = 0;
  bool built = false;
};

// trip list of a CCD++ pass (see mfx_ccd_trip_loop)
struct MfxTrips { int32_t* q = nullptr; int32_t* pk = nullptr; int32_t* meta = nullptr; };
// one pass launch over a padded view (ccd_blocks.h)
struct MfxBlocks {
  int2* rec = nullptr;             // [workgroup][group][step]: x = first slot of the trip's pieces (-1: no trip), y = end mask
  int32_t* wg_t0 = nullptr;        // first trip of the workgroup's window (position / 128)
  int32_t* wg_n = nullptr;         // trips in it
  int32_t* wg_stride = nullptr;    // chunks of 64 trips from one step of the workgroup to its next (1: a contiguous window)
  int64_t* wg_rec = nullptr;       // where its records start
  int32_t* wg_tag = nullptr;       // the caller's tag (column view: the strip, -1 = the light region)
  int nwg = 0;
  int64_t nslots = 0;
};

struct ProfSlot {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  double ms = 0;
  int64_t launches = 0;
};

struct mfx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  DevCSR mat[3];
  int32_t nU = 0, nI = 0, K = 0, L = 0, C = 0, ld = 0;
  float *U = nullptr, *V = nullptr, *Ubest = nullptr, *Vbest = nullptr, *Vsync = nullptr;
  uint8_t *invU = nullptr, *invI = nullptr;
  bool have_invalid = false;
  int64_t n_invalid = -1;        // invalid users + items after mfx_compute_invalid (-1: unknown, treated as some)
  bool force_masks = false;      // mfx_eval_filtered has swapped its own masks in

  // epoch rating list (visiting order) and host-provided permutation
  int32_t *eu = nullptr, *ei = nullptr;
  float* er = nullptr;
  int64_t elist_n = 0, elist_cap = 0;
  uint64_t* order = nullptr;
  int64_t order_n = 0, order_cap = 0;
  uint32_t* order32 = nullptr;   // staging of mfx_sgd_set_order32
  int64_t order32_cap = 0;
  uint32_t* sw_buf = nullptr;    // mfx_sgd_apply_swaps32: eight arrays of order_n entries + the sort's workspace, kept between epochs
  size_t sw_cap = 0;             // entries per array
  char* sw_tmp = nullptr;
  size_t sw_tmp_bytes = 0;
  bool order32_valid = false;    // order32 holds the same list as order (mfx_sgd_set_order32 / mfx_sgd_apply_swaps32; not after the 64-bit call)
  int32_t* ulist = nullptr;  // user list for MFX_SGD_USERS
  int64_t ulist_cap = 0;
  // MFX_SGD_TILED: slot lists (sgd_slots.hip owns the type)
  void* slots = nullptr;
  int item_parts = 0;        // mfx_sgd_set_item_parts: > 1 = the multi-GPU rotation runs epochs part by part
  uint64_t train_gen = 0;    // bumped by every mfx_set_csr(MFX_MAT_TRAIN): what caches keyed on the train matrix compare
  // MFX_SGD_LEVELS: level lists (sgd_levels.hip owns the type)
  void* levels = nullptr;
  bool last_exact_flow = false;   // which schedule the last MFX_SGD_LEVELS epoch ran on
  void* flow = nullptr;       // its dataflow schedule (sgd_flow.hip owns the type)

  // reduction scratch
  double* red_d = nullptr;   // [blocks][4]
  int64_t* red_i = nullptr;  // [blocks]
  int red_blocks = 0;
  double* red_out = nullptr; // pinned host [8]

  // CCD++ state
  float *res_row = nullptr, *res_col = nullptr, *uk = nullptr, *vk = nullptr;
  float *uk_pend = nullptr, *vk_pend = nullptr;   // factor whose residual subtract is deferred
  bool ccd_active = false, ccd_pending = false;

  RowSegs segs[2];
  // ALS partial-Gramian slabs
  float* als_slabs = nullptr;
  int64_t als_slab_cap = 0;
  // CCD++ partial (num, denom) slots and expanded column ids
  double* ccd_part = nullptr;
  int64_t ccd_part_cap = 0;
  int32_t* colid = nullptr;
  // CCD++ row view, PADDED (ccd_blocks.h): row r at [ccd_rpos[r], ccd_rpos[r+1]), both multiples of 8; ccd_nnzp entries, a multiple of 128
  int64_t ccd_nnzp = 0;
  int64_t* ccd_rpos = nullptr;
  uint16_t* ccd_ind16 = nullptr;   // item ids, 16-bit when v_k fits LDS (at most 38 400 items) ...
  int32_t* ccd_ind32 = nullptr;    // ... else 32-bit; the padding entries carry nI (the +0.0 slot behind v_k)
  int32_t* ccd_rowid = nullptr;    // row of every padded entry
  MfxBlocks ccd_blocks;            // the row pass
  int32_t* ccd_rfirst = nullptr;   // [nU] first (num, den) slot of a row's pieces
  int32_t* ccd_rcnt = nullptr;     // [nU] how many (0: a row without ratings)
  int32_t* ccd_lrow = nullptr;     // rows with more than 32 pieces (finished by a 16-lane group each)
  int64_t ccd_nlrow = 0;
  void* ccd_cols = nullptr;   // strip-major column view (ccd_cols.hip owns the type)
  void* cd = nullptr;         // trainCCD state (cd.hip owns the type)
  void* als_wide = nullptr;   // ALS for K > 64 (als_wide.hip owns the type)
  float *ub = nullptr, *ib = nullptr, *ub_best = nullptr, *ib_best = nullptr;   // ModelMFBias vectors (sgd_bias.hip), or NULL
  bool bias_epoch = false;    // mfx_bias_epoch is driving the epoch-list machinery of mfx_sgd_epoch
  float* dimreg = nullptr;    // per-dimension regulariser [ld] of trainSGDParSVD (svd.hip), or NULL
  void* ifw = nullptr;        // rating weights of ModelInvPopMF (sgd_ifw.hip owns the type), or NULL
  uint64_t var_gen = 0;       // bumped whenever the rating weights / rank tables change
  int32_t *tmfd_u = nullptr, *tmfd_i = nullptr;   // ModelPoissonDropout: lambda per user / item, or NULL
  double* tmfd_exp = nullptr;                     // exp(-lambda), lambda = 0..K
  uint32_t tmfd_seed = 0;
  int2 *tmf_u = nullptr, *tmf_i = nullptr;   // (train frequency bits, truncated rank) per user / item (sgd_tmf.hip), or NULL

  // comm
  void* comm = nullptr;      // ncclComm_t
  int nranks = 1, rank = 0;
  bool comm_checked_parts = false, comm_checked_als = false;   // mfx_comm_check_replicas ran for the rotation / the sharded ALS sweep
  float* comm_tmp = nullptr;
  size_t comm_tmp_cap = 0;              // floats
  mfx_reduce_fn ext_reduce = nullptr;   // caller-supplied all-reduce on a host copy (mfx_comm_init_external)
  void* ext_user = nullptr;
  void* ext_stage = nullptr;            // pinned staging buffer for it
  size_t ext_stage_bytes = 0;
  double* gcol = nullptr;               // sharded runs: ratings per item over ALL ranks [train ncols]
  float* als_global = nullptr;          // sharded ALS: per-item (A, b) summed over ranks
  size_t als_global_cap = 0;            // floats

  bool prof_on = false;
  int prof_period = 1;       // > 1: only every prof_period-th SGD epoch records events
  int64_t prof_tick = 0;
  ProfSlot prof[MFX_K_COUNT];
};

int mfx_fail(mfx_ctx* ctx, int code, const char* fmt, ...);

#define HIPCHK(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return mfx_fail(ctx, e_ == hipErrorOutOfMemory ? MFX_E_OOM : MFX_E_HIP, "%s: %s", \
                      #expr, hipGetErrorString(e_));                                   \
  } while (0)

#define NEED(cond, code, ...)                            \
  do {                                                   \
    if (!(cond)) return mfx_fail(ctx, code, __VA_ARGS__); \
  } while (0)

// Records a start/stop HIP event pair around a launch on ctx->stream when
// profiling is on; mfx_prof_get() resolves the pairs.
struct ProfScope {
  mfx_ctx* ctx;
  int k;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(mfx_ctx* c, int kernel);
  ~ProfScope();
};

constexpr size_t MFX_ALLOC_PAD = 1024;     // bytes; a trip ends up to E - 1 = 127 entries behind its segment (508 bytes of a 4-byte array); checked
                                           // per trip list by mfx_trips_fit, not only sized here
template <typename T>
static inline int dev_alloc(mfx_ctx* ctx, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  HIPCHK(hipMalloc((void**)p, n * sizeof(T) + MFX_ALLOC_PAD));   // readable bytes behind every array: the streaming kernels load whole aligned
                                                                 // 64-entry trips and mask what lies outside (mfx_ccd_trip_loop)
  return MFX_OK;
}
template <typename T>
static inline void dev_free(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

// user / item block of the 8x8 tiling (MFX_SGD_TILED): a hash of the index, so the
// blocks are balanced in expectation and need no table.
__host__ __device__ static inline uint32_t mfx_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__host__ __device__ static inline int mfx_user_block(int32_t u) { return (int)(mfx_mix32((uint32_t)u * 0x9e3779b1U + 0x1234567U) & 7U); }
__host__ __device__ static inline int mfx_item_block(int32_t i) { return (int)(mfx_mix32((uint32_t)i * 0x85ebca6bU + 0x89abcdeU) & 7U); }

// Bijection on [0,n): alternating (unbalanced) Feistel network on ceil(log2 n) bits, cycle-walked
// into range (documented in DESIGN.md; tests re-check that the device lists are permutations).
__host__ __device__ static inline uint64_t mfx_feistel(uint64_t x, int abits, int bbits, uint32_t k0, uint32_t k1) {
  const uint32_t maskA = (abits >= 32) ? 0xffffffffU : ((1U << abits) - 1U);
  const uint32_t maskB = (bbits >= 32) ? 0xffffffffU : ((1U << bbits) - 1U);
  uint32_t Lh = (uint32_t)(x >> bbits) & maskA, Rh = (uint32_t)x & maskB;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    if ((r & 1) == 0) Lh ^= mfx_mix32(Rh * 0x9e3779b1U + k0 + (uint32_t)r * 0x85ebca6bU) & maskA;
    else Rh ^= mfx_mix32(Lh * 0xc2b2ae35U + k1 + (uint32_t)r * 0x27d4eb2fU) & maskB;
  }
  return ((uint64_t)Lh << bbits) | Rh;
}
__host__ __device__ static inline int64_t mfx_perm_index(int64_t t, int64_t n, int abits, int bbits, uint32_t k0, uint32_t k1) {
  uint64_t y = mfx_feistel((uint64_t)t, abits, bbits, k0, k1);
  while (y >= (uint64_t)n) y = mfx_feistel(y, abits, bbits, k0, k1);
  return (int64_t)y;
}


// The CCD++ pass of one 16-lane group (lane j): (num, den) = (sum res*o, sum o*o) over the entries of a segment,
// o = other[ind[t]] -- float products, double accumulation (modelMF.cpp:1069-1070, 1085-1086) -- handed lane-wise to
// fin(packed, meta, num, den), which finishes with its butterfly over the group: a fixed association.
//
// The passes are HBM-streaming and the segments are short (C4: 200 entries per row, 270 per (strip, column) piece), so what
// matters is that loads stay in flight across segment ends.  The host flattens the segments into TRIPS (mfx_trips_*): a trip
// is 64 entries of one segment starting at an aligned position; its 16-byte record holds the position, where the segment
// lies inside it, and what fin needs.  A group works through a contiguous range of the trip list -- ranges are cut at
// segment boundaries with equal numbers of trips, i.e. equal work.  The kernel loop has no data-dependent control flow
// around its loads: step k issues the data loads of trip k+2 (16 ALIGNED bytes of residuals and 8 or 16 of indices per lane,
// unconditionally; entries in front of the segment or behind it are masked when they are consumed), sums trip k, and issues
// the record load of trip k+4.  It is unrolled four steps by hand: rotating the buffers with copies would make the compiler
// wait for the loads it has just issued (measured on the first version: every wait was vmcnt(0), the waves were parked 80 %
// of the time and the pass ran at 3.4 TB/s whatever the bytes per entry).
// Arrays: every device allocation has MFX_ALLOC_PAD readable bytes behind it (dev_alloc); the residual pad must hold finite
// values (mfx_ccdpp_begin zeroes it), because a masked entry still multiplies its residual -- by the +0.0 it gathers at `zero`, an
// index at which `other` holds +0.0 (the callers keep such a slot behind the vector): the products are +-0 and leave the
// double sums untouched; one compare and one select per entry instead of three selects.  (A NaN residual next to a segment
// would leak into it; residuals are finite unless the model has diverged, and then Model::isTerminateModel's guard fires.)
// IdxT: int32_t, or uint16_t where the gathered vector has at most 65536 entries (6 instead of 8 bytes per entry and trip).
template <typename IdxT> struct MfxIdx4;
template <> struct MfxIdx4<int32_t> {
  typedef int raw __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ int get(raw v, int q) { return v[q]; }
};
template <> struct MfxIdx4<uint16_t> {
  typedef unsigned raw __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ int get(raw v, int q) { return (int)((q & 1) ? (v[q >> 1] >> 16) : (v[q >> 1] & 0xffffu)); }
};
// four consecutive ids as ints (streaming kernels other than the passes)
struct MfxCcdTrip {
  typedef int i4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ i4 load4(const int32_t* p) { return *(const i4*)p; }
  static __device__ __forceinline__ i4 load4(const uint16_t* p) {
    const MfxIdx4<uint16_t>::raw v = *(const MfxIdx4<uint16_t>::raw*)p;
    return i4{MfxIdx4<uint16_t>::get(v, 0), MfxIdx4<uint16_t>::get(v, 1), MfxIdx4<uint16_t>::get(v, 2), MfxIdx4<uint16_t>::get(v, 3)};
  }
};
// sum over the 16 lanes of a DPP row, every lane gets it; levels xor 1, 2 (quad_perm), then the partner quad / half
// (row_half_mirror, row_mirror: after the lower levels the lanes of a quad / half hold the same bits) -- no LDS round trips
template <int CTRL>
__device__ __forceinline__ double mfx_dpp_f64(double x) {
  // (the form without an "old" operand: these controls read no lane outside the row, and an old value costs a move each)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double mfx_row16_sum(double v) {
  v += mfx_dpp_f64<0xB1>(v);
  v += mfx_dpp_f64<0x4E>(v);
  v += mfx_dpp_f64<0x141>(v);
  v += mfx_dpp_f64<0x140>(v);
  return v;
}

// Result stores of the pass kernels, issued as inline assembly ON PURPOSE: gfx9 counts loads and stores on one counter
// (vmcnt), and with a store pending the compiler must assume out-of-order completion and turns every later wait for a load
// into vmcnt(0) -- the whole prefetch pipeline drains at each segment end.  A store the compiler does not see keeps its load
// counting exact, and the counting stays SAFE: loads complete in order among themselves, so "at most N operations outstanding"
// with N later loads issued still implies the awaited load has arrived, whatever the stores do.  Nothing reads these
// locations again inside the kernel; the end of the kernel makes them visible.
__device__ __forceinline__ void mfx_store_unseen(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void mfx_store_unseen(double* p, double a, double b) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 v = {(unsigned)__double2loint(a), (unsigned)__double2hiint(a), (unsigned)__double2loint(b), (unsigned)__double2hiint(b)};
  // (s_nop 0: a store of more than 64 bits needs one wait state before a VALU may overwrite its data registers, and the
  // hazard recognizer does not look inside an asm statement -- tests/test_trips_cpu.py checks the disassembly)
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}


// trip record: x = position / 4, y = a | i << 5 | len << 10 | last << 21 (a = segment start - trip start of its first trip < 32,
// i = trip number inside the segment, len = entries of the segment <= 2047), z = meta.  On the device the three fields are three
// arrays (MfxTrips): loaded as one 12-byte tuple, the register allocator split the tuple over the loop-carried registers with
// copies at the loop end, and a copy of a value that has just been requested is a wait for it.

constexpr int MFX_TRIP_LAST = 1 << 21;
// A trip starts on a 128-byte line of the residuals (32 entries; MFX_CCD_ALIGN overrides at build time): the trips of a
// segment then share no line, and a line is requested from L2 once per trip instead of once per trip that touches part of it.
#ifndef MFX_CCD_ALIGN
#define MFX_CCD_ALIGN 32
#endif
constexpr int64_t MFX_TRIP_ALIGN = MFX_CCD_ALIGN;
// Entries per lane and trip: 8 -- two 16-byte residual loads and ONE 16-byte load of eight 16-bit ids per lane, 128 entries per
// trip.  (With 4 per lane the id load was 8 bytes and there were twice the records: vector memory instructions per entry are
// what these kernels pay for.)  MFX_CCD_EPL=4 at build time gives the 64-entry trips back.
// The row view (segments of ~200 entries) takes 8, the column view (many (strip, column) pieces of a few entries, where a step
// costs its instructions whether its lanes are live or not) 4: measured both ways, row pass 0.148 -> 0.129 ms, column pass
// 0.167 -> 0.175.  E = entries per trip = 16 lanes x entries per lane is a parameter of the builders and of the loop.
struct MfxSeg { int64_t b, e; int32_t meta; };
static inline int mfx_seg_trips(const MfxSeg& g, int E) { return (int)std::max<int64_t>(1, (g.e - (g.b & ~(MFX_TRIP_ALIGN - 1)) + E - 1) / E); }
// returns the position behind the last entry the segment's trips LOAD (whole trips are loaded, what lies outside the segment is
// masked): up to E - 1 entries behind the segment's end
static inline int64_t mfx_trips_append(std::vector<int4>& trips, const MfxSeg& g, int E) {
  const int64_t t0 = g.b & ~(MFX_TRIP_ALIGN - 1);
  const int a = (int)(g.b & (MFX_TRIP_ALIGN - 1)), len = (int)(g.e - g.b), ntr = mfx_seg_trips(g, E);
  for (int i = 0; i < ntr; i++) {
    int4 r;
    r.x = (int)(uint32_t)((t0 + E * (int64_t)i) >> 2);
    r.y = a | (i << 5) | (len << 10) | (i == ntr - 1 ? MFX_TRIP_LAST : 0);
    r.z = g.meta;
    r.w = 0;
    trips.push_back(r);
  }
  return t0 + (int64_t)E * ntr;
}
// The invariant the allocation pad exists for (and that a 16-byte pad broke in round 2, gpurun_out/r2_ccd14.log: a trip of the
// last row's segment read up to 63 entries = 252 bytes behind hipMalloc's end; the fault surfaced at the next synchronisation,
// in mfx_eval): an array of `n` entries of `elem` bytes that trips up to position `max_end` are loaded from must have them.
static inline bool mfx_trips_fit(int64_t max_end, int64_t n, size_t elem) {
  return max_end <= n || (size_t)(max_end - n) * elem <= MFX_ALLOC_PAD;
}
// Lay the segments segs[k0, k1) -- given in MEMORY order -- out for nwg workgroups of gpw groups each: the workgroups get
// consecutive runs of segments with equal numbers of trips; inside a workgroup every next segment goes to the group with
// the fewest trips so far.  The groups of a workgroup therefore advance through ONE window of memory together (with a range
// of its own per group, 32 768 groups were 32 768 streams of 256-byte reads: every access opened a DRAM row of its own and
// the pass stayed at 3.5 TB/s whatever else was improved), and they finish together.  The trips of a group are consecutive
// in the list; gptr receives nwg * gpw range starts (the caller appends the end of its last range).
// *max_end (if given) is raised to the position behind the last entry any trip loads (mfx_trips_fit).
static inline void mfx_trips_layout(const std::vector<MfxSeg>& segs, size_t k0, size_t k1, int nwg, int gpw, int E, std::vector<int4>& trips,
                                    std::vector<int32_t>& gptr, int64_t* max_end = nullptr) {
  int64_t total = 0;
  for (size_t k = k0; k < k1; k++) total += mfx_seg_trips(segs[k], E);
  std::vector<std::vector<int32_t>> mine((size_t)gpw);
  std::vector<int64_t> load((size_t)gpw);
  size_t k = k0;
  int64_t done = 0;
  for (int w = 0; w < nwg; w++) {
    const int64_t want = total * (w + 1) / nwg;
    for (auto& v : mine) v.clear();
    std::fill(load.begin(), load.end(), 0);
    int g = 0;
    while (k < k1 && (done < want || w == nwg - 1)) {
      // least loaded group, scanning from the one after the last choice (ties go round-robin)
      int best = g;
      for (int q = 0; q < gpw; q++) { const int c = (g + q) % gpw; if (load[(size_t)c] < load[(size_t)best]) best = c; }
      const int nt = mfx_seg_trips(segs[k], E);
      mine[(size_t)best].push_back((int32_t)k);
      load[(size_t)best] += nt;
      done += nt;
      g = (best + 1) % gpw;
      k++;
    }
    for (int q = 0; q < gpw; q++) {
      gptr.push_back((int32_t)trips.size());
      for (int32_t sidx : mine[(size_t)q]) {
        const int64_t end = mfx_trips_append(trips, segs[(size_t)sidx], E);
        if (max_end && end > *max_end) *max_end = end;
      }
    }
  }
}
static inline void mfx_trips_free(MfxTrips& t) { dev_free(t.q); dev_free(t.pk); dev_free(t.meta); }
static inline int mfx_trips_upload(mfx_ctx* ctx, const std::vector<int4>& trips, MfxTrips* out) {
  mfx_trips_free(*out);
  std::vector<int32_t> a(trips.size()), b(trips.size()), c(trips.size());
  for (size_t k = 0; k < trips.size(); k++) { a[k] = trips[k].x; b[k] = trips[k].y; c[k] = trips[k].z; }
  int rc;
  if ((rc = dev_alloc(ctx, &out->q, a.size())) || (rc = dev_alloc(ctx, &out->pk, a.size())) || (rc = dev_alloc(ctx, &out->meta, a.size()))) return rc;
  if (!a.empty()) {
    HIPCHK(hipMemcpy(out->q, a.data(), sizeof(int32_t) * a.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out->pk, b.data(), sizeof(int32_t) * a.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out->meta, c.data(), sizeof(int32_t) * a.size(), hipMemcpyHostToDevice));
  }
  return MFX_OK;
}

// Diagnostic builds (scripts/exp_ccd.sh; results are WRONG on purpose): what the pass time does when one part is taken out.
// bit 0: no gathers (o = 1)  bit 1: no data loads  bit 2: no segment ends (fin never runs)  bit 3: no record loads
// bit 4: float sums instead of double  bit 5: memory operations only (no masks, gathers, doubles)  bit 6: synthetic records, trip n
// = entries [64 n, 64 n + 64) (a clean stream per group).  0 (the product): nothing of
// this is compiled.
#ifndef MFX_CCD_EXP
#define MFX_CCD_EXP 0
#endif
// The sums of a finished segment go to part[2 * meta], part[2 * meta + 1] -- ONE STEP LATER: the hardware counts the store on
// the same counter as the loads, so a store issued at the end of a step sits between the next step's wait and the data it waits
// for (the wait then lasts until the store is acknowledged or a younger load returns: about 1 us per segment end, which was a
// third of the pass).  Issued a step later it is older than everything that step waits for.
template <bool BUF, int EPL, typename IdxT>
__device__ __forceinline__ void mfx_ccd_trip_loop(const MfxTrips trips, int n0, int n1, const float* __restrict__ res,
                                                  const IdxT* __restrict__ ind, uint32_t res_bytes, const float* other, int zero, int j,
                                                  double* __restrict__ part) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef typename MfxIdx4<IdxT>::raw raw_t;
  if (n0 >= n1) return;
  static_assert(EPL == 4 || EPL == 8, "entries per lane");
  constexpr int H = EPL / 4;                      // 4-entry quads per lane and trip
  constexpr int MFX_TRIP_EPL = EPL, MFX_TRIP_E = 16 * EPL;
  struct Data { raw_t x[H]; f4 r[H]; };
  struct Rec { int x, y, z; };
  const int exp_q0 = (MFX_CCD_EXP & 8) ? trips.q[n0] : 0, exp_q1 = (MFX_CCD_EXP & 8) ? trips.q[n1 - 1] : 0, exp_m = (MFX_CCD_EXP & (8 | 64)) ? trips.meta[n0] : 0;
  // Records are fetched four trips at a time: lane j of the group loads record base + (j & 3) (three loads per FOUR steps
  // instead of per step: small loads cost the vector memory pipe as much as wide ones -- scripts/stream_probe.hip pattern 5),
  // a step takes its record from lane s of the row with a DPP row_share.
  auto rec = [&](int base) {                                                // behind the range: the last record again (masked below)
    const int n = base + (j & 3);
    const int nc = n < n1 ? n : n1 - 1;
    Rec r;
    if (MFX_CCD_EXP & 64) { r.x = (int)((unsigned)n % (res_bytes / (4u * MFX_TRIP_E) - 2u)) * (MFX_TRIP_E / 4); r.y = (MFX_TRIP_E << 10) | ((n & 3) == 3 ? MFX_TRIP_LAST : 0); r.z = exp_m; return r; }   // trip n reads entries [E n, E n + E): every line once, one stream per group
    if (MFX_CCD_EXP & 8) { r.x = min(exp_q0 + (MFX_TRIP_E / 4) * (n - n0), exp_q1); r.y = (MFX_TRIP_E << 10) | ((n & 3) == 3 ? MFX_TRIP_LAST : 0); r.z = exp_m; return r; }   // in-bounds: a group's trips ascend in memory
    r.x = trips.q[nc]; r.y = trips.pk[nc]; r.z = trips.meta[nc];
    return r;
  };
  // BUF (arrays below 4 GB): buffer loads, and a lane whose four entries all lie outside the segment asks for an offset behind
  // the buffer -- the hardware returns zeros without a memory access.  A segment of 208 entries is covered by four 64-entry
  // trips; without this the vector memory pipe moves 256 entries for it (measured: 1.23 x the arrays).
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)res, 0, BUF ? (int)res_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ind = __builtin_amdgcn_make_buffer_rsrc((void*)ind, 0, BUF ? (int)(res_bytes / 4 * sizeof(IdxT)) : 0, 0x00020000);
  auto data = [&](const Rec& r, bool live) {      // r: the record of the trip, the same in all lanes of the group
    Data d;
    const int64_t t = ((int64_t)(uint32_t)r.x << 2) + MFX_TRIP_EPL * j;
    if (MFX_CCD_EXP & 2) {
#pragma unroll
      for (int h = 0; h < H; h++) { d.x[h] = raw_t{}; d.x[h][0] = r.x; d.r[h] = f4{1.0f, 2.0f, 3.0f, 4.0f}; }
      return d;
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    if (BUF) {
      const unsigned len = ((unsigned)r.y >> 10) & 0x7ffu;
      const unsigned rel = (unsigned)(MFX_TRIP_E * ((r.y >> 5) & 31) - (r.y & 31) + MFX_TRIP_EPL * j);
      const unsigned t32 = (unsigned)t;
      bool any = false;
#pragma unroll
      for (int h = 0; h < H; h++) {
        const bool need = live && rel + 4u * h + 3u < len + 3u;             // some entry of the quad lies in [0, len)
        any |= need;
        d.r[h] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, need ? (t32 + 4u * h) * 4u : 0xfffffff0u, 0, 0));
        if constexpr (sizeof(IdxT) == 4) d.x[h] = __builtin_bit_cast(raw_t, __builtin_amdgcn_raw_buffer_load_b128(rs_ind, need ? (t32 + 4u * h) * 4u : 0xfffffff0u, 0, 0));
        else if constexpr (H == 1) d.x[h] = __builtin_bit_cast(raw_t, __builtin_amdgcn_raw_buffer_load_b64(rs_ind, need ? t32 * 2u : 0xfffffff0u, 0, 0));
      }
      if constexpr (sizeof(IdxT) == 2 && H == 2) {       // eight 16-bit ids: ONE 16-byte load
        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_ind, any ? t32 * 2u : 0xfffffff0u, 0, 0);
        d.x[0] = raw_t{v[0], v[1]};
        d.x[1] = raw_t{v[2], v[3]};
      }
      return d;
    }
#pragma unroll
    for (int h = 0; h < H; h++) d.r[h] = *(const f4*)(res + t + 4 * h);
    if constexpr (sizeof(IdxT) == 2 && H == 2) {
      const u4 v = *(const u4*)(ind + t);
      d.x[0] = raw_t{v[0], v[1]};
      d.x[1] = raw_t{v[2], v[3]};
    } else {
#pragma unroll
      for (int h = 0; h < H; h++) d.x[h] = *(const raw_t*)(ind + t + 4 * h);
    }
    return d;
  };
  double num = 0.0, den = 0.0, pnum = 0.0, pden = 0.0;
  int pslot = -1;                 // pending result (lane j == 0 of the group)
  // record S of a batch (lane S of the row), the same in all lanes afterwards
#define MFX_REC_OF(BATCH, S)                                                                                              \
  Rec{__builtin_amdgcn_mov_dpp(BATCH.x, 0x150 + (S), 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(BATCH.y, 0x150 + (S), 0xF, 0xF, true), \
      __builtin_amdgcn_mov_dpp(BATCH.z, 0x150 + (S), 0xF, 0xF, true)}
  // one step: the prefetch (data of n + 2; its batch was requested at least four steps ago), the gathers of trip n, the sums
#define MFX_TRIP_STEP(N, CURB, S, DC, NEXTB, NEXTS, DN)                                                                         \
  {                                                                                                                       \
    if (pslot >= 0) { mfx_store_unseen(part + 2 * (int64_t)pslot, pnum, pden); pslot = -1; }                              \
    DN = data(MFX_REC_OF(NEXTB, NEXTS), (N) + 2 < n1);   /* first: the wave then waits for trip n with n + 1 AND n + 2 in flight */ \
    const Rec rc = MFX_REC_OF(CURB, S);                                                                                   \
    const int pk = (N) < n1 ? rc.y : 0;                                                                                   \
    const unsigned len = ((unsigned)pk >> 10) & 0x7ffu;                                                                   \
    const unsigned rel = (unsigned)(MFX_TRIP_E * ((pk >> 5) & 31) - (pk & 31) + MFX_TRIP_EPL * j);                        \
    const int meta = rc.z;                                                                                                \
    _Pragma("unroll") for (int h = 0; h < H; h++) {                                                                       \
      float o[4];                                                                                                         \
      _Pragma("unroll") for (int q = 0; q < 4; q++)                                                                       \
        o[q] = (MFX_CCD_EXP & 1) ? (rel + (unsigned)(4 * h + q) < len ? __int_as_float(MfxIdx4<IdxT>::get(DC.x[h], q) | 0x3f800000) : 0.0f) \
                                 : other[rel + (unsigned)(4 * h + q) < len ? MfxIdx4<IdxT>::get(DC.x[h], q) : zero];       \
      const f4 rr = DC.r[h];                                                                                              \
      if (MFX_CCD_EXP & 32) { /* memory operations only: the loaded values are summed without masks, gathers or doubles */ \
        float fs = rr[0] + rr[1] + rr[2] + rr[3] + __int_as_float(MfxIdx4<IdxT>::get(DC.x[h], 0) + MfxIdx4<IdxT>::get(DC.x[h], 3)); \
        num += (double)fs;                                                                                                \
      } else if (MFX_CCD_EXP & 16) {                                                                                      \
        float fn = 0.0f, fd = 0.0f;                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; q++) { fn += rr[q] * o[q]; fd += o[q] * o[q]; }                          \
        num += (double)fn; den += (double)fd;                                                                             \
      } else {                                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 4; q++) { num += (double)(rr[q] * o[q]); den += (double)(o[q] * o[q]); }  \
      }                                                                                                                   \
    }                                                                                                                     \
    if ((pk & MFX_TRIP_LAST) && !(MFX_CCD_EXP & 4)) {                                                                     \
      pnum = mfx_row16_sum(num); pden = mfx_row16_sum(den);                                                               \
      pslot = j == 0 ? meta : -1;                                                                                         \
      num = 0.0; den = 0.0;                                                                                               \
    }                                                                                                                     \
  }
  Rec ra = rec(n0), rb = rec(n0 + 4);        // batches: trips n .. n + 3 and n + 4 .. n + 7
  Data d0 = data(MFX_REC_OF(ra, 0), true), d1 = data(MFX_REC_OF(ra, 1), n0 + 1 < n1), d2, d3;
  // eight steps per turn, the two batches trading places: a batch is re-loaded in place once its last step is over (a copy
  // `ra = rb` makes the compiler load the new batch into temporaries and wait for them at the loop end)
  for (int n = n0; n < n1; n += 8) {
    MFX_TRIP_STEP(n, ra, 0, d0, ra, 2, d2)
    MFX_TRIP_STEP(n + 1, ra, 1, d1, ra, 3, d3)
    MFX_TRIP_STEP(n + 2, ra, 2, d2, rb, 0, d0)
    MFX_TRIP_STEP(n + 3, ra, 3, d3, rb, 1, d1)
    ra = rec(n + 8);
    MFX_TRIP_STEP(n + 4, rb, 0, d0, rb, 2, d2)
    MFX_TRIP_STEP(n + 5, rb, 1, d1, rb, 3, d3)
    MFX_TRIP_STEP(n + 6, rb, 2, d2, ra, 0, d0)
    MFX_TRIP_STEP(n + 7, rb, 3, d3, ra, 1, d1)
    rb = rec(n + 12);
  }
  if (pslot >= 0) mfx_store_unseen(part + 2 * (int64_t)pslot, pnum, pden);
#undef MFX_TRIP_STEP
#undef MFX_REC_OF
}

static inline void mfx_tree_shape(int K, int* L, int* C) {
  if (K <= 16) { *L = 4; *C = 1; }
  else if (K <= 32) { *L = 8; *C = 1; }
  else { *L = 16; *C = (K + 63) / 64; }
}

// kernels launched from other translation units
int mfx_launch_sgd(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count);
int mfx_launch_sgd_users(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t nusers);
int mfx_launch_sgd_tiled(mfx_ctx* ctx, const mfx_sgd_opts* o);
int mfx_launch_sgd_ifw(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count);      // sgd_ifw.hip
void mfx_ifw_free_internal(mfx_ctx* ctx);
// (freq, score) pairs and rhoRMS of the installed weights (sgd_ifw.hip)
void mfx_ifw_tables(mfx_ctx* ctx, const float2** ua, const float2** ia, float* rho);
int mfx_launch_sgd_tmf(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count);      // sgd_tmf.hip
void mfx_tmf_free_internal(mfx_ctx* ctx);
int mfx_launch_sgd_dimreg(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count);   // svd.hip
int mfx_launch_sgd_levels(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count);   // sgd_levels.hip
void mfx_levels_free_internal(mfx_ctx* ctx);
int mfx_launch_sgd_flow(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count);     // sgd_flow.hip
bool mfx_flow_usable(const mfx_ctx* ctx, int64_t count);
int mfx_launch_bias_flow(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count);   // sgd_flow.hip
int mfx_launch_sgd_bias(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count);    // sgd_bias.hip
void mfx_bias_free_internal(mfx_ctx* ctx);
bool mfx_flow_info(mfx_ctx* ctx, int64_t info[4], double* prep_ms);
void mfx_flow_free_internal(mfx_ctx* ctx);
int mfx_slots_materialise_order(mfx_ctx* ctx);
int mfx_slots_check_abort(mfx_ctx* ctx);      // sticky abort flag of the tiled schedule's drain (sgd_slots.hip)
void mfx_slots_free_internal(mfx_ctx* ctx);
int mfx_launch_eval(mfx_ctx* ctx, const DevCSR& m, const float* U, const float* V,
                    int with_norms, mfx_eval_out* out);
int mfx_launch_eval2(mfx_ctx* ctx, const DevCSR& ma, int norms_a, const DevCSR& mb, int norms_b, const float* U, const float* V,
                     mfx_eval_out* out_a, mfx_eval_out* out_b);
int mfx_comm_free_internal(mfx_ctx* ctx);
// Every rank holds the same bytes in buf[0 .. n)?  A 31-bit checksum of the buffer, all-reduced: N x own == sum on every rank.
// MFX_E_COMM with `what` in the message otherwise (comm.hip; the first all-gather of item parts and the first sharded ALS item
// sweep of a communicator run it: their rank-dependent indexing is invisible on one rank).
int mfx_comm_check_replicas(mfx_ctx* ctx, const float* buf, size_t n, const char* what);
int mfx_comm_allreduce(mfx_ctx* ctx, void* dev, size_t count, int dtype);   // sum over ranks, dtype 0 f32 / 1 f64
bool mfx_comm_has_rccl(const mfx_ctx* ctx);
int mfx_comm_reduce_scatter(mfx_ctx* ctx, float* buf, size_t count);            // in place, slices of `count` floats (RCCL only)
int mfx_comm_allgather(mfx_ctx* ctx, const float* mine, float* all, size_t count);
static inline bool mfx_sharded(const mfx_ctx* ctx);
// ratings per item summed over the ranks (device, cached until the train matrix changes)
int mfx_comm_global_col_counts(mfx_ctx* ctx, const double** out);
void mfx_comm_drop_col_counts(mfx_ctx* ctx);
void mfx_ccd_free_internal(mfx_ctx* ctx);
void mfx_als_free_internal(mfx_ctx* ctx);
void mfx_als_wide_free_internal(mfx_ctx* ctx);
int mfx_als_wide_half_sweep(mfx_ctx* ctx, int side, float reg);
void mfx_segs_free_internal(mfx_ctx* ctx);
// CCD++ column view (ccd_cols.hip)
int mfx_ccd_cols_build(mfx_ctx* ctx);
void mfx_ccd_cols_free(mfx_ctx* ctx);
int mfx_ccd_cols_pass(mfx_ctx* ctx, const float* uk, float* vk, float reg, float freq_thresh, int k);
int mfx_ccd_cols_resid(mfx_ctx* ctx, int mode, const float* uk0, const float* vk0, const float* uk1, const float* vk1);
int mfx_ccd_cols_export(mfx_ctx* ctx, float* host_out);
// column view of a matrix built on the device (setup.hip)
int mfx_build_col_index_device(mfx_ctx* ctx, DevCSR& m);
// map[d] = CSR position of the d-th entry of the (stable) column view (setup.hip); caller frees
int mfx_build_c2r_map_device(mfx_ctx* ctx, const DevCSR& m, uint32_t** map);
void mfx_cd_free_internal(mfx_ctx* ctx);
int mfx_get_segments(mfx_ctx* ctx, int side, RowSegs** out);

static inline bool mfx_sharded(const mfx_ctx* ctx) { return ctx->comm != nullptr || ctx->ext_reduce != nullptr; }

#endif
