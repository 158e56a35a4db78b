// mfx_internal.h -- context and helpers shared by the HIP translation units.
#ifndef MFX_INTERNAL_H_
#define MFX_INTERNAL_H_

#include <hip/hip_runtime.h>

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
  int32_t* ulist = nullptr;  // user list for MFX_SGD_USERS
  int64_t ulist_cap = 0;
  // MFX_SGD_TILED: slot lists (sgd_slots.hip owns the type)
  void* slots = nullptr;
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
  float* comm_tmp = nullptr;
  size_t comm_tmp_cap = 0;              // floats
  mfx_reduce_fn ext_reduce = nullptr;   // caller-supplied all-reduce on a host copy (mfx_comm_init_external)
  void* ext_user = nullptr;
  void* ext_stage = nullptr;            // pinned staging buffer for it
  size_t ext_stage_bytes = 0;
  double* gcol = nullptr;               // sharded runs: ratings per item over ALL ranks [train ncols]
  float* als_global = nullptr;          // sharded ALS: per-item (A, b) summed over ranks

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

template <typename T>
static inline int dev_alloc(mfx_ctx* ctx, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  HIPCHK(hipMalloc((void**)p, n * sizeof(T)));
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


// The CCD++ pass loop of one 16-lane group (lane j): for the segments s = s_first, s_first + s_step, ... < s_last,
// (num, den) = (sum res*o, sum o*o) over the entries [seg_beg[s], seg_end[s]), o = other[ind[t]] -- float products, double
// accumulation (modelMF.cpp:1069-1070, 1085-1086) -- handed lane-wise to fin(s, num, den), which finishes with its
// butterfly over the group: a fixed association.
// Every lane loads 16 ALIGNED bytes of indices and of residuals per trip (64 entries per group per trip; entries in front of
// the segment or behind it are masked to +0.0).  Segments are short (C4: 200 entries per row, 270 per (strip, column)
// piece), so the loop is software-pipelined ACROSS segments: while segment s is summed, the first trip of the next one and
// the bounds of the one after are already in flight -- a group never sits idle on a dependent bounds -> data round trip.
struct MfxCcdTrip {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef int i4 __attribute__((ext_vector_type(4)));
  i4 x;
  f4 r;
  __device__ __forceinline__ void load(const float* __restrict__ res, const int32_t* __restrict__ ind, int64_t t, int64_t e, int64_t nmax) {
    x = i4{0, 0, 0, 0};
    r = f4{0.0f, 0.0f, 0.0f, 0.0f};
    if (t >= e) return;
    if (t + 4 <= nmax) {
      x = *(const i4*)(ind + t);
      r = *(const f4*)(res + t);
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (t + q < nmax) { x[q] = ind[t + q]; r[q] = res[t + q]; }
    }
  }
  __device__ __forceinline__ void consume(const float* other, int64_t t, int64_t b, int64_t e, double& num, double& den) const {
    float o[4], rr[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const bool ok = t + q >= b && t + q < e;
      o[q] = other[ok ? x[q] : 0];
      o[q] = ok ? o[q] : 0.0f;
      rr[q] = ok ? r[q] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) { num += (double)(rr[q] * o[q]); den += (double)(o[q] * o[q]); }
  }
};
template <class Fin>
__device__ __forceinline__ void mfx_ccd_pass_loop(const int64_t* __restrict__ seg_beg, const int64_t* __restrict__ seg_end, int64_t s_first,
                                                  int64_t s_last, int64_t s_step, const float* __restrict__ res,
                                                  const int32_t* __restrict__ ind, const float* other, int j, int64_t nmax, Fin&& fin) {
  int64_t s = s_first;
  if (s >= s_last) return;
  int64_t b = seg_beg[s], e = seg_end[s];
  int64_t sn = s + s_step, bn = 0, en = 0;
  if (sn < s_last) { bn = seg_beg[sn]; en = seg_end[sn]; }
  MfxCcdTrip cur, nxt, more;
  cur.load(res, ind, (b & ~(int64_t)3) + 4 * j, e, nmax);
  for (;;) {
    const int64_t snn = sn + s_step;
    int64_t bnn = 0, enn = 0;
    if (snn < s_last) { bnn = seg_beg[snn]; enn = seg_end[snn]; }                        // bounds two segments ahead
    nxt.load(res, ind, (bn & ~(int64_t)3) + 4 * j, sn < s_last ? en : bn, nmax);      // first trip of the next segment
    double num = 0.0, den = 0.0;
    int64_t t = (b & ~(int64_t)3) + 4 * j;
    cur.consume(other, t, b, e, num, den);
    for (t += 64; t < e; t += 128) {          // long segments: two trips in flight
      cur.load(res, ind, t, e, nmax);
      more.load(res, ind, t + 64, e, nmax);
      cur.consume(other, t, b, e, num, den);
      more.consume(other, t + 64, b, e, num, den);
    }
    fin(s, num, den);
    if (sn >= s_last) break;
    s = sn; b = bn; e = en; cur = nxt;
    sn = snn; bn = bnn; en = enn;
  }
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
void mfx_slots_free_internal(mfx_ctx* ctx);
int mfx_launch_eval(mfx_ctx* ctx, const DevCSR& m, const float* U, const float* V,
                    int with_norms, mfx_eval_out* out);
int mfx_launch_eval2(mfx_ctx* ctx, const DevCSR& ma, int norms_a, const DevCSR& mb, int norms_b, const float* U, const float* V,
                     mfx_eval_out* out_a, mfx_eval_out* out_b);
int mfx_comm_free_internal(mfx_ctx* ctx);
int mfx_comm_allreduce(mfx_ctx* ctx, void* dev, size_t count, int dtype);   // sum over ranks, dtype 0 f32 / 1 f64
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
