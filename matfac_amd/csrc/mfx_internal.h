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
  float* uk_init = nullptr;                       // u_k as extracted (the fused first sweep's add-back uses it after the row pass has moved u_k on)
  float2* ccd_rpair = nullptr;                    // (u_pend[r], u_k[r]) per user: what a lane of the fused first sweep gathers for its row
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

constexpr size_t MFX_ALLOC_PAD = 1024;     // readable bytes behind every device array (the streaming kernels load whole 16-byte pieces)
template <typename T>
static inline int dev_alloc(mfx_ctx* ctx, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  HIPCHK(hipMalloc((void**)p, n * sizeof(T) + MFX_ALLOC_PAD));
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


// four consecutive ids as ints (the elementwise CCD++ kernels)
struct MfxCcdTrip {
  typedef int i4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ i4 load4(const int32_t* p) { return *(const i4*)p; }
  static __device__ __forceinline__ i4 load4(const uint16_t* p) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 v = *(const u2*)p;
    return i4{(int)(v[0] & 0xffffu), (int)(v[0] >> 16), (int)(v[1] & 0xffffu), (int)(v[1] >> 16)};
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
// the first column sweep of a factor with the residual update on the way: res = (res - uk0 vk0) + uk1 vk1, sums over the new values with uk
bool mfx_ccd_cols_can_fuse(mfx_ctx* ctx);
int mfx_ccd_pairs(mfx_ctx* ctx, int64_t n, const float* a0, const float* a1, float2* out);   // ccd.hip
int mfx_ccd_cols_pass_fused(mfx_ctx* ctx, const float* uk0, const float* vk0, const float* uk1, const float* vk1, const float* uk, float* vk,
                            float reg, float freq_thresh, int k);
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
