/*
 * mfx.h -- C ABI of the MI355X (gfx950) matrix-completion hot path.
 *
 * This is the drop-in boundary for the training loops of mohit-shrma/matfac's
 * ModelMF (reference: modelMF.cpp / model.cpp).  The reference has no FFI layer:
 * its boundary is the C++ class API `void ModelMF::trainX(const Data&, Model&
 * bestModel, unordered_set<int>& invalidUsers, unordered_set<int>& invalidItems)`
 * (modelMF.h:30-56) over GKlib `gk_csr_t` matrices and Eigen factor matrices.
 * The host classes in matfac_amd/host/ keep that class API; each of their inner
 * loops is one call below.  Every entry point cites the reference loop it replaces.
 *
 * Conventions
 *  - plain C types only; no exceptions cross this boundary; nothing calls exit().
 *  - every function returns MFX_OK (0) or a negative mfx_status; the message is
 *    available from mfx_last_error().  (The reference prints to cerr and exit()s:
 *    model.cpp:1481-1484, main.cpp:62-64.)
 *  - host pointers are borrowed for the duration of the call only.
 *  - one mfx_ctx <-> one device <-> one HIP stream; calls on a ctx are serialised
 *    by the caller.  Different ctxs may be driven from different threads/processes.
 *  - there is NO CPU fallback: without a HIP device mfx_create() fails with
 *    MFX_E_NODEVICE.
 *
 * Device layout (DESIGN.md "Data layout in HBM"): factors are row-major
 * float[n][ld], ld = 4*L*C >= K, zero padded, where L lanes of a wavefront own
 * one rating and each lane owns 4 consecutive floats per chunk:
 *      K <= 16: L=4,C=1   K <= 32: L=8,C=1   else L=16, C=ceil(K/64).
 * The fp32 dot product is evaluated as: per-lane fma chain over the lane's
 * elements (chunk-major) starting from 0, then an xor butterfly (levels 1, 2,
 * ..., L/2) over the L lanes.  (Eigen's order for row.dot(row) is unspecified; the CPU
 * oracle can evaluate either order.)
 */
#ifndef MFX_H_
#define MFX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mfx_ctx mfx_ctx;

typedef enum {
  MFX_OK = 0,
  MFX_E_ARG = -1,      /* bad argument / shape mismatch            */
  MFX_E_HIP = -2,      /* HIP runtime error                        */
  MFX_E_COMM = -3,     /* RCCL error / librccl not loadable        */
  MFX_E_OOM = -4,      /* device or host allocation failed         */
  MFX_E_STATE = -5,    /* call sequence error (e.g. no train matrix) */
  MFX_E_NODEVICE = -6  /* no HIP device: there is no CPU fallback  */
} mfx_status;

enum { MFX_MAT_TRAIN = 0, MFX_MAT_VAL = 1, MFX_MAT_TEST = 2 };
enum { MFX_ROWMAJOR = 0, MFX_COLMAJOR = 1 };   /* host layout; Eigen::MatrixXf is COLMAJOR */
enum { MFX_SNAP_CURRENT = 0, MFX_SNAP_BEST = 1 };
enum { MFX_SIDE_USERS = 0, MFX_SIDE_ITEMS = 1 };

/* ---- lifetime ----------------------------------------------------------- */
int mfx_version(void);
int mfx_device_count(int* n);
int mfx_create(int device, mfx_ctx** out);
void mfx_destroy(mfx_ctx* ctx);
/* message of the last failing call on ctx (ctx may be NULL: last mfx_create error) */
const char* mfx_last_error(const mfx_ctx* ctx);
int mfx_synchronize(mfx_ctx* ctx);

/* ---- data: replaces Data / gk_csr_t (datastruct.h:72-136, datastruct.cpp:3-120) */
/* Upload one rating matrix.  rowptr has nrows+1 entries (GKlib ssize_t), indices
 * are 0-based int32, values float.  The column view (gk_csr_CreateIndex(COL),
 * datastruct.cpp:18) may be NULL: it is then built here by a stable counting sort
 * (users ascending inside a column).  Only TRAIN needs a column view.          */
int mfx_set_csr(mfx_ctx* ctx, int which, int32_t nrows, int32_t ncols,
                const int64_t* rowptr, const int32_t* rowind, const float* rowval,
                const int64_t* colptr, const int32_t* colind, const float* colval);

/* ---- model: replaces Model's factor storage (model.h:36-37, model.cpp:2315-2366) */
/* nUsers/nItems are Data::nUsers / Data::nItems (nItems = 1 + max item index over
 * train, test and val: datastruct.cpp:91).  Allocates U, V and the best snapshot. */
int mfx_set_model(mfx_ctx* ctx, int32_t nUsers, int32_t nItems, int32_t K);
int mfx_set_factors(mfx_ctx* ctx, const float* U, const float* V, int layout);
int mfx_get_factors(mfx_ctx* ctx, int snapshot, float* U, float* V, int layout);
/* getInvalidUsersItems (util.cpp:511-544) + modelMF.cpp:40-45: users/items with no
 * train rating (or beyond the train matrix) are invalid.  Computes the masks on
 * the device, keeps them for the evaluation kernels and returns them (1 = invalid);
 * the outputs may be NULL.                                                       */
int mfx_compute_invalid(mfx_ctx* ctx, uint8_t* invalidUsers, uint8_t* invalidItems);
/* bestModel = *this (model.cpp:1500-1504) / *this = bestModel (model.cpp:1492),
 * factor matrices only; learnRate bookkeeping stays with the host Model.        */
int mfx_snapshot_best(mfx_ctx* ctx);
int mfx_restore_best(mfx_ctx* ctx);

/* ---- SGD: replaces modelMF.cpp:83-105 (train), :1747-1763 (hogTrain),
 *      :637-659 (trainUShuffle), :273-304 (trainSGDPar)                          */
enum {
  MFX_SGD_HOGWILD = 0,  /* all ratings of the epoch list in flight, lock-free (a5)     */
  MFX_SGD_SERIAL = 1,   /* one rating at a time in list order: bit-exact a4 / a7 order  */
  MFX_SGD_USERS = 2,    /* one wave-group per user row, items in CSR order, users in
                           the (shuffled) user list: parallel trainUShuffle (a7)      */
  MFX_SGD_TILED = 3,    /* Hogwild scheduled in 8x8 (user-block x item-block) tiles: in
                           round r the workgroups running on XCD x visit tile
                           (x, (x+r) mod 8) only, so every factor row has ONE XCD (one
                           coherent L2) reading and writing it at a time; the order
                           inside a tile is the per-epoch device permutation.  This is
                           the stratification of trainSGDPar (modelMF.cpp:229-304) mapped
                           onto the chip's 8 L2 domains; MFX_ORDER_DEVICE only.        */
  MFX_SGD_LEVELS = 4    /* the SEQUENTIAL loop in list order, bit for bit what MFX_SGD_SERIAL
                           gives, run in parallel: a rating waits for the previous rating
                           of its user and of its item in the list and for nothing else
                           (visits that share no row commute exactly).  Dataflow schedule
                           (owned rows + version counters, sgd_flow.hip) or dependency
                           levels with a grid barrier (sgd_levels.hip).  The exact replay
                           of ModelMF::train / trainUShuffle / trainSGDPar orders
                           (modelMF.cpp:83-105, 637-659, 273-304); with rating weights,
                           truncated ranks or a per-dimension regulariser installed, of
                           the sibling models' loops (dataflow schedule only).             */
};
enum {
  MFX_ORDER_DEVICE = 0,  /* fresh device-side pseudo-random permutation per (seed, epoch) */
  MFX_ORDER_HOST = 1,    /* the permutation last given to mfx_sgd_set_order()             */
  MFX_ORDER_NATURAL = 2  /* CSR order                                                     */
};
enum {
  MFX_ARITH_REF64 = 0,   /* modelMF.cpp:91-103: double diff, double bracket, one rounding */
  MFX_ARITH_REF64F = 1,  /* modelMF.cpp:288-299: float diff, double bracket              */
  MFX_ARITH_F32 = 2      /* modelMF.cpp:1755-1762: scalars narrowed to float, fp32 axpy  */
};
typedef struct {
  int32_t mode, order, arith;
  float learnRate, uReg, iReg;
  uint32_t seed;       /* MFX_ORDER_DEVICE: permutation key (with epoch)        */
  int32_t epoch;
  int32_t blocks;      /* HOGWILD/TILED: 256-thread workgroups in flight (each keeps
                          4*64/L ratings in flight); 0 = auto: min(nUsers,nItems)/64
                          clamped to [8,2048] -- lock-free SGD loses updates when the
                          ratings in flight are not << min(nUsers,nItems)            */
  int32_t own;         /* MFX_SGD_TILED: which rows a workgroup owns in LDS for a slot (lossless
                          updates on that side): 0 = item rows (default; popular items are where
                          lock-free updates collide), 1 = user rows, 2 = alternate by epoch --
                          1 and 2 are experiments: measured worse / unstable       */
  int64_t first, count; /* sub-range of the epoch list; count <= 0: everything  */
  int32_t flags;       /* MFX_SGD_F_* (0 for production runs)                    */
  int32_t item_part;   /* 0: the whole train matrix; p + 1: only the ratings whose item belongs to item
                          part p of mfx_sgd_set_item_parts (MFX_SGD_TILED; the multi-GPU rotation)  */
} mfx_sgd_opts;
/* Test hooks of MFX_SGD_TILED (tests/test_sgd_gpu.py, tests/test_fullsize_gpu.py):
 *  ONE_GROUP     the slots are visited tile by tile and slot by slot by ONE lane group, one rating at a
 *                time, with the production kernel's per-rating code: the list mfx_debug_epoch_list returns IS
 *                the visiting order, so a sequential replay of that list (the reference's loop,
 *                modelMF.cpp:83-105 / :1747-1763) must reproduce the factors.
 *  COUNT_VISITS  every rating record the update loop consumes bumps a device counter; read them (and zero
 *                them) with mfx_debug_visit_counts: "every rating exactly once per epoch" as observed by
 *                the kernel itself.
 *  DRAIN_ONLY    the eight XCD-scheduled round launches are skipped and the placement-independent drain launch does
 *                the whole epoch: what a partition mode with one reported XCC_ID would run (and the only way to
 *                make the drain's barrier-separated diagonals do real work on a healthy device).                */
enum { MFX_SGD_F_ONE_GROUP = 1, MFX_SGD_F_COUNT_VISITS = 2, MFX_SGD_F_DRAIN_ONLY = 4 };
/* Permutation of the train ratings (indices into the CSR-order rating list with
 * invalid users/items removed -- on a train matrix that is every rating), as
 * std::vector<size_t> uiRatingInds in modelMF.cpp:67-68; for MFX_SGD_USERS the
 * list is the shuffled valid-user list (modelMF.cpp:620-635).                   */
int mfx_sgd_set_order(mfx_ctx* ctx, const uint64_t* perm, int64_t n);
/* the same list as 32-bit indices (half the bytes to shuffle on the host and to upload; the host classes use it for lists below
 * 2^32 entries)                                                                  */
int mfx_sgd_set_order32(mfx_ctx* ctx, const uint32_t* perm, int64_t n);
/* std::shuffle of that list ON THE DEVICE: the list of mfx_sgd_set_order32 (n entries) becomes what
 *   for (i = 1; i < n; i++) swap(a[i], a[pos[i]]);        0 <= pos[i] <= i, pos[0] ignored
 * leaves -- the plain loop libstdc++'s std::shuffle runs beyond 65 536 entries (modelMF.cpp:76-81: the reference
 * shuffles the SAME vector every epoch) -- and stays the MFX_ORDER_HOST list.  The host only draws the positions
 * (the generator's stream); the 20 M random swaps of an ML-20M epoch, 45 ms on a host core, take ~ 3 ms here.
 * pos is copied when the call returns.                                            */
int mfx_sgd_apply_swaps32(mfx_ctx* ctx, const uint32_t* pos, int64_t n);
/* test hook: the 32-bit list the device holds (out == NULL queries n) */
int mfx_debug_order32(mfx_ctx* ctx, uint32_t* out, int64_t cap, int64_t* n);
int mfx_sgd_epoch(mfx_ctx* ctx, const mfx_sgd_opts* opts);
/* test hook: the (u,i,r) list the last epoch visited, in visiting order */
int mfx_debug_epoch_list(mfx_ctx* ctx, int32_t* u, int32_t* i, float* r, int64_t cap,
                         int64_t* n);
/* the schedule of the last MFX_SGD_LEVELS epoch: info = {1 (dataflow), longest queue, lane groups, owned side} or
 * {0 (levels), levels, levels run with the grid barrier, tail threshold}; prep_ms (may be NULL) = host time spent
 * building it                                                                                                  */
int mfx_debug_levels_info(mfx_ctx* ctx, int64_t info[4], double* prep_ms);
/* test hook: the queues of the last dataflow epoch: records = int32[4] per rating {other-side row, owned row, rating bits,
 * expected version (= the rank of the rating in the other-side row's chain)}, queue g = records [qoff[g], qoff[g+1]); records == NULL queries the two counts                     */
int mfx_debug_flow_queues(mfx_ctx* ctx, int32_t* records, int64_t cap, int64_t* qoff, int64_t* n_records, int64_t* n_groups);
/* test hook: digest of the slot lists the last MFX_SGD_TILED epoch ran on.  counts = {slots, ratings, row
 * references, rows per slot}; sums = FNV-1a of {rating records, slot_beg, slot_ibeg, slot rows, tile_slot} */
int mfx_debug_slots_digest(mfx_ctx* ctx, int64_t counts[4], uint64_t sums[5]);
/* test hook: the user block (0 .. 7) and item block (0 .. 7) of every row in the 8 x 8 tiling the last MFX_SGD_TILED epoch ran on
 * (balanced over the ratings: rows in descending order of their rating count, each onto the lightest block so far)          */
int mfx_debug_tile_blocks(mfx_ctx* ctx, uint8_t* user_block, int64_t n_users, uint8_t* item_block, int64_t n_items);
/* test hook: visits per rating record (slot-list order, as mfx_debug_epoch_list indexes its positions through the
 * per-slot permutation) accumulated by epochs run with MFX_SGD_F_COUNT_VISITS; zeroes the counters.  n receives
 * the number of records; counts may be NULL to query n.                                                         */
int mfx_debug_visit_counts(mfx_ctx* ctx, uint32_t* counts, int64_t cap, int64_t* n);
/* test hook: raises the sticky abort flag of the tiled schedule's drain on the device, as a drain that gave up at its grid
 * barrier does.  The next tiled epoch copies it back; the first call that synchronises after that (mfx_synchronize,
 * mfx_eval*, mfx_get_factors, or the next tiled epoch) returns MFX_E_HIP once, and the flag is cleared.                  */
int mfx_debug_raise_drain_abort(mfx_ctx* ctx);
/* test hook: the column view of the train matrix as the device holds it (given or built by mfx_set_csr) */
int mfx_debug_col_view(mfx_ctx* ctx, int64_t* colptr, int32_t* colind, float* colval);

/* ---- evaluation: replaces Model::objective (model.cpp:1770-1815) and
 *      Model::RMSE (model.cpp:214-251)                                          */
typedef struct {
  double sse;      /* sum over valid (u,i) of (r - p.q)^2             */
  int64_t n;       /* ratings counted                                 */
  double unorm2;   /* sum over valid users of ||p_u||^2 (TRAIN only)  */
  double inorm2;   /* sum over valid items of ||q_i||^2 (TRAIN only)  */
} mfx_eval_out;
/* objective = sse + uReg*unorm2 + iReg*inorm2 ; RMSE = sqrt(sse/n).
 * with_norms != 0 also fills unorm2/inorm2.                                     */
int mfx_eval(mfx_ctx* ctx, int which, int snapshot, int with_norms, mfx_eval_out* out);
/* Two evaluations with one copy back and one synchronisation: what Model::isTerminateModel needs every iteration
 * (objective on train with the norms, RMSE on validation; model.cpp:1476-1480).                                  */
int mfx_eval2(mfx_ctx* ctx, int whichA, int with_normsA, int whichB, int with_normsB, int snapshot,
              mfx_eval_out* outA, mfx_eval_out* outB);
/* Model::RMSE(mat, filtItems, ...) (model.cpp:348-394) and Model::RMSEU(mat, filtUsers, ...) (:446-486): the
 * evaluation restricted to the users / items whose keep flag is non-zero (either array may be NULL = keep
 * all; keepUsers has nUsers entries, keepItems nItems).  out->sse and out->n as mfx_eval, no norms.          */
int mfx_eval_filtered(mfx_ctx* ctx, int which, int snapshot, const uint8_t* keepUsers,
                      const uint8_t* keepItems, mfx_eval_out* out);

/* ---- ALS: replaces modelMF.cpp:805-841 (users) / :844-880 (items) ----------- */
int mfx_als_half_sweep(mfx_ctx* ctx, int side, float reg);

/* ---- CCD++: replaces modelMF.cpp:1013-1121 (trainCCDPP) and :1258-1360
 *      (trainCCDPPFreqAdap)                                                     */
/* begin: res = gk_csr_Dup(trainMat) on both views, uFac.fill(0) (:1013,:1020)    */
int mfx_ccdpp_begin(mfx_ctx* ctx);
/* one factor k: optional add-back (iter > 0), `inner` sweeps of {row pass, column
 * pass}, subtract, write the columns back.  freq_thresh < 0: plain CCD++;
 * otherwise items with train frequency < freq_thresh get v_k = 0 for k > 0.     */
int mfx_ccdpp_rank1(mfx_ctx* ctx, int32_t k, int32_t inner, float uReg, float iReg,
                    int32_t add_back, float freq_thresh);
int mfx_ccdpp_end(mfx_ctx* ctx);
/* test hook: the two residual views */
int mfx_debug_residuals(mfx_ctx* ctx, float* res_row, float* res_col);

/* ---- cyclic coordinate descent: replaces the loops of ModelMF::trainCCD
 *      (modelMF.cpp:1528-1565 users, :1567-1605 items) -------------------------- */
/* res = gk_csr_Dup(trainMat) on both views (:1509) and uFac = 0 (:1516-1522).
 * Needs the stable column view (the one gk_csr_CreateIndex / mfx_set_csr build).  */
int mfx_ccd_begin(mfx_ctx* ctx);
/* One sweep over the users (MFX_SIDE_USERS, reg = uReg) or the items.  Every row
 * visits the K factors in its own order: order[row * K + step] (host, one row per
 * user resp. item of the train matrix; the reference's std::shuffle(udims, mt),
 * :1539-1540) or, with order == NULL, a permutation derived on the device from
 * (seed, iter, side, row).                                                        */
int mfx_ccd_sweep(mfx_ctx* ctx, int32_t side, float reg, const uint16_t* order, uint32_t seed,
                  int32_t iter);
int mfx_ccd_end(mfx_ctx* ctx);
/* test hook: the residuals on both views (either pointer may be NULL) */
int mfx_debug_ccd_residuals(mfx_ctx* ctx, float* res_row, float* res_col);

/* ---- ModelMF::trainSGDParSVD (modelMF.cpp:353-557) ------------------------------ */
/* Replaces svdFrmSvdlibCSREig(trainMat, facDim, uFac, iFac, false) (svdFrmsvdlib.cpp:69-133, SVDLIBC las2):
 * rank-K truncated SVD of the train matrix; uFac <- left, iFac <- right singular vectors (rows of items
 * beyond the train matrix keep their values), singular[K] <- singular values, descending.  Randomized block
 * subspace iteration: power_iters passes over R and R^T on a (K + oversample)-dimensional block.            */
int mfx_svd_init(mfx_ctx* ctx, int32_t power_iters, int32_t oversample, uint32_t seed, float* singular);
/* Per-dimension regulariser of the SVD variant: with reg != NULL, MFX_SGD_HOGWILD / MFX_SGD_SERIAL epochs run
 * x_k -= lr * (-2 diff y_k + 2 reg[k] x_k) with a float diff (modelMF.cpp:494-505) on both sides and ignore
 * opts->uReg/iReg/arith; NULL switches back.  reg[k] = (sing_a + 1) / (sing_b + sigma_k).                    */
int mfx_sgd_set_dim_reg(mfx_ctx* ctx, const float* reg);
/* Model::objectiveSing (model.cpp:1818-1865): as mfx_eval, with unorm2 / inorm2 = sum over valid rows of
 * sum_k x_k^2 * w[k].                                                                                        */
int mfx_eval_weighted(mfx_ctx* ctx, int which, int snapshot, const float* w, mfx_eval_out* out);

/* ---- ModelInvPopMF (--algo=IFWMF): inverse-frequency-weighted MF ------------------ */
/* Per-rating weight on the error term (modelInvPopMF.cpp:161-166): wt = invPopI[item], or invPopU[u] when
 * itemFreq[item] > userFreq[u]; wt = 1/(1 + rhoRMS*wt).  Arrays of nUsers resp. nItems floats (the reference's
 * doubles narrowed as `float wt = invPopI[item]` does).  With weights set, MFX_SGD_HOGWILD / MFX_SGD_SERIAL epochs
 * run p -= lr*(-2*wt*diff*q + 2*uReg*p) etc. in the double bracket (:168-176) and ignore opts->arith.  All four
 * pointers NULL: weights off.                                                                                   */
int mfx_sgd_set_ifw(mfx_ctx* ctx, const float* userFreq, const float* invPopU, const float* itemFreq,
                    const float* invPopI, float rhoRMS);
/* ModelInvPopMF::objective (modelInvPopMF.cpp:3-55): out->sse = sum wt*diff*diff over the valid train ratings,
 * n and the two norms as mfx_eval.                                                                              */
int mfx_eval_ifw(mfx_ctx* ctx, int snapshot, mfx_eval_out* out);

/* ---- ModelDropoutSigmoid (--algo=TMF): truncated-rank MF ---------------------------- */
/* Every rating uses only the first `rank` dimensions, rank = userRank[u] when userFreq[u] < itemFreq[item], else
 * itemRank[item] (modelDropoutSigmoid.cpp:158-188; the caller evaluates ceil(sigmoid(..)*facDim) per user and per
 * item, ranks in [1, K]).  With the table set, MFX_SGD_HOGWILD / MFX_SGD_SERIAL epochs update those dimensions only
 * (float diff, double bracket) and mfx_eval / mfx_eval_filtered use the truncated estimate (the class's estRating
 * override, :5-24).  All four pointers NULL: off.                                                                */
int mfx_set_tmf(mfx_ctx* ctx, const float* userFreq, const int32_t* userRank, const float* itemFreq,
                const int32_t* itemRank);

/* ModelPoissonDropout (--algo=TMFDropout, modelPoissonDropout.cpp:186-221): on top of mfx_set_tmf (whose ranks then
 * serve the evaluation: cdfRanks, :5-23), every SGD visit uses Poisson(lambda) dimensions, clipped to [1, K], lambda =
 * userLambda[u] when userFreq[u] < itemFreq[item], else itemLambda[item].  The draw is a pure function of
 * (seed, opts->epoch, u, item) -- the reference's per-thread mt19937 streams depend on the thread count.  NULLs: off.  */
int mfx_set_tmf_dropout(mfx_ctx* ctx, const int32_t* userLambda, const int32_t* itemLambda, uint32_t seed);

/* ---- ModelMFBias: the bias-only sibling (modelMFBias.cpp) ------------------------------- */
/* estRating = uBias[u] + iBias[item] (:94-99, a float sum); one visit (:178-197), diff taken ONCE before both steps:
 *   uBias[u]    -= learnRate*(-2.0*diff + 2.0*uReg*uBias[u]);   iBias[item] -= learnRate*(-2.0*diff + 2.0*iReg*iBias[item])
 * in double, narrowed to float on the store.  The vectors live next to the factor matrices of mfx_set_model (nUsers
 * resp. nItems floats); mfx_snapshot_best / mfx_restore_best carry them once they are set.                               */
int mfx_bias_set(mfx_ctx* ctx, const float* uBias, const float* iBias);
int mfx_bias_get(mfx_ctx* ctx, int snapshot, float* uBias, float* iBias);
/* one epoch of ModelMFBias::train's loop over the epoch list (opts->order as mfx_sgd_epoch; the reference shuffles the
 * rating tuples with std::shuffle every epoch, :166).  opts->mode: MFX_SGD_SERIAL (one lane in list order) or
 * MFX_SGD_LEVELS (the same result from the dataflow schedule of sgd_flow.hip); opts->arith is ignored.                    */
int mfx_bias_epoch(mfx_ctx* ctx, const mfx_sgd_opts* opts);
/* ModelMFBias::objective (:40-91) and Model::RMSE through the class's estRating: out->sse, out->n as mfx_eval,
 * out->unorm2 = sum over valid users of uBias[u]^2, out->inorm2 = sum over valid items of iBias[item]^2
 * (objective = sse + uReg*unorm2 + iReg*inorm2: the factor norms are computed and dropped by the reference).            */
int mfx_bias_eval(mfx_ctx* ctx, int which, int snapshot, mfx_eval_out* out);

/* ---- multi-GPU: user-row-block sharding, item-factor exchange over RCCL ------ */
/* The reference is single-process; this is new.  Each rank owns a
 * user block (its CSR rows + U shard) and a replica of V.  After local work,
 * V <- V_sync + sum_over_ranks (V - V_sync)   (MFX_REDUCE_DELTA_SUM)   or
 * V <- mean_over_ranks V                       (MFX_REDUCE_AVERAGE).              */
enum { MFX_REDUCE_DELTA_SUM = 0, MFX_REDUCE_AVERAGE = 1 };
/* Item-part rotation (the reference's own stratification, modelMF.cpp:273-304, at GPU granularity): the items are cut into
 * nparts parts (item i belongs to part i % nparts), an epoch on N = nparts ranks is N sub-epochs -- in sub-epoch s rank g runs
 * mfx_sgd_epoch with item_part = ((g + s) % N) + 1 on the ONLY current copy of that part's rows and then hands them on:
 * mfx_rotate_item_part(send, recv) sends the rows of part `send` to rank g - 1 and receives the rows of part `recv` from rank
 * g + 1 (a ring shift of 1/N of V; RCCL send/recv).  No update is lost, summed twice or down-weighted.  After the last
 * sub-epoch mfx_allgather_item_parts(part) -- every rank r contributes the part (part - rank + r) mod N it holds -- makes V
 * complete and identical on all ranks (needed before an evaluation).  nparts = 0 or 1 switches the parts off.            */
int mfx_sgd_set_item_parts(mfx_ctx* ctx, int nparts);
int mfx_rotate_item_part(mfx_ctx* ctx, int send_part, int recv_part);
int mfx_allgather_item_parts(mfx_ctx* ctx, int my_part);
#define MFX_UNIQUE_ID_BYTES 128
int mfx_comm_unique_id(void* id128);
int mfx_comm_init(mfx_ctx* ctx, int nranks, int rank, const void* id128);
/* Bring-your-own all-reduce (MPI, gloo, ...): fn must sum `count` elements of `host_buf` (dtype 0 = float,
 * 1 = double) over all ranks in place and return 0.  The library stages device buffers through pinned host
 * memory around the call.  Everything that works over RCCL works over this, slower.                          */
typedef int (*mfx_reduce_fn)(void* user, void* host_buf, int64_t count, int dtype);
int mfx_comm_init_external(mfx_ctx* ctx, int nranks, int rank, mfx_reduce_fn fn, void* user);
int mfx_comm_destroy(mfx_ctx* ctx);
/* declare the current V identical on all ranks (call after mfx_set_factors, before
 * the first local epoch): V_sync <- V                                             */
int mfx_comm_mark_synced(mfx_ctx* ctx);
int mfx_allreduce_item_factors(mfx_ctx* ctx, int op);
/* sum a few doubles over ranks (objective / RMSE partials) */
int mfx_allreduce_f64(mfx_ctx* ctx, double* vals, int n);

/* ---- measurement ------------------------------------------------------------ */
/* HIP-event timing of the hot kernels on the ctx stream (bench.py roofline).    */
enum {
  MFX_K_SGD = 0, MFX_K_PERMUTE = 1, MFX_K_EVAL = 2, MFX_K_ALS_GRAM = 3,
  MFX_K_ALS_SOLVE = 4, MFX_K_CCD_ROW = 5, MFX_K_CCD_COL = 6, MFX_K_CCD_RESID = 7,
  MFX_K_SGD_SWEEP = 8, /* MFX_SGD_TILED: the placement-independent leftover sweep */
  MFX_K_CD = 9,        /* one mfx_ccd_sweep (transpose + row kernels + view copy) */
  MFX_K_COUNT = 10
};
/* on = 0: off; 1: an event pair around every launch; N > 1: around the launches of every N-th mfx_sgd_epoch only
 * (an event pair per launch costs ~6 % of a 1 ms epoch; sampling keeps the timed region honest)                  */
int mfx_prof_enable(mfx_ctx* ctx, int on);
int mfx_prof_reset(mfx_ctx* ctx);
int mfx_prof_get(mfx_ctx* ctx, int kernel, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* MFX_H_ */
