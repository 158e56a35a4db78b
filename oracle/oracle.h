/*
 * oracle.h -- CPU restatement of the matfac (mohit-shrma/matfac) training hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (matfac_amd/, include/mfx.h) never links, loads or calls it.
 *
 * PARITY PINNING.  The reference ships no tests, golden vectors or fixtures
 *  and cannot be compiled here (GKlib, Eigen, SVDLIBC,
 * gflags are absent).  The restatement is therefore pinned by
 *   (i)  the hand-derived known-answer tests of SURVEY.md section 8a (rows a4, a9, a10),
 *   (ii) the libstdc++ facts recorded in SURVEY.md section 8c (minstd_rand0 init
 *        stream, mt19937 outputs, std::shuffle permutations),
 * and, at the two third-party boundaries (Eigen ldlt()/dot(), GKlib gk_csr_Read),
 * it is "parity unpinned": the published algorithm is restated, no reference
 * output exists to compare against.
 *
 * All factor matrices here are ROW-MAJOR float [n][K] (the reference's Eigen
 * matrices are column-major; that is storage, not arithmetic).  A column-major
 * variant exists only for the cpu_baseline timing (orc_time_hogwild).
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference root).
 */
#ifndef MATFAC_ORACLE_H_
#define MATFAC_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dot-product order ------------------------------------------------- */
/* ORC_DOT_SEQ : k = 0..K-1, separate fp32 multiply and add (what Eigen's
 *               non-vectorisable strided row.dot(row) does; model.cpp:547-549).
 * ORC_DOT_TREE: the order the HIP kernels use (per-lane fma chain + xor
 *               butterfly, levels 1..L/2, over L lanes, C chunks); see orc_tree_shape().      */
enum { ORC_DOT_SEQ = 0, ORC_DOT_TREE = 1 };

/* ---- SGD arithmetic ---------------------------------------------------- */
/* ORC_ARITH_REF64 : modelMF.cpp:91-103 -- fp32 dot, double diff, double bracket,
 *                   one rounding to fp32 on "-=" (a4, a7).
 * ORC_ARITH_REF64F: modelMF.cpp:288-299 -- as above but diff is float (a6).
 * ORC_ARITH_F32   : modelMF.cpp:1755-1762 -- Eigen row expressions; the double
 *                   scalars are narrowed to float, the axpy is fp32 (a5).      */
enum { ORC_ARITH_REF64 = 0, ORC_ARITH_REF64F = 1, ORC_ARITH_F32 = 2 };

/* lanes-per-update L and 4-float chunks-per-lane C the device uses for rank K */
void orc_tree_shape(int K, int* L, int* C);
float orc_dot(const float* p, const float* q, int K, int dot_mode);

/* ---- RNG / shuffles (libstdc++; model.cpp:2331-2362, modelMF.cpp:63,78) -- */
void orc_init_factors(int seed, int nU, int nI, int K, float* U, float* V);
void* orc_mt_create(uint32_t seed);
void orc_mt_free(void* h);
uint32_t orc_mt_next(void* h);
void orc_mt_shuffle_u64(void* h, uint64_t* arr, int64_t n);          /* std::shuffle */
void orc_mt_shuffle_i32(void* h, int32_t* arr, int64_t n);
/* util.cpp:1047-1064 with the thread blocks shuffled one after another */
void orc_mt_par_block_shuffle_u64(void* h, uint64_t* arr, int64_t n, int nthreads);
/* util.cpp:1077-1107 */
void orc_mt_block_seq(void* h, int dim, int32_t* rows, int32_t* cols);

/* ---- CSR helpers (datastruct.cpp:3-120, util.cpp:511-544) -------------- */
/* stable counting sort: users ascending inside each column (gk_csr_CreateIndex) */
void orc_create_col_index(int32_t nrows, int32_t ncols, const int64_t* rowptr,
                          const int32_t* rowind, const float* rowval,
                          int64_t* colptr, int32_t* colind, float* colval);
/* invalid = zero train ratings; users >= train nrows / items >= train ncols too */
void orc_invalid(int32_t nrows, int32_t ncols, const int64_t* rowptr,
                 const int32_t* rowind, int32_t nUsers, int32_t nItems,
                 uint8_t* invU, uint8_t* invI);
/* text CSR (GK_CSR_FMT_CSR, values, 0-indexed). Two-call protocol: pass NULL
 * arrays to get sizes. Returns 0 on success. */
int orc_read_csr_text(const char* path, int32_t* nrows, int32_t* ncols, int64_t* nnz,
                      int64_t* rowptr, int32_t* rowind, float* rowval);
int orc_write_csr_text(const char* path, int32_t nrows, const int64_t* rowptr,
                       const int32_t* rowind, const float* rowval);
/* io.cpp:139-154 / io.cpp:83-121 */
int orc_write_mat(const char* path, const float* M, int nrows, int ncols);
int orc_read_mat(const char* path, float* M, int nrows, int ncols);

/* ---- SGD (modelMF.cpp:83-105, 279-302, 637-659, 1747-1763) ------------- */
/* sequential pass over ratings (u[],i[],r[]) visited in order[] (NULL = 0..n-1) */
void orc_sgd_pass(int K, float* U, float* V, const int32_t* u, const int32_t* i,
                  const float* r, const uint64_t* order, int64_t n, float lr,
                  float uReg, float iReg, int arith, int dot_mode);
/* OpenMP Hogwild over the same list (racy by design) */
void orc_sgd_hogwild(int K, float* U, float* V, const int32_t* u, const int32_t* i,
                     const float* r, const uint64_t* order, int64_t n, float lr,
                     float uReg, float iReg, int arith, int dot_mode, int nthreads);
/* stratified SGD (modelMF.cpp:229-304 + util.cpp:1077-1107).  create: shuffle the
 * valid users and items with mt and deal them into T parts (the first part
 * gets one extra element, as in the reference); epoch: T rounds, each a random
 * matching user-part -> item-part.  Tiles of one round are disjoint, so the
 * sequential sweep equals the reference's parallel one.  The parts are
 * std::unordered_set<int> built by the same insertion sequence, so the
 * within-part user order is the reference's (same libstdc++). */
void* orc_strat_create(void* mt, int32_t nrows, int32_t ncols, const uint8_t* invU,
                       const uint8_t* invI, int T);
void orc_strat_free(void* h);
void orc_strat_parts(void* h, int32_t nrows, int32_t ncols, int32_t* userPart,
                     int32_t* itemPart);
void orc_strat_epoch(void* h, void* mt, int K, float* U, float* V,
                     const int64_t* rowptr, const int32_t* rowind, const float* rowval,
                     float lr, float uReg, float iReg, int dot_mode);

/* ---- objective / RMSE (model.cpp:1770-1815, 214-251) ------------------- */
double orc_objective(int K, const float* U, const float* V, int32_t nUsers,
                     int32_t nItems, int32_t nrows, const int64_t* rowptr,
                     const int32_t* rowind, const float* rowval, const uint8_t* invU,
                     const uint8_t* invI, float uReg, float iReg, int dot_mode,
                     double* sse, double* unorm2, double* inorm2);
double orc_rmse(int K, const float* U, const float* V, int32_t nUsers, int32_t nItems,
                int32_t nrows, const int64_t* rowptr, const int32_t* rowind,
                const float* rowval, const uint8_t* invU, const uint8_t* invI,
                int dot_mode, double* sse, int64_t* cnt);

/* ---- ALS (modelMF.cpp:805-880) ----------------------------------------- */
/* Eigen LDLT<MatrixXf,Lower> restated (Eigen/src/Cholesky/LDLT.h, unblocked
 * in-place factorisation with diagonal pivoting + solve); A is K x K row-major,
 * destroyed.  Parity unpinned (Eigen is not in the reference tree). */
void orc_ldlt_solve(int K, float* A, const float* b, float* x);
/* side 0: users from the row view; side 1: items from the column view.
 * ptr/ind/val is the CSR (side 0) or CSC (side 1) of the train matrix.       */
void orc_als_half(int side, int K, float* X, const float* Y, int32_t nX,
                  const int64_t* ptr, const int32_t* ind, const float* val,
                  const uint8_t* invX, float reg, int nthreads);

/* ---- CCD++ (modelMF.cpp:1013-1121; FreqAdap 1258-1360) ----------------- */
/* one rank-one step for factor k on residual views res_row/res_col.
 * freq_thresh < 0: plain CCD++; >= 0: items with train frequency < thresh get
 * v_k = 0 when k > 0 (modelMF.cpp:1336-1342).                                */
void orc_ccdpp_rank1(int K, int k, float* U, float* V, int32_t nUsers, int32_t nItems,
                     int32_t ncols, const int64_t* rowptr, const int32_t* rowind,
                     float* res_row, const int64_t* colptr, const int32_t* colind,
                     float* res_col, const uint8_t* invU, const uint8_t* invI,
                     float uReg, float iReg, int add_back, int inner,
                     float freq_thresh, int nthreads);

/* ---- trainSGDParSVD pieces (modelMF.cpp:489-507, model.cpp:1818-1865) ---- */
void orc_sgd_pass_dimreg(int K, float* U, float* V, const int32_t* u, const int32_t* i, const float* r,
                         const uint64_t* order, int64_t n, float learnRate, const float* regk, int dot_mode);
double orc_objective_sing(int K, const float* U, const float* V, int32_t nUsers, int32_t nItems, int32_t nrows,
                          const int64_t* rowptr, const int32_t* rowind, const float* rowval, const uint8_t* invU,
                          const uint8_t* invI, const float* sing, int dot_mode, double* sse_out, double* ureg_out,
                          double* ireg_out);

/* ---- ModelInvPopMF / IFWMF (modelInvPopMF.cpp:3-55, 84-113, 152-178) ---- */
void orc_ifw_pop(int32_t nrows, int32_t ncols, const int64_t* rowptr, const int32_t* rowind, const uint8_t* invU,
                 const uint8_t* invI, double* userFreq, double* itemFreq, double* invPopU, double* invPopI);
void orc_sgd_pass_ifw(int K, float* U, float* V, const int32_t* u, const int32_t* i, const float* r, const uint64_t* order,
                      int64_t n, float learnRate, float uReg, float iReg, const double* userFreq, const double* itemFreq,
                      const double* invPopU, const double* invPopI, float rhoRMS, int dot_mode);
double orc_objective_ifw(int K, const float* U, const float* V, int32_t nUsers, int32_t nItems, int32_t nrows,
                         const int64_t* rowptr, const int32_t* rowind, const float* rowval, const uint8_t* invU,
                         const uint8_t* invI, float uReg, float iReg, const double* userFreq, const double* itemFreq,
                         const double* invPopU, const double* invPopI, float rhoRMS, int dot_mode, double* wsse_out);

/* ---- ModelDropoutSigmoid / TMF (modelDropoutSigmoid.cpp:5-24, 152-188) ---- */
void orc_tmf_ranks(int32_t n, const double* freq, double meanFreq, double stdFreq, float rhoRMS, float alpha, int facDim,
                   int32_t* rank);
void orc_sgd_pass_tmf(int K, float* U, float* V, const int32_t* u, const int32_t* i, const float* r, const uint64_t* order,
                      int64_t n, float learnRate, float uReg, float iReg, const double* userFreq, const double* itemFreq,
                      const int32_t* ru, const int32_t* ri, int dot_mode);
double orc_rmse_tmf(int K, const float* U, const float* V, int32_t nUsers, int32_t nItems, int32_t nrows,
                    const int64_t* rowptr, const int32_t* rowind, const float* rowval, const uint8_t* invU,
                    const uint8_t* invI, const double* userFreq, const double* itemFreq, const int32_t* ru,
                    const int32_t* ri, int dot_mode, double* sse_out, int64_t* cnt);

/* ---- ModelPoissonDropout / TMFDropout (modelPoissonDropout.cpp:5-47, 186-221) ---- */
void orc_cdf_ranks(int facDim, int32_t* cdfRanks);
int32_t orc_poisson_rank(int32_t lambda, uint32_t seed, uint32_t epoch, uint32_t u, uint32_t item, int32_t K);
void orc_sgd_pass_tmfd(int K, float* U, float* V, const int32_t* u, const int32_t* i, const float* r, const uint64_t* order,
                       int64_t n, float learnRate, float uReg, float iReg, const double* userFreq, const double* itemFreq,
                       const int32_t* lu, const int32_t* li, uint32_t seed, uint32_t epoch, int dot_mode);

/* ---- CCD (modelMF.cpp:1528-1605), sequential ---------------------------- */
void orc_ccd_iter(int K, float* U, float* V, int32_t nUsers, int32_t nItems,
                  int32_t ncols, const int64_t* rowptr, const int32_t* rowind,
                  float* res_row, const int64_t* colptr, const int32_t* colind,
                  float* res_col, const uint8_t* invU, const uint8_t* invI, float uReg,
                  float iReg, void* mt, uint16_t* uorder, uint16_t* iorder, int orders_given);

/* ---- full training loops with isTerminateModel (model.cpp:1471-1540) ---- */
enum {
  ORC_M_SGD = 0,      /* ModelMF::train            modelMF.cpp:4-151     */
  ORC_M_HOGSGD = 1,   /* ModelMF::hogTrain         modelMF.cpp:1656-1808 */
  ORC_M_SGDPAR = 2,   /* ModelMF::trainSGDPar      modelMF.cpp:154-350   */
  ORC_M_SGDU = 3,     /* ModelMF::trainUShuffle    modelMF.cpp:560-706   */
  ORC_M_ALS = 4,      /* ModelMF::trainALS         modelMF.cpp:709-928   */
  ORC_M_CCDPP = 5,    /* ModelMF::trainCCDPP       modelMF.cpp:931-1169  */
  ORC_M_CCDPP_FA = 6, /* ModelMF::trainCCDPPFreqAdap modelMF.cpp:1172-1423 */
  ORC_M_CCD = 7       /* ModelMF::trainCCD         modelMF.cpp:1426-1653 */
};

typedef struct {
  int32_t method, K, maxIter, seed, nthreads, dot_mode;
  float uReg, iReg, learnRate;
  int32_t nUsers, nItems;              /* Data::nUsers / nItems (datastruct.cpp:23,91) */
  int32_t tr_nrows, tr_ncols;          /* train matrix */
  const int64_t* tr_rowptr; const int32_t* tr_rowind; const float* tr_rowval;
  const int64_t* tr_colptr; const int32_t* tr_colind; const float* tr_colval;
  int32_t va_nrows; const int64_t* va_rowptr; const int32_t* va_rowind; const float* va_rowval;
} orc_train_cfg;

/* U,V: in = initial factors, out = LAST iterate; Ubest,Vbest: out = best-val
 * snapshot (bestModel).  objTraj/valTraj (maxIter entries) receive the per-
 * iteration objective and validation RMSE.  Returns the number of iterations
 * executed; *bestIter and *finalLearnRate are filled.                         */
int orc_train(const orc_train_cfg* cfg, float* U, float* V, float* Ubest, float* Vbest,
              double* objTraj, double* valTraj, int32_t* bestIter, float* finalLearnRate,
              uint8_t* invU, uint8_t* invI);

/* ---- ModelMFBias (modelMFBias.cpp:40-99, 163-197) ------------------------------ */
void orc_init_bias(int seed, int nU, int nI, int K, float* uBias, float* iBias);   /* model.cpp:2331-2362 */
void orc_bias_pass(float* uBias, float* iBias, const int32_t* u, const int32_t* i, const float* r, const uint64_t* order,
                   int64_t n, float learnRate, float uReg, float iReg);
double orc_bias_eval(const float* uBias, const float* iBias, int32_t nUsers, int32_t nItems, int32_t nrows, const int64_t* rowptr,
                     const int32_t* rowind, const float* rowval, const uint8_t* invU, const uint8_t* invI, float uReg, float iReg,
                     double* sse_out, int64_t* cnt_out, double* ub_out, double* ib_out);

/* ---- data preparation in front of the path (io.cpp:410-459, 726-787) ---------- */
void orc_split_colors(int64_t nnz, float testPc, float valPc, int seed, int32_t* color);
int64_t orc_rand_pairs(int32_t nUsers, int32_t nItems, int seed, int32_t nnz, int32_t* pairs);

/* ---- cpu_baseline timing (bench.py only) -------------------------------- */
/* OpenMP Hogwild epoch (modelMF.cpp:1746-1767 bracket), colmajor=1 reproduces the
 * reference's Eigen column-major factor storage. Returns seconds for `epochs`
 * epochs over the given list. */
double orc_time_hogwild(int K, int32_t nU, int32_t nI, float* U, float* V,
                        const int32_t* u, const int32_t* i, const float* r, int64_t n,
                        float lr, float uReg, float iReg, int nthreads, int colmajor,
                        int epochs);
/* trainSGDPar's epoch (modelMF.cpp:271-309 bracket) with the T blocks of a round in an OpenMP parallel for;
 * h from orc_strat_create.  Returns seconds for `epochs` epochs.                                          */
double orc_time_strat(void* h, void* mt, int K, float* U, float* V, const int64_t* rowptr,
                      const int32_t* rowind, const float* rowval, float lr, float uReg, float iReg,
                      int epochs);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
