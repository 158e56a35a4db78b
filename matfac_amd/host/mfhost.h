/* mfhost.h -- C entry points of the host-side library (libmfhost.so).
 * Host data preparation around the hot path; no device code here.            */
#ifndef MFHOST_H_
#define MFHOST_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* synthetic rating matrices: which = 0 full, 1 train, 2 val, 3 test */
typedef struct mfh_synth mfh_synth;
mfh_synth* mfh_synth_create(int32_t nUsers, int32_t nItems, int64_t nnz, uint32_t seed, double alpha_u,
                            double alpha_i, double noise, int32_t K0, double frac_train, double frac_val,
                            uint32_t shard /* user-block id: items depend on seed only */,
                            double r0_u, double r0_i /* Zipf-Mandelbrot rank offsets */);
void mfh_synth_free(mfh_synth* s);
int mfh_synth_shape(const mfh_synth* s, int which, int32_t* nrows, int32_t* ncols, int64_t* nnz);
int mfh_synth_copy(const mfh_synth* s, int which, int64_t* rowptr, int32_t* rowind, float* rowval);
int32_t mfh_synth_nitems(const mfh_synth* s);

/* Model::Model(const Params&) factor initialisation (model.cpp:2331-2341):
 * std::default_random_engine(seed), uniform_real_distribution<double>(-0.01f, 0.01f),
 * uFac row by row then iFac; row-major outputs [n][K].  Either output may be NULL
 * (its draws are still consumed so that V does not depend on whether U was asked for). */
void mfh_init_factors(int32_t seed, int32_t nUsers, int32_t nItems, int32_t K, float* U, float* V);

/* Run ModelMF::<method> (sgd | hogsgd | sgdpar | sgdu | als | ccdpp | ccd++) the way main() does
 * (main.cpp:1325-1348, 1377-1382) on in-memory matrices.  All three matrices have `nrows` rows.
 * Outputs (each may be NULL): last iterate, bestModel, stats = {train RMSE, test RMSE, val RMSE of
 * bestModel, final learnRate of the model, of bestModel, nItems, seconds in the iteration loop,
 * iterations run} (8 doubles), invalid user/item masks.
 * prefix NULL: no factor files are written. */
int mfh_train(const char* method, int32_t nrows, const int64_t* tr_ptr, const int32_t* tr_ind,
              const float* tr_val, int32_t tr_ncols, const int64_t* va_ptr, const int32_t* va_ind,
              const float* va_val, int32_t va_ncols, const int64_t* te_ptr, const int32_t* te_ind,
              const float* te_val, int32_t te_ncols, int32_t K, int32_t maxIter, int32_t seed, float learnRate,
              float uReg, float iReg, const char* prefix, float* Ulast, float* Vlast, float* Ubest, float* Vbest,
              double* stats, uint8_t* invU, uint8_t* invI);

/* csr_read_text / csr_create_col_index (csr.cpp) and Data::Data(const Params&) (params_data.cpp) for tests */
int mfh_csr_read_text(const char* path, int32_t* nrows, int32_t* ncols, int64_t* nnz, int64_t* rowptr,
                      int32_t* rowind, float* rowval, int64_t* colptr, int32_t* colind, float* colval, char* err,
                      int errcap);
int mfh_data_shape(const char* train, const char* test, const char* val, int32_t* nUsers, int32_t* nItems,
                   int32_t* trainNNZ);

/* factor files (mf_model.cpp writeMat/readMat/writeMatBin/readMatBin) for tests; a name ending in .binmat reads binary */
int mfh_mat_write(const char* path, const float* data, int32_t n, int32_t k, int32_t bin);
int mfh_mat_read(const char* path, float* data, int32_t n, int32_t k);

/* test hook: 1 when mfhShuffle (the block-ahead form of the epoch shuffle, mf_model.cpp) and std::shuffle turn the list 0 .. n-1 into
 * the same permutation with mt19937(seed) and leave the generator in the same state; secs (may be NULL) = {seconds of std::shuffle, of mfhShuffle,
 * the form mfhShuffle takes: 0 library call, 1 block-ahead, 2 block-ahead with the restated generator and distribution} */
int mfh_shuffle_check(int64_t n, uint32_t seed, double* secs);
/* the same with mfhShuffle working on a list of 32-bit entries (what ModelMF::train shuffles below 2^32 ratings) */
int mfh_shuffle_check32(int64_t n, uint32_t seed, double* secs);
/* test hook: the swap positions mfhShufflePositions draws for a list of n entries (what the device applies with mfx_sgd_apply_swaps32):
 * 1 when `for i: swap(a[i], a[pos[i]])` on 0 .. n-1 gives std::shuffle's list and the generator ends in the same state, 0 when not, 2 when
 * the positions form does not apply (short list / another standard library); pos (may be NULL) receives the n positions; secs[0] = seconds */
int mfh_shuffle_positions_check(int64_t n, uint32_t seed, uint32_t* pos, double* secs);

#ifdef __cplusplus
}
#endif
#endif
