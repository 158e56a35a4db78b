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

#ifdef __cplusplus
}
#endif
#endif
