// init.cpp -- factor initialisation of Model::Model(const Params&) (model.cpp:2331-2341).
#include <random>

#include "mfhost.h"

extern "C" void mfh_init_factors(int32_t seed, int32_t nUsers, int32_t nItems, int32_t K, float* U, float* V) {
  std::default_random_engine gen(seed);
  const float lo = -0.01, hi = 0.01;
  std::uniform_real_distribution<double> draw(lo, hi);
  for (int64_t t = 0, n = (int64_t)nUsers * K; t < n; t++) {
    const float x = (float)draw(gen);
    if (U) U[t] = x;
  }
  for (int64_t t = 0, n = (int64_t)nItems * K; t < n; t++) {
    const float x = (float)draw(gen);
    if (V) V[t] = x;
  }
}
