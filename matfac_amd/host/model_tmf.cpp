#include "model_tmf.h"

#include <cmath>
#include <iostream>

// the constructor bodies of modelDropoutSigmoid.h:39-98: min/max and meanStdDev (util.cpp:278-294) of userFreq ++ itemFreq
ModelDropoutSigmoid::ModelDropoutSigmoid(const Params& params, int seed, std::vector<double>& userRankMap,
                                         std::vector<double>& itemRankMap, std::vector<double>& userFreq,
                                         std::vector<double>& itemFreq)
    : ModelMF(params, seed), userRankMap(userRankMap), itemRankMap(itemRankMap), userFreq(userFreq), itemFreq(itemFreq) {
  std::vector<double> v(userFreq.begin(), userFreq.end());
  v.insert(v.end(), itemFreq.begin(), itemFreq.end());
  if (v.empty()) return;
  minFreq = maxFreq = v[0];
  double sum = 0;
  for (double x : v) { sum += x; minFreq = std::min(minFreq, x); maxFreq = std::max(maxFreq, x); }
  const double mean = sum / v.size();
  double sq_sum = 0;
  for (double x : v) sq_sum += (x - mean) * (x - mean);
  meanFreq = mean;
  stdFreq = sqrt(sq_sum / v.size());
}

int ModelDropoutSigmoid::updMinRank(double freq) const {
  const double scaleFreq = (freq - meanFreq) / stdFreq;
  const double sigmPc = 1.0 / (1.0 + exp(-rhoRMS * (scaleFreq - alpha)));
  int r = std::ceil(sigmPc * ((double)facDim));
  if (r < MF_EPS) r = 1;
  if (r > facDim) r = facDim;
  return r;
}

void ModelDropoutSigmoid::train(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) {
  run(K_TMF, "train", data, bestModel, invalidUsers, invalidItems);
}

// one rank per user and per item (the rank of a rating is that of its rarer side), then on the device: from here
// on the session's evaluation kernels use the truncated estimate as well, which is what estRating() does to
// Model::RMSE / objective in the reference
void ModelDropoutSigmoid::beforeLoop(Kind kind, const Data& data, IntSet&, IntSet&) {
  if (kind != K_TMF) return;
  if ((int)userFreq.size() < data.trainMat->nrows || (int)itemFreq.size() < data.trainMat->ncols) {
    std::cerr << "\nModelDropoutSigmoid: userFreq/itemFreq do not cover the train matrix" << std::endl;
    exit(-2);
  }
  std::vector<float> uf((size_t)nUsers, 0.0f), itf((size_t)nItems, 0.0f);
  std::vector<int32_t> ru((size_t)nUsers, 1), ri((size_t)nItems, 1);
  for (int u = 0; u < nUsers; u++) {
    const double f = u < (int)userFreq.size() ? userFreq[u] : 0.0;
    uf[u] = (float)f;
    ru[u] = updMinRank(f);
  }
  for (int i = 0; i < nItems; i++) {
    const double f = i < (int)itemFreq.size() ? itemFreq[i] : 0.0;
    itf[i] = (float)f;
    ri[i] = updMinRank(f);
  }
  dev->check(mfx_set_tmf(dev->ctx, uf.data(), ru.data(), itf.data(), ri.data()), "mfx_set_tmf");
  std::cout << "rhoRMS: " << rhoRMS << " alpha: " << alpha << std::endl;
  std::cout << "minFreq: " << minFreq << " maxFreq: " << maxFreq << std::endl;
}

double ModelDropoutSigmoid::estRating(int user, int item) {
  syncHost();
  const bool isUMinFreq = userFreq[user] < itemFreq[item];
  const int r = updMinRank(isUMinFreq ? userFreq[user] : itemFreq[item]);
  double rat = 0;
  for (int k = 0; k < r; k++) rat += uFac(user, k) * iFac(item, k);
  return rat;
}
