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
    throw MfxError(-100, "ModelDropoutSigmoid: userFreq/itemFreq do not cover the train matrix");
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

// ---------------------------------------------------------------------------
// ModelPoissonDropout
// ---------------------------------------------------------------------------
ModelPoissonDropout::ModelPoissonDropout(const Params& params, int seed, std::vector<double>& userRankMap,
                                         std::vector<double>& itemRankMap, std::vector<double>& userFreq,
                                         std::vector<double>& itemFreq)
    : ModelDropoutSigmoid(params, seed, userRankMap, itemRankMap, userFreq, itemFreq) {
  factorial.push_back(1);
  for (int i = 1; i <= params.facDim + 1; i++) factorial.push_back(factorial.back() * ((double)i));
  initCDFRanks();
}

void ModelPoissonDropout::initCDFRanks() {
  cdfRanks = std::vector<int>(facDim, 0);
  double cdf = 0, wt = 0;
  for (int lambda = 1; lambda <= facDim; lambda++) {
    cdf = std::exp(-lambda) * (std::pow(lambda, 0) / factorial[0]);
    int k = 0;
    for (k = 0; k < facDim; k++) {
      wt = std::exp(-lambda) * (std::pow(lambda, k + 1) / factorial[k + 1]);
      cdf += wt;
      if (cdf >= 0.99) break;
    }
    cdfRanks[lambda - 1] = k;
    if (k == facDim) cdfRanks[lambda - 1] = k - 1;
  }
}

void ModelPoissonDropout::train(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) {
  run(K_TMFD, "train", data, bestModel, invalidUsers, invalidItems);
}

// lambda per user / item for the draws, cdfRanks[lambda-1]+1 dimensions for every estimate of the session
void ModelPoissonDropout::beforeLoop(Kind kind, const Data& data, IntSet&, IntSet&) {
  if (kind != K_TMFD) return;
  if ((int)userFreq.size() < data.trainMat->nrows || (int)itemFreq.size() < data.trainMat->ncols) {
    throw MfxError(-100, "ModelPoissonDropout: userFreq/itemFreq do not cover the train matrix");
  }
  std::vector<float> uf((size_t)nUsers, 0.0f), itf((size_t)nItems, 0.0f);
  std::vector<int32_t> lu((size_t)nUsers, 1), li((size_t)nItems, 1), eu((size_t)nUsers, 1), ei((size_t)nItems, 1);
  for (int u = 0; u < nUsers; u++) {
    const double f = u < (int)userFreq.size() ? userFreq[u] : 0.0;
    uf[u] = (float)f;
    lu[u] = updMinRank(f);                                  // lambda = ceil(sigmPc * facDim), > 0 (:196-197)
    eu[u] = std::min(cdfRanks[lu[u] - 1] + 1, facDim);      // k <= cdfRanks[lambda-1] && k < facDim (:17)
  }
  for (int i = 0; i < nItems; i++) {
    const double f = i < (int)itemFreq.size() ? itemFreq[i] : 0.0;
    itf[i] = (float)f;
    li[i] = updMinRank(f);
    ei[i] = std::min(cdfRanks[li[i] - 1] + 1, facDim);
  }
  dev->check(mfx_set_tmf(dev->ctx, uf.data(), eu.data(), itf.data(), ei.data()), "mfx_set_tmf");
  dev->check(mfx_set_tmf_dropout(dev->ctx, lu.data(), li.data(), (uint32_t)trainSeed), "mfx_set_tmf_dropout");
  std::cout << "minFreq: " << minFreq << " maxFreq: " << maxFreq << std::endl;
  std::cout << "rhoRMS: " << rhoRMS << " alpha: " << alpha << std::endl;
}

double ModelPoissonDropout::estRating(int user, int item) {
  syncHost();
  const bool isUMinFreq = userFreq[user] < itemFreq[item];
  const int lambda = updMinRank(isUMinFreq ? userFreq[user] : itemFreq[item]);
  double rat = 0;
  for (int k = 0; k <= cdfRanks[lambda - 1] && k < facDim; k++) rat += uFac(user, k) * iFac(item, k);
  return rat;
}
