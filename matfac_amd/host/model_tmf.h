// model_tmf.h -- ModelDropoutSigmoid (--algo=TMF, modelDropoutSigmoid.h): every rating trains and is estimated
// on the first updMinRank dimensions only, updMinRank = ceil(sigmoid(rhoRMS*(z - alpha)) * facDim) with z the z-score
// (over all user and item train frequencies) of the smaller of its two frequencies.  Same constructor arguments as
// the reference class; it rides on ModelMF's loop (the reference derives it from Model and repeats the stratified loop).
#ifndef MFHOST_MODEL_TMF_H_
#define MFHOST_MODEL_TMF_H_

#include <vector>

#include "mf_model.h"

class ModelDropoutSigmoid : public ModelMF {
 public:
  std::vector<double> userRankMap, itemRankMap, userFreq, itemFreq;
  double minFreq = 0, maxFreq = 0, meanFreq = 0, stdFreq = 1;

  ModelDropoutSigmoid(const Params& params, int seed, std::vector<double>& userRankMap, std::vector<double>& itemRankMap,
                      std::vector<double>& userFreq, std::vector<double>& itemFreq);

  void train(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;   // modelDropoutSigmoid.cpp:26-240
  double estRating(int user, int item) override;                                                          // :5-24
  // ceil(sigmoid * facDim), clamped to [1, facDim] (:158-172)
  int updMinRank(double freq) const;

 protected:
  void beforeLoop(Kind kind, const Data& data, IntSet& invalidUsers, IntSet& invalidItems) override;
};


// ModelPoissonDropout (--algo=TMFDropout, modelPoissonDropout.h): TMF whose update rank is a Poisson(lambda) draw per
// visit, lambda = the TMF rank of the rarer side; estimates use the dimensions 0..cdfRanks[lambda-1].
class ModelPoissonDropout : public ModelDropoutSigmoid {
 public:
  std::vector<double> factorial;
  std::vector<int> cdfRanks;

  ModelPoissonDropout(const Params& params, int seed, std::vector<double>& userRankMap, std::vector<double>& itemRankMap,
                      std::vector<double>& userFreq, std::vector<double>& itemFreq);
  void initCDFRanks();                                                                                    // modelPoissonDropout.cpp:25-47
  void train(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;   // :50-290
  double estRating(int user, int item) override;                                                          // :5-23

 protected:
  void beforeLoop(Kind kind, const Data& data, IntSet& invalidUsers, IntSet& invalidItems) override;
};

#endif
