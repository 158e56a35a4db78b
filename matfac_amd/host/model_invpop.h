// model_invpop.h -- ModelInvPopMF (--algo=IFWMF, modelInvPopMF.h): MF with the squared error of every rating
// weighted by 1/(1 + rhoRMS * popularity score) of its rarer side.  Same public surface as the reference class;
// it rides on ModelMF's loop (the reference derives it from Model and repeats the loop).
#ifndef MFHOST_MODEL_INVPOP_H_
#define MFHOST_MODEL_INVPOP_H_

#include <map>
#include <vector>

#include "mf_model.h"

class ModelInvPopMF : public ModelMF {
 public:
  std::map<int, double> invPopU, invPopI;
  std::vector<double> userFreq, itemFreq;
  int nTrainUsers = 0, nTrainItems = 0;

  ModelInvPopMF(const Params& params, int seed, std::vector<double>& userFreq, std::vector<double>& itemFreq)
      : ModelMF(params, seed), userFreq(userFreq), itemFreq(itemFreq) {}
  ModelInvPopMF(const Params& params, const char* uFacName, const char* iFacName, int seed, std::vector<double>& userFreq,
                std::vector<double>& itemFreq)
      : ModelMF(params, uFacName, iFacName, seed), userFreq(userFreq), itemFreq(itemFreq) {}

  void train(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;    // modelInvPopMF.cpp:58-226
  double objective(const Data& data, IntSet& invalidUsers, IntSet& invalidItems) override;                // :3-55

 protected:
  void beforeLoop(Kind kind, const Data& data, IntSet& invalidUsers, IntSet& invalidItems) override;
  void afterLoop(Kind kind) override;
  bool baseObjective() const override { return false; }
  bool weightsOn = false;
};

#endif
