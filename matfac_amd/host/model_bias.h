// model_bias.h -- ModelMFBias, the bias-only model (reference: modelMFBias.h:16-36, modelMFBias.cpp).
#ifndef MFHOST_MODEL_BIAS_H_
#define MFHOST_MODEL_BIAS_H_
#include "mf_model.h"

class ModelMFBias : public Model {
 public:
  ModelMFBias(const Params& params) : Model(params) {}
  ModelMFBias(const Params& params, int seed) : Model(params, seed) {}
  void train(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;   // :104-228
  double estRating(int user, int item) override;                                                         // :94-99
  double objective(const Data& data) override;                                                           // :3-37
  double objective(const Data& data, IntSet& invalidUsers, IntSet& invalidItems) override;               // :40-91
  void save(std::string prefix);                       // model.cpp:31-60: factors + <prefix>_uBias_<sig>.vec, _iBias_, _gBias
  void syncHost() override;                            // + the bias vectors of this object's device snapshot
  void pushToDevice() override;

 protected:
  void evalDevice(const csr_t* mat, int withNorms, mfx_eval_out* out) override;   // every RMSE goes through estRating (:94-99)
  bool baseObjective() const override { return false; }
};
#endif
