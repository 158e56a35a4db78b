#include "model_invpop.h"

#include <iostream>

void ModelInvPopMF::train(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) {
  run(K_IFW, "train", data, bestModel, invalidUsers, invalidItems);
}

// modelInvPopMF.cpp:84-113: scores of the valid users / items, normalised to sum 1, then on the device
void ModelInvPopMF::beforeLoop(Kind kind, const Data& data, IntSet& invalidUsers, IntSet& invalidItems) {
  if (kind != K_IFW) return;
  const csr_t* trainMat = data.trainMat;
  std::vector<int> trainUsers, trainItems;
  for (int u = 0; u < trainMat->nrows; u++)
    if (invalidUsers.count(u) == 0) trainUsers.push_back(u);
  nTrainUsers = (int)trainUsers.size();
  for (int item = 0; item < trainMat->ncols; item++)
    if (invalidItems.count(item) == 0) trainItems.push_back(item);
  nTrainItems = (int)trainItems.size();
  if ((int)userFreq.size() < trainMat->nrows || (int)itemFreq.size() < trainMat->ncols) {
    throw MfxError(-100, "ModelInvPopMF: userFreq/itemFreq do not cover the train matrix");
  }
  double sumPopScore = 0;
  for (auto& u : trainUsers) {
    invPopU[u] = userFreq[u] / ((double)nTrainItems);
    sumPopScore += invPopU[u];
  }
  for (auto& u : trainUsers) invPopU[u] = invPopU[u] / sumPopScore;
  sumPopScore = 0;
  for (auto& item : trainItems) {
    invPopI[item] = itemFreq[item] / ((double)nTrainUsers);
    sumPopScore += invPopI[item];
  }
  for (auto& item : trainItems) invPopI[item] = invPopI[item] / sumPopScore;
  // the kernels read (freq, score) pairs; `float wt = invPopI[item]` (:161) narrows the score exactly like this
  std::vector<float> uf((size_t)nUsers, 0.0f), up((size_t)nUsers, 0.0f), itf((size_t)nItems, 0.0f), ip((size_t)nItems, 0.0f);
  for (int u = 0; u < trainMat->nrows && u < nUsers; u++) uf[u] = (float)userFreq[u];
  for (int i = 0; i < trainMat->ncols && i < nItems; i++) itf[i] = (float)itemFreq[i];
  for (auto& kv : invPopU) if (kv.first >= 0 && kv.first < nUsers) up[kv.first] = (float)kv.second;
  for (auto& kv : invPopI) if (kv.first >= 0 && kv.first < nItems) ip[kv.first] = (float)kv.second;
  dev->check(mfx_sgd_set_ifw(dev->ctx, uf.data(), up.data(), itf.data(), ip.data(), rhoRMS), "mfx_sgd_set_ifw");
  weightsOn = true;
}

void ModelInvPopMF::afterLoop(Kind kind) {
  if (kind != K_IFW) return;
  dev->check(mfx_sgd_set_ifw(dev->ctx, nullptr, nullptr, nullptr, nullptr, 0.0f), "mfx_sgd_set_ifw");
  weightsOn = false;
}

// modelInvPopMF.cpp:3-55: sum wt*diff*diff + uReg*sum ||p||^2 + iReg*sum ||q||^2
double ModelInvPopMF::objective(const Data& data, IntSet& invalidUsers, IntSet& invalidItems) {
  if (!weightsOn || !dev) return Model::objective(data, invalidUsers, invalidItems);
  mfx_eval_out o;
  dev->check(mfx_eval_ifw(dev->ctx, devSnap, &o), "mfx_eval_ifw");
  return o.sse + o.unorm2 * uReg + o.inorm2 * iReg;
}
