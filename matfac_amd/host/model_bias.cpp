// model_bias.cpp -- ModelMFBias on the device (reference: modelMFBias.cpp).  The class keeps the reference's surface;
// the loop body (:178-197) is mfx_bias_epoch, the objective / RMSE sums are mfx_bias_eval.
#include "model_bias.h"

#include <algorithm>
#include <chrono>
#include <fstream>
#include <iostream>
#include <numeric>
#include <random>

static double vecNorm(const std::vector<float>& v) {
  double s = 0;
  for (float x : v) s += (double)x * x;
  return std::sqrt(s);
}

double ModelMFBias::estRating(int user, int item) {
  syncHost();
  return uBias[(size_t)user] + iBias[(size_t)item];      // a float sum widened to double (:95)
}

void ModelMFBias::syncHost() {
  if (!hostStale || !dev) return;
  Model::syncHost();
  uBias.resize((size_t)nUsers);
  iBias.resize((size_t)nItems);
  dev->check(mfx_bias_get(dev->ctx, devSnap, uBias.data(), iBias.data()), "mfx_bias_get");
}

void ModelMFBias::pushToDevice() {
  Model::pushToDevice();
  dev->check(mfx_bias_set(dev->ctx, uBias.data(), iBias.data()), "mfx_bias_set");
}

void ModelMFBias::evalDevice(const csr_t* mat, int withNorms, mfx_eval_out* out) {
  (void)withNorms;
  const int w = dev ? dev->which(mat) : -1;
  if (w < 0) {
    throw MfxError(-100, "ModelMFBias: matrix is not part of the device session");
  }
  dev->check(mfx_bias_eval(dev->ctx, w, devSnap, out), "mfx_bias_eval");
}

// :40-91: squared error over the valid ratings + uReg*sum uBias^2 + iReg*sum iBias^2 (the factor norms are formed and
// left out of the sum by the reference, :84-85)
double ModelMFBias::objective(const Data& data, IntSet& invalidUsers, IntSet& invalidItems) {
  (void)invalidUsers; (void)invalidItems;
  mfx_eval_out o;
  evalDevice(data.trainMat, 1, &o);
  return o.sse + o.unorm2 * uReg + o.inorm2 * iReg;
}
double ModelMFBias::objective(const Data& data) {
  IntSet a, b;
  return objective(data, a, b);
}

static void writeVec(const std::vector<float>& v, const char* name) {
  std::ofstream op(name);
  if (!op.is_open()) return;
  for (float x : v) op << x << std::endl;               // io.cpp writeVector: one value per line
}

void ModelMFBias::save(std::string prefix) {
  if (getenv("MFX_NO_SAVE")) return;
  syncHost();
  const std::string sign = modelSignature();
  writeMat(uFac, nUsers, facDim, (prefix + "_uFac_" + sign + ".mat").c_str());
  writeMat(iFac, nItems, facDim, (prefix + "_iFac_" + sign + ".mat").c_str());
  writeVec(uBias, (prefix + "_uBias_" + sign + ".vec").c_str());
  std::cout << "user bias norm: " << vecNorm(uBias) << std::endl;
  writeVec(iBias, (prefix + "_iBias_" + sign + ".vec").c_str());
  std::cout << "item bias norm: " << vecNorm(iBias) << std::endl;
  std::ofstream g((prefix + "_" + sign + "_gBias").c_str());
  if (g.is_open()) g << mu << std::endl;
}

void ModelMFBias::train(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) {
  std::cout << "\nModelMFBias::train trainSeed: " << trainSeed;
  const csr_t* trainMat = data.trainMat;
  {                                                      // mu = meanRating(trainMat) (util.cpp:99-111): kept, never used by estRating
    double avg = 0;
    for (int64_t e = 0; e < trainMat->nnz(); e++) avg += trainMat->rowval[e];
    mu = avg / (double)trainMat->nnz();
  }
  std::cout << "\nGlobal bias: " << mu;
  attach(data);
  bestModel.dev = dev;
  bestModel.devSnap = MFX_SNAP_BEST;
  {
    // bestModel starts as its own initialisation
    ModelMFBias* bb = dynamic_cast<ModelMFBias*>(&bestModel);
    if (bb && (int)bb->uBias.size() == nUsers && (int)bb->iBias.size() == nItems && bb->uFac.rows == nUsers) {
      dev->check(mfx_set_factors(dev->ctx, bb->uFac.data(), bb->iFac.data(), MFX_ROWMAJOR), "set best");
      dev->check(mfx_bias_set(dev->ctx, bb->uBias.data(), bb->iBias.data()), "set best bias");
      dev->check(mfx_snapshot_best(dev->ctx), "snapshot best");
      pushToDevice();
    } else {
      dev->check(mfx_snapshot_best(dev->ctx), "snapshot best");
    }
    bestModel.hostStale = true;
  }
  std::cout << "\nObj b4 svd: " << objective(data) << " Train RMSE: " << RMSE(data.trainMat) << " Train nnz: " << data.trainNNZ << std::endl;

  int iter, bestIter = -1;
  double bestObj, prevObj, bestValRMSE, prevValRMSE;
  deviceInvalid(data, invalidUsers, invalidItems);
  prevObj = objective(data, invalidUsers, invalidItems);
  bestObj = prevObj;
  bestValRMSE = prevValRMSE = RMSE(data.valMat, invalidUsers, invalidItems);
  std::cout << "\nObj aftr svd: " << prevObj << " Train RMSE: " << RMSE(data.trainMat);
  std::cout << "\nModelMFBias::train trainSeed: " << trainSeed << " invalidUsers: " << invalidUsers.size()
            << " invalidItems: " << invalidItems.size() << std::endl;
  syncHost();
  std::cout << "ubias norm: " << vecNorm(uBias) << " iBias norm: " << vecNorm(iBias) << std::endl;

  std::mt19937 mt(trainSeed);
  // getUIRatings (util.cpp:722-747): every train rating belongs to a valid user and item, so the tuple list is the CSR
  // list.  The reference shuffles the TUPLES in place every epoch (:166), i.e. each shuffle permutes the previous order:
  // the same thing on an index vector.
  const int64_t nRatings = trainMat->nnz();
  std::vector<size_t> inds((size_t)nRatings);
  std::iota(inds.begin(), inds.end(), 0);
  std::cout << "\nNo. of training ratings: " << nRatings;
  mfx_sgd_opts o = mfx_sgd_opts();
  o.uReg = uReg; o.iReg = iReg; o.seed = (uint32_t)trainSeed;
  const char* ex = getenv("MFX_EXACT");
  o.mode = (ex && atoi(ex) == 2) ? MFX_SGD_SERIAL : MFX_SGD_LEVELS;
  o.order = MFX_ORDER_HOST;
  const auto loopStart = std::chrono::steady_clock::now();
  for (iter = 0; iter < maxIter; iter++) {
    mfhShuffle(inds, mt);              // std::shuffle, bit for bit (mf_model.cpp)
    dev->check(mfx_sgd_set_order(dev->ctx, (const uint64_t*)inds.data(), nRatings), "set_order");
    o.learnRate = learnRate;
    o.epoch = iter;
    dev->check(mfx_bias_epoch(dev->ctx, &o), "mfx_bias_epoch");
    hostStale = true;
    if (iter % MF_OBJ_ITER == 0 || iter == maxIter - 1) {
      if (isTerminateModel(bestModel, data, iter, bestIter, bestObj, prevObj, bestValRMSE, prevValRMSE, invalidUsers, invalidItems)) break;
      if (iter % MF_DISP_ITER == 0)
        std::cout << "ModelMFBias::train trainSeed: " << trainSeed << " Iter: " << iter << " Objective: " << std::scientific << prevObj
                  << " Train RMSE: " << RMSE(data.trainMat, invalidUsers, invalidItems) << " Val RMSE: " << prevValRMSE << std::endl;
      if (iter % MF_SAVE_ITER == 0 || iter == maxIter - 1) {
        ModelMFBias* bb = dynamic_cast<ModelMFBias*>(&bestModel);
        if (bb) bb->save(std::string(data.prefix));
      }
    }
  }
  lastLoopSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - loopStart).count();
  lastIters = std::min(iter + 1, maxIter);
  syncHost();
  bestModel.syncHost();
}
