// examples/dropin_main.cpp -- INTEGRATION.md section A as a program: the `--algo=mf` branch of the reference's main()
// (main.cpp:1325-1348, 1377-1382) written against THIS repo's headers with the reference's identifiers.  The only
// edits a matfac maintainer makes are the include line and the Params construction (gflags are not used here).
//   g++ -std=c++17 -Iinclude -Imatfac_amd/host examples/dropin_main.cpp -Lmatfac_amd -lmfhost -lmfx -o dropin
#include <iostream>
#include <memory>
#include <string>
#include <unordered_set>

#include "mf_model.h"      // instead of modelMF.h

int main(int argc, char* argv[]) {
  if (argc < 5) {
    std::cerr << "usage: dropin train.csr test.csr val.csr prefix [method=hogsgd] [facdim=16] [maxiter=20]" << std::endl;
    return -1;
  }
  std::string trainMat = argv[1], testMat = argv[2], valMat = argv[3], prefix = argv[4], none;
  const std::string method = argc > 5 ? argv[5] : "hogsgd";
  const int facDim = argc > 6 ? atoi(argv[6]) : 16, maxIter = argc > 7 ? atoi(argv[7]) : 20;
  // Params(facDim, maxIter, svdFacDim, seed, uReg, iReg, learnRate, rhoRMS, alpha, files...)   datastruct.h:32-51
  Params params(facDim, maxIter, facDim, 1, 0.01f, 0.01f, 0.005f, 0.0f, 0.0f, trainMat, testMat, valMat, none, none, none, none,
                none, prefix);
  Data data(params);
  params.nUsers = data.nUsers;
  params.nItems = data.nItems;
  params.display();

  std::unordered_set<int> invalidUsers, invalidItems;
  std::cout << "\nStarting model train...";
  std::unique_ptr<Model> mfModel = std::make_unique<ModelMF>(params, params.seed);
  std::unique_ptr<Model> bestModel = std::make_unique<ModelMF>(params, params.seed);
  if (method == "ccd++") mfModel->trainCCDPPFreqAdap(data, *bestModel, invalidUsers, invalidItems);
  else if (method == "ccd") mfModel->trainCCD(data, *bestModel, invalidUsers, invalidItems);
  else if (method == "als") mfModel->trainALS(data, *bestModel, invalidUsers, invalidItems);
  else if (method == "hogsgd") mfModel->hogTrain(data, *bestModel, invalidUsers, invalidItems);
  else if (method == "sgdu") mfModel->trainUShuffle(data, *bestModel, invalidUsers, invalidItems);
  else if (method == "sgdpar") mfModel->trainSGDPar(data, *bestModel, invalidUsers, invalidItems);
  else if (method == "sgdparsvd") mfModel->trainSGDParSVD(data, *bestModel, invalidUsers, invalidItems);
  else mfModel->train(data, *bestModel, invalidUsers, invalidItems);

  std::cout << "\nTrain RMSE: " << bestModel->RMSE(data.trainMat, invalidUsers, invalidItems);
  std::cout << "\nTest RMSE: " << bestModel->RMSE(data.testMat, invalidUsers, invalidItems);
  std::cout << "\nValidation RMSE: " << bestModel->RMSE(data.valMat, invalidUsers, invalidItems) << std::endl;
  return 0;
}
