// mf_main.cpp -- command line driver `mf` with the reference's flags (main.cpp:26-46) for the
// --algo=mf methods: sgd | sgdpar | sgdu | hogsgd | als | ccd++ (main.cpp:1325-1348), followed by
// the Train/Test/Validation RMSE report of main.cpp:1377-1382.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <string>

#include "mf_model.h"

static std::map<std::string, std::string> flags = {
    {"maxiter", "5000"}, {"facdim", "5"},   {"svdfacdim", "5"}, {"ureg", "0.01"}, {"ireg", "0.01"},
    {"learnrate", "0.005"}, {"rhorms", "0.0"}, {"alpha", "0.0"}, {"seed", "1"}, {"trainmat", ""},
    {"testmat", ""}, {"valmat", ""}, {"graphmat", ""}, {"origufac", ""}, {"origifac", ""},
    {"initufac", ""}, {"initifac", ""}, {"prefix", ""}, {"mf_method", "sgd"}, {"algo", "mf"}};

static void parse(int argc, char** argv) {
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if (a.rfind("--", 0) != 0 && a.rfind("-", 0) != 0) continue;
    a = a.substr(a.rfind("--", 0) == 0 ? 2 : 1);
    std::string name = a, val;
    const size_t eq = a.find('=');
    if (eq != std::string::npos) { name = a.substr(0, eq); val = a.substr(eq + 1); }
    else if (i + 1 < argc) val = argv[++i];
    if (!flags.count(name)) {
      std::cerr << "ERROR: unknown command line flag '" << name << "'" << std::endl;
      exit(1);
    }
    flags[name] = val;
  }
}

int main(int argc, char** argv) {
  parse(argc, argv);
  bool isexit = false;
  if (flags["trainmat"].empty() || flags["testmat"].empty() || flags["valmat"].empty()) {
    std::cerr << "Missing either: train, test or val matrix" << std::endl;
    isexit = true;
  }
  if (flags["prefix"].empty()) {
    std::cerr << "Missing prefix string to prepend: --prefix " << std::endl;
    isexit = true;
  }
  if (isexit) exit(-1);
  Params params(atoi(flags["facdim"].c_str()), atoi(flags["maxiter"].c_str()), atoi(flags["svdfacdim"].c_str()),
                atoi(flags["seed"].c_str()), (float)atof(flags["ureg"].c_str()), (float)atof(flags["ireg"].c_str()),
                (float)atof(flags["learnrate"].c_str()), (float)atof(flags["rhorms"].c_str()),
                (float)atof(flags["alpha"].c_str()), flags["trainmat"], flags["testmat"], flags["valmat"],
                flags["graphmat"], flags["origufac"], flags["origifac"], flags["initufac"], flags["initifac"],
                flags["prefix"]);
  Data data(params);
  params.nUsers = data.nUsers;
  params.nItems = data.nItems;
  params.display();

  if (flags["algo"] != "mf") {
    std::cerr << "Invalid algo input: " << flags["algo"] << " (this build implements --algo=mf)" << std::endl;
    exit(0);
  }
  std::unordered_set<int> invalidUsers, invalidItems;
  std::cout << "\nStarting model train...";
  std::unique_ptr<Model> mfModel, bestModel;
  if (!flags["initufac"].empty() && !flags["initifac"].empty()) {   // parsed but unused by the reference's main
    mfModel.reset(new ModelMF(params, flags["initufac"].c_str(), flags["initifac"].c_str(), params.seed));
    bestModel.reset(new ModelMF(params, flags["initufac"].c_str(), flags["initifac"].c_str(), params.seed));
  } else {
    mfModel.reset(new ModelMF(params, params.seed));
    bestModel.reset(new ModelMF(params, params.seed));
  }
  const std::string m = flags["mf_method"];
  if (m == "ccd++") mfModel->trainCCDPPFreqAdap(data, *bestModel, invalidUsers, invalidItems);
  else if (m == "ccdpp") mfModel->trainCCDPP(data, *bestModel, invalidUsers, invalidItems);   // reachable only programmatically in the reference
  else if (m == "ccd") mfModel->trainCCD(data, *bestModel, invalidUsers, invalidItems);
  else if (m == "als") mfModel->trainALS(data, *bestModel, invalidUsers, invalidItems);
  else if (m == "hogsgd") mfModel->hogTrain(data, *bestModel, invalidUsers, invalidItems);
  else if (m == "sgdu") mfModel->trainUShuffle(data, *bestModel, invalidUsers, invalidItems);
  else if (m == "sgdpar") mfModel->trainSGDPar(data, *bestModel, invalidUsers, invalidItems);
  else if (m == "sgdparsvd") mfModel->trainSGDParSVD(data, *bestModel, invalidUsers, invalidItems);
  else mfModel->train(data, *bestModel, invalidUsers, invalidItems);

  if (bestModel->dev) {
    std::cout << "\nTrain RMSE: " << bestModel->RMSE(data.trainMat, invalidUsers, invalidItems);
    std::cout << "\nTest RMSE: " << bestModel->RMSE(data.testMat, invalidUsers, invalidItems);
    std::cout << "\nValidation RMSE: " << bestModel->RMSE(data.valMat, invalidUsers, invalidItems) << std::endl;
  }
  return 0;
}
