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
#include "model_invpop.h"
#include "model_tmf.h"
#include "model_bias.h"

#include <algorithm>
#include <fstream>
#include <vector>

typedef std::vector<std::pair<int, std::vector<int>>> Partition;

// util.cpp:555-569
static std::pair<std::vector<double>, std::vector<double>> getRowColFreq(const csr_t* mat) {
  std::vector<double> rowFreq((size_t)mat->nrows, 0), colFreq((size_t)mat->ncols, 0);
  for (int u = 0; u < mat->nrows; u++)
    for (int64_t ii = mat->rowptr[u]; ii < mat->rowptr[u + 1]; ii++) {
      rowFreq[u] += 1;
      colFreq[mat->rowind[ii]] += 1;
    }
  return std::make_pair(rowFreq, colFreq);
}

// main.cpp:1109-1135: quarters of the frequency-sorted list; the last part takes the remainder
static void setAdapRank(std::vector<int>& rankMap, Partition& parts, std::vector<std::pair<int, double>>& freqPairs, int facDim) {
  const int n = (int)freqPairs.size();
  int currFac = facDim, i = 0, partInd = 0;
  while (i < n) {
    int endItem = i + 0.25 * ((float)n);
    if (endItem > n || partInd == 3) endItem = n;
    std::cout << "start: " << i << " end: " << endItem << " currFac: " << currFac << std::endl;
    std::vector<int> p;
    for (int item = i; item < endItem; item++) {
      rankMap[freqPairs[item].first] = currFac;
      p.push_back(freqPairs[item].first);
    }
    parts.push_back(std::make_pair(partInd, p));
    currFac = currFac / 2;
    if (0 == currFac) currFac = 1;
    i = endItem;
    partInd++;
  }
}

// main.cpp:1137-1167.  The reference sorts with std::sort and a comparator on the frequency alone, so the
// order of equally frequent users/items is unspecified there; here ties keep ascending index (stable).
static void getUserItemRankMap(const Data& data, const Params& params, Partition& partItems, Partition& partUsers,
                               std::vector<int>& itemRank, std::vector<int>& userRank) {
  auto rowColFreq = getRowColFreq(data.trainMat);
  auto order = [](const std::vector<double>& f) {
    std::vector<std::pair<int, double>> pr;
    for (int i = 0; i < (int)f.size(); i++) pr.push_back(std::make_pair(i, f[i]));
    std::stable_sort(pr.begin(), pr.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.second > b.second; });
    return pr;
  };
  userRank.assign(rowColFreq.first.size(), 0);
  itemRank.assign(rowColFreq.second.size(), 0);
  auto ip = order(rowColFreq.second);
  setAdapRank(itemRank, partItems, ip, params.facDim);
  auto up = order(rowColFreq.first);
  setAdapRank(userRank, partUsers, up, params.facDim);
}

// main.cpp:700-768: count and RMSE of the test / validation ratings per item and per user frequency quartile
static void quartileRMSEs(Model& bestModel, const Data& data, Partition& partItems, Partition& partUsers,
                          std::unordered_set<int>& invalidUsers, std::unordered_set<int>& invalidItems) {
  std::cout << std::endl;
  std::cout << "Train RMSE: " << bestModel.RMSE(data.trainMat, invalidUsers, invalidItems) << std::endl;
  std::cout << "Test RMSE: " << bestModel.RMSE(data.testMat, invalidUsers, invalidItems) << std::endl;
  std::cout << "Val RMSE: " << bestModel.RMSE(data.valMat, invalidUsers, invalidItems) << std::endl;
  auto report = [&](const char* title, csr_t* mat) {
    std::cout << title << std::endl;
    std::cout << "Items Part: ";
    for (auto& p : partItems) {
      std::unordered_set<int> filt(p.second.begin(), p.second.end());
      auto c = bestModel.RMSE(mat, filt, invalidUsers, invalidItems);
      std::cout << c.first << " " << c.second << " ";
    }
    std::cout << std::endl;
    std::cout << "Users Part: ";
    for (auto& p : partUsers) {
      std::unordered_set<int> filt(p.second.begin(), p.second.end());
      auto c = bestModel.RMSEU(mat, filt, invalidUsers, invalidItems);
      std::cout << c.first << " " << c.second << " ";
    }
    std::cout << std::endl;
  };
  report("Test RMSE: ", data.testMat);
  report("Validation RMSE: ", data.valMat);
}

// main.cpp:1091-1107
static void writePartition(Partition& parts, std::unordered_set<int>& invalid, const char* opFileName) {
  std::ofstream opFile(opFileName);
  if (!opFile.is_open()) return;
  for (auto& part : parts)
    for (int elem : part.second)
      if (invalid.count(elem) == 0) opFile << part.first << " " << elem << std::endl;
}

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

static int run_main(int argc, char** argv);
int main(int argc, char** argv) {
  try {
    return run_main(argc, argv);
  } catch (const MfxError& e) {          // the library classes throw; the driver ends like the reference's does
    if (e.code == MFH_EXIT_OK) return 0;       // model.cpp:1481-1484: "No validation data", exit(0) (already printed)
    if (e.code == MFH_EXIT_FAIL) return 255;   // exit(-1) (already printed)
    std::cerr << "\n" << e.what() << std::endl;
    return 254;                          // exit(-2)
  }
}
static int run_main(int argc, char** argv) {
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

  if (flags["algo"] != "mf" && flags["algo"] != "IFWMF" && flags["algo"] != "TMF" && flags["algo"] != "TMFDropout" && flags["algo"] != "mfbias") {
    std::cerr << "Invalid algo input: " << flags["algo"] << " (this build implements --algo=mf, IFWMF, TMF, TMFDropout and mfbias)" << std::endl;
    exit(0);
  }
  Partition partItems, partUsers;
  std::vector<int> itemRank, userRank;
  getUserItemRankMap(data, params, partItems, partUsers, itemRank, userRank);   // main.cpp:1246-1250
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
  if (flags["algo"] == "mfbias") {      // ModelMFBias: built by the reference (CMakeLists.txt) but left out of its --algo dispatch (main.cpp:912, commented)
    mfModel.reset(new ModelMFBias(params, params.seed));
    bestModel.reset(new ModelMFBias(params, params.seed));
    mfModel->train(data, *bestModel, invalidUsers, invalidItems);
  } else if (flags["algo"] == "IFWMF") {       // main.cpp:1361-1366
    auto rowColFreq = getRowColFreq(data.trainMat);
    mfModel.reset(new ModelInvPopMF(params, params.seed, rowColFreq.first, rowColFreq.second));
    bestModel.reset(new ModelInvPopMF(params, params.seed, rowColFreq.first, rowColFreq.second));
    mfModel->train(data, *bestModel, invalidUsers, invalidItems);
  } else if (flags["algo"] == "TMFDropout") {   // main.cpp:1355-1360
    auto rowColFreq = getRowColFreq(data.trainMat);
    std::vector<double> userRankPc, itemRankPc;
    mfModel.reset(new ModelPoissonDropout(params, params.seed, userRankPc, itemRankPc, rowColFreq.first, rowColFreq.second));
    bestModel.reset(new ModelPoissonDropout(params, params.seed, userRankPc, itemRankPc, rowColFreq.first, rowColFreq.second));
    mfModel->train(data, *bestModel, invalidUsers, invalidItems);
  } else if (flags["algo"] == "TMF") {   // main.cpp:1349-1354 (userRankPc / itemRankPc are carried, not used by train)
    auto rowColFreq = getRowColFreq(data.trainMat);
    std::vector<double> userRankPc, itemRankPc;
    mfModel.reset(new ModelDropoutSigmoid(params, params.seed, userRankPc, itemRankPc, rowColFreq.first, rowColFreq.second));
    bestModel.reset(new ModelDropoutSigmoid(params, params.seed, userRankPc, itemRankPc, rowColFreq.first, rowColFreq.second));
    mfModel->train(data, *bestModel, invalidUsers, invalidItems);
  } else if (m == "ccd++") mfModel->trainCCDPPFreqAdap(data, *bestModel, invalidUsers, invalidItems);
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
    std::cout << std::endl << "**** Model parameters ****" << std::endl;
    mfModel->display();
    std::cout << std::endl;
    std::cout << "invalid users: " << invalidUsers.size() << " invalid items: " << invalidItems.size() << std::endl;
    quartileRMSEs(*bestModel, data, partItems, partUsers, invalidUsers, invalidItems);   // main.cpp:1409-1413
    // the reference drops these two into the working directory; here they go next to the factor files
    writePartition(partItems, invalidItems, (std::string(params.prefix) + "_itemPartition.txt").c_str());
    writePartition(partUsers, invalidUsers, (std::string(params.prefix) + "_userPartition.txt").c_str());
  }
  return 0;
}
