#include "params_data.h"
#include "mf_model.h"

#include <algorithm>
#include <cstdlib>
#include <iostream>

Params::Params(int facDim, int maxIter, int svdFacDim, int seed, float uReg, float iReg, float learnRate,
               float rhoRMS, float alpha, std::string& trainMatFile, std::string& testMatFile,
               std::string& valMatFile, std::string& graphMatFile, std::string& origUFacFile,
               std::string& origIFacFile, std::string& initUFacFile, std::string& initIFacFile,
               std::string& prefix)
    : nUsers(-1), nItems(-1), facDim(facDim), maxIter(maxIter), svdFacDim(svdFacDim), seed(seed), uReg(uReg),
      iReg(iReg), learnRate(learnRate), rhoRMS(rhoRMS), alpha(alpha), trainMatFile(trainMatFile.c_str()),
      testMatFile(testMatFile.c_str()), valMatFile(valMatFile.c_str()),
      graphMatFile(graphMatFile.empty() ? nullptr : graphMatFile.c_str()),
      origUFacFile(origUFacFile.empty() ? nullptr : origUFacFile.c_str()),
      origIFacFile(origIFacFile.empty() ? nullptr : origIFacFile.c_str()),
      initUFacFile(initUFacFile.empty() ? nullptr : initUFacFile.c_str()),
      initIFacFile(initIFacFile.empty() ? nullptr : initIFacFile.c_str()), prefix(prefix.c_str()) {}

void Params::display() const {
  auto s = [](const char* p) { return p ? p : " "; };
  std::cout << "*** PARAMETERS ***" << std::endl;
  std::cout << "nUsers: " << nUsers << " nItems: " << nItems << std::endl;
  std::cout << "facDim: " << facDim << " svdFacDim: " << svdFacDim << std::endl;
  std::cout << "maxIter: " << maxIter << std::endl;
  std::cout << "uReg: " << uReg << " iReg: " << iReg << std::endl;
  std::cout << "rhoRMS: " << rhoRMS << " alpha: " << alpha << std::endl;
  std::cout << "learnRate: " << learnRate << std::endl;
  std::cout << "trainMat: " << s(trainMatFile) << std::endl;
  std::cout << "testMat: " << s(testMatFile) << std::endl;
  std::cout << "valMat: " << s(valMatFile) << std::endl;
  std::cout << "graphMat: " << s(graphMatFile) << std::endl;
  std::cout << "origUFac: " << s(origUFacFile) << std::endl;
  std::cout << "origIFac: " << s(origIFacFile) << std::endl;
  std::cout << "initUFac: " << s(initUFacFile) << std::endl;
  std::cout << "initIFac: " << s(initIFacFile) << std::endl;
}

static int max_item(const csr_t* m, int cur) {
  if (!m) return cur;
  for (int64_t e = 0; e < m->nnz(); e++) cur = std::max(cur, (int)m->rowind[e]);
  return cur;
}

// nUsers = trainMat->nrows (datastruct.cpp:23); nItems = maxItemInd + 1 where maxItemInd starts at
// trainMat->ncols - 1 and is raised by every item index seen in train, test and val (:22-91).
void Data::finish() {
  nUsers = trainMat->nrows;
  trainNNZ = (int)trainMat->nnz();
  int maxItemInd = trainMat->ncols - 1;
  maxItemInd = max_item(trainMat, maxItemInd);
  maxItemInd = max_item(testMat, maxItemInd);
  maxItemInd = max_item(valMat, maxItemInd);
  nItems = maxItemInd + 1;
  // The reference builds the column views here (gk_csr_CreateIndex, datastruct.cpp:60-62).  The MF path reads
  // only the train matrix' column view, and only on the device: mfx_set_csr builds it there (setup.hip) when
  // none is passed.  csr_create_col_index(m) is there for a caller that wants one on the host.
}

Data::Data(csr_t* p_trainMat, csr_t* p_testMat) : trainMat(p_trainMat), testMat(p_testMat) {
  nUsers = trainMat->nrows;
  nItems = trainMat->ncols;
}

Data::Data(csr_t* train, csr_t* test, csr_t* val, const char* pfx)
    : prefix(pfx), trainMat(train), testMat(test), valMat(val) {
  finish();
}

Data::Data(const Params& params) {
  facDim = params.facDim;
  prefix = params.prefix;
  std::string err;
  auto load = [&](const char* what, const char* file) -> csr_t* {
    if (!file) return nullptr;
    std::cout << "Reading " << what << " matrix 0-indexed... " << file << std::endl;
    csr_t* m = csr_read_text(file, &err);
    if (!m) {  // GKlib aborts on unreadable files; so do we, loudly
      std::cerr << "\n" << err << std::endl;
      throw MfxError(MFH_EXIT_FAIL, err);        // (the CLI ends with status 255 = exit(-1); the C API returns the code)
    }
    return m;
  };
  trainMat = load("partial train", params.trainMatFile);
  testMat = load("test", params.testMatFile);
  valMat = load("val", params.valMatFile);
  if (!trainMat) {
    std::cerr << "\nNo train matrix" << std::endl;
    throw MfxError(MFH_EXIT_FAIL, "No train matrix");
  }
  finish();
  std::cout << "\ntrain nnz = " << trainNNZ << std::endl;
  std::cout << "train nrows: " << trainMat->nrows << " ncols: " << trainMat->ncols << std::endl;
  std::cout << "nItems: " << nItems << std::endl;
}

Data::~Data() {
  csr_free(&trainMat);
  csr_free(&testMat);
  csr_free(&valMat);
  csr_free(&graphMat);
}
