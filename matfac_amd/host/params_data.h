// params_data.h -- Params and Data with the reference's field names and semantics
// (datastruct.h:12-136, datastruct.cpp:3-120), over this library's csr_t instead of GKlib.
#ifndef MFHOST_PARAMS_DATA_H_
#define MFHOST_PARAMS_DATA_H_
#include <string>
#include <vector>

#include "csr.h"

// datastruct.h:12-69.  The const char* members BORROW the caller's strings, as in the reference.
class Params {
 public:
  int nUsers, nItems, facDim, maxIter, svdFacDim, seed;
  float uReg, iReg, learnRate, rhoRMS, alpha;
  const char *trainMatFile, *testMatFile, *valMatFile, *graphMatFile;
  const char *origUFacFile, *origIFacFile, *initUFacFile, *initIFacFile, *prefix;

  Params(int facDim, int maxIter, int svdFacDim, int seed, float uReg, float iReg, float learnRate,
         float rhoRMS, float alpha, std::string& trainMatFile, std::string& testMatFile,
         std::string& valMatFile, std::string& graphMatFile, std::string& origUFacFile,
         std::string& origIFacFile, std::string& initUFacFile, std::string& initIFacFile,
         std::string& prefix);
  void display() const;
};

// datastruct.h:72-136.  Owns the three rating matrices (row AND column view built, as the
// reference does with gk_csr_CreateIndex), derives nUsers = train rows and
// nItems = 1 + max item index over train, test and val.
class Data {
 public:
  const char* prefix = nullptr;
  csr_t* trainMat = nullptr;
  csr_t* testMat = nullptr;
  csr_t* valMat = nullptr;
  csr_t* graphMat = nullptr;  // never read on the MF path; kept for interface parity
  int facDim = 0;
  int trainNNZ = 0;
  int nUsers = -1;
  int nItems = -1;

  Data(csr_t* p_trainMat, csr_t* p_testMat);   // datastruct.h:109-114 (takes ownership)
  explicit Data(const Params& params);         // datastruct.cpp:3-120
  // convenience for tests/benchmarks: adopt in-memory matrices (takes ownership)
  Data(csr_t* train, csr_t* test, csr_t* val, const char* prefix);
  ~Data();
  Data(const Data&) = delete;
  Data& operator=(const Data&) = delete;

 private:
  void finish();
};
#endif
