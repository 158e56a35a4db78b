// mf_model.h -- Model and ModelMF with the reference's class surface (model.h:22-105,
// modelMF.h:21-57).  The factor matrices live in HBM while a trainer runs; every inner loop
// of the reference's trainers is one call into the C ABI of include/mfx.h.
#ifndef MFHOST_MF_MODEL_H_
#define MFHOST_MF_MODEL_H_
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <random>
#include <vector>

#include "mfx.h"
#include "params_data.h"

// std::shuffle(a.begin(), a.end(), g) bit for bit (same permutation, same generator state), with the swap positions drawn a block
// ahead so that their cache lines are on the way when the swaps follow (mf_model.cpp)
void mfhShuffle(std::vector<size_t>& a, std::mt19937& g);
void mfhShuffle(std::vector<uint32_t>& a, std::mt19937& g);      // the same swaps on a list of 32-bit entries (half the bytes)
// the positions std::shuffle would swap with for a list of n entries (pos[i] <= i, pos[0] = 0), the generator advanced like the library's
// loop: the swaps themselves then run on the device (mfx_sgd_apply_swaps32).  false: short list or another standard library.
bool mfhShufflePositions(std::vector<uint32_t>& pos, size_t n, std::mt19937& g);
int mfhShuffleForm();      // 0: the library call, 1: block-ahead, 2: block-ahead with the restated generator (self-checks passed)

// constants of const.h:4-12 / modelMF.h:16-17
#define MF_OBJ_ITER 1
#define MF_DISP_ITER 50
#define MF_SAVE_ITER 50
#define MF_CHANCE_ITER 500
#define MF_EPS 1e-5

// Row-major dense float matrix (the reference uses column-major Eigen::MatrixXf; the
// text factor files and every formula are layout independent).
struct DenseF32 {
  int rows = 0, cols = 0;
  std::vector<float> a;
  DenseF32() {}
  DenseF32(int r, int c) : rows(r), cols(c), a((size_t)r * c, 0.0f) {}
  float& operator()(int r, int c) { return a[(size_t)r * cols + c]; }
  float operator()(int r, int c) const { return a[(size_t)r * cols + c]; }
  float* data() { return a.data(); }
  const float* data() const { return a.data(); }
  void fill(float v) { std::fill(a.begin(), a.end(), v); }
  double norm() const {
    double s = 0;
    for (float x : a) s += (double)x * x;
    return std::sqrt(s);
  }
};

// What the library classes throw instead of calling exit(): a failed device call (code = mfx_status, what() = the text of
// mfx_last_error) or a misuse of the class surface (code -100).  The `mf` driver catches it, prints it and exits like the
// reference does; a program that embeds libmfhost.so decides for itself.
// codes of MfxError that stand for the reference's exit() calls on the train path (never taken inside the library: the CLI maps them
// back to the reference's exit status, mfh_* entry points of capi.cpp return them as error codes)
enum { MFH_EXIT_OK = -101 /* model.cpp:1481-1484: exit(0) */, MFH_EXIT_FAIL = -102 /* unreadable matrix: GKlib aborts, exit(-1) */ };
class MfxError : public std::runtime_error {
 public:
  int code;
  MfxError(int c, const std::string& msg) : std::runtime_error(msg), code(c) {}
};

// One mfx_ctx with the three rating matrices of a Data uploaded.  Shared by a trainer's
// model (snapshot CURRENT) and its bestModel (snapshot BEST).
class MfxSession {
 public:
  mfx_ctx* ctx = nullptr;
  const csr_t* mats[3] = {nullptr, nullptr, nullptr};
  int nUsers = 0, nItems = 0, K = 0;
  ~MfxSession();
  static std::shared_ptr<MfxSession> open(const Data& data, int nUsers, int nItems, int K);
  int which(const csr_t* m) const;
  void check(int rc, const char* what) const;   // throws MfxError(rc, what + mfx_last_error) on failure
};

class Model {
 public:
  int nUsers = 0, nItems = 0, facDim = 0, trainSeed = -1;
  float origLearnRate = 0, learnRate = 0, rhoRMS = 0, alpha = 0;
  int maxIter = 0;
  float uReg = 0, iReg = 0, sing_a = 0, sing_b = 0;
  DenseF32 uFac, iFac;
  std::vector<float> uBias, iBias;
  std::vector<float> singularVals;   // model.h: Eigen::VectorXf singularVals (set by trainSGDParSVD)
  double mu = 0;

  Model(const Params& params);                                   // model.cpp:2315-2366
  Model(int nUsers, int nItems, int facDim);                     // model.cpp:2310-2311
  Model(const Params& params, int seed);                         // model.cpp:2375-2377
  Model(const Params& params, const char* uFacName, const char* iFacName, int seed);  // :2380-2386
  virtual ~Model() {}

  // the trainer entry points of model.h:56-105 (base class: not implemented, as in the reference)
  typedef std::unordered_set<int> IntSet;
  virtual void train(const Data&, Model&, IntSet&, IntSet&) { notInBase("train"); }
  virtual void trainSGDPar(const Data&, Model&, IntSet&, IntSet&) { notInBase("trainSGDPar"); }
  virtual void trainSGDParSVD(const Data&, Model&, IntSet&, IntSet&) { notInBase("trainSGDParSVD"); }
  virtual void trainUShuffle(const Data&, Model&, IntSet&, IntSet&) { notInBase("trainUShuffle"); }
  virtual void trainALS(const Data&, Model&, IntSet&, IntSet&) { notInBase("trainALS"); }
  virtual void trainCCDPP(const Data&, Model&, IntSet&, IntSet&) { notInBase("trainCCDPP"); }
  virtual void trainCCDPPFreqAdap(const Data&, Model&, IntSet&, IntSet&) { notInBase("trainCCDPPFreqAdap"); }
  virtual void trainCCD(const Data&, Model&, IntSet&, IntSet&) { notInBase("trainCCD"); }
  virtual void hogTrain(const Data&, Model&, IntSet&, IntSet&) { notInBase("hogTrain"); }

  virtual double estRating(int user, int item);                                  // model.cpp:547-549
  double RMSE(csr_t* mat);                                                         // model.cpp:191-211
  double RMSE(csr_t* mat, IntSet& invalidUsers, IntSet& invalidItems);            // model.cpp:214-251
  std::pair<int, double> RMSE(csr_t* mat, IntSet& filtItems, IntSet& invalidUsers, IntSet& invalidItems);   // :348-394
  std::pair<int, double> RMSEU(csr_t* mat, IntSet& filtUsers, IntSet& invalidUsers, IntSet& invalidItems);  // :446-486
  virtual double objective(const Data& data);                                     // model.cpp:1694-1722
  virtual double objective(const Data& data, IntSet& invalidUsers, IntSet& invalidItems);  // :1770-1815
  bool isTerminateModel(Model& bestModel, const Data& data, int iter, int& bestIter, double& bestObj,
                        double& prevObj, double& bestValRMSE, double& prevValRMSE, IntSet& invalidUsers,
                        IntSet& invalidItems);
  bool isTerminateModelSing(Model& bestModel, const Data& data, int iter, int& bestIter, double& bestObj,
                            double& prevObj, double& bestValRMSE, double& prevValRMSE, IntSet& invalidUsers,
                            IntSet& invalidItems);                                  // model.cpp:1543-1612
  double objectiveSing(const Data& data, IntSet& invalidUsers, IntSet& invalidItems);   // model.cpp:1818-1865
  bool terminateImpl(bool sing, Model& bestModel, const Data& data, int iter, int& bestIter, double& bestObj,
                     double& prevObj, double& bestValRMSE, double& prevValRMSE, IntSet& invalidUsers,
                     IntSet& invalidItems);                                     // model.cpp:1471-1540
  std::string modelSignature();                                                    // model.cpp:11-19
  void display();
  void saveFacs(std::string prefix);                                               // model.cpp:89-101
  void loadFacs(std::string prefix);                                               // model.cpp:104-128
  void saveBinFacs(std::string prefix);                                            // model.cpp:131-140
  void loadBinFacs(std::string prefix);                                            // model.cpp:143-159

  // ---- device mirror ------------------------------------------------------------
  std::shared_ptr<MfxSession> dev;   // set while/after a trainer ran
  int devSnap = MFX_SNAP_CURRENT;    // which device snapshot this object stands for
  bool hostStale = false;            // device copy is newer than uFac/iFac
  double lastLoopSeconds = 0;        // wall time of the last trainer's iteration loop (updates + termination checks)
  int lastIters = 0;
  bool rolledBack = false;           // the last isTerminateModel call found NaN and restored bestModel (model.cpp:1486-1498)
  virtual void syncHost();           // download uFac/iFac if hostStale
  virtual void pushToDevice();       // upload uFac/iFac to the CURRENT snapshot
  // scalar fields of `*this = other` without touching the factor storage
  void copyScalarsFrom(const Model& o);

 protected:
  void notInBase(const char* what);
  void attach(const Data& data);     // open a session for data (if none) and upload the factors
  // getInvalidUsersItems + modelMF.cpp:40-45 on the device, returned as sets
  void deviceInvalid(const Data& data, IntSet& invalidUsers, IntSet& invalidItems);
  virtual void evalDevice(const csr_t* mat, int withNorms, mfx_eval_out* out);
  // true while objective() is the base formula, so that isTerminateModel may take the objective and the
  // validation RMSE from one mfx_eval2 call; a class that overrides objective() returns false
  virtual bool baseObjective() const { return true; }
};

class ModelMF : public Model {
 public:
  ModelMF(int nUsers, int nItems, int facDim) : Model(nUsers, nItems, facDim) {}
  ModelMF(const Params& params) : Model(params) {}
  ModelMF(const Params& params, int seed) : Model(params, seed) {}
  ModelMF(const Params& params, const char* uFacName, const char* iFacName, int seed)
      : Model(params, uFacName, iFacName, seed) {}
  void train(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;
  void trainSGDPar(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;
  void trainSGDParSVD(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;
  void trainUShuffle(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;
  void trainALS(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;
  void trainCCDPP(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;
  void trainCCDPPFreqAdap(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;
  void trainCCD(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;
  void hogTrain(const Data& data, Model& bestModel, IntSet& invalidUsers, IntSet& invalidItems) override;

 protected:
  enum Kind { K_SGD, K_HOG, K_SGDPAR, K_SGDU, K_ALS, K_CCDPP, K_CCDPP_FA, K_CCD, K_SGDPARSVD, K_IFW, K_TMF, K_TMFD };
  void run(Kind kind, const char* name, const Data& data, Model& bestModel, IntSet& invalidUsers,
           IntSet& invalidItems);
  // called once the invalid sets are known and before the first objective: sibling models set their state here
  virtual void beforeLoop(Kind, const Data&, IntSet&, IntSet&) {}
  virtual void afterLoop(Kind) {}
};

// text factor files (io.cpp:83-154)
void writeMat(const DenseF32& mat, int nrows, int ncols, const char* fileName);
bool readMat(DenseF32& mat, int nrows, int ncols, const char* fileName);     // text, or binary when the name ends in .binmat
void writeMatBin(const DenseF32& mat, int nrows, int ncols, const char* fileName);
bool readMatBin(DenseF32& mat, int nrows, int ncols, const char* fileName);
bool isFileExist(const char* fileName);
#endif
