// mf_model.cpp -- Model / ModelMF on top of the C ABI (include/mfx.h).
//
// What stays on the host is what the reference keeps in its control flow: the
// hyper-parameters, the RNG (std::mt19937 / std::shuffle exactly as modelMF.cpp uses them),
// the termination rules of Model::isTerminateModel, logging and the factor files.  Every loop
// over ratings, rows or columns is a device call.
#include "mf_model.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <iostream>
#include <mutex>
#include <numeric>
#include <omp.h>
#include <random>
#include <sstream>
#include <thread>
#include <unordered_set>

// ---------------------------------------------------------------------------
// device session
// ---------------------------------------------------------------------------
MfxSession::~MfxSession() {
  if (ctx) mfx_destroy(ctx);
}

void MfxSession::check(int rc, const char* what) const {
  if (rc == MFX_OK) return;
  throw MfxError(rc, std::string("mfx error ") + std::to_string(rc) + " in " + what + ": " + mfx_last_error(ctx));
}

static void upload(MfxSession& s, int which, const csr_t* m, int nItems) {
  if (!m) return;
  // ncols is declared as the model's nItems so that every index is in range for the kernels
  s.check(mfx_set_csr(s.ctx, which, m->nrows, std::max(m->ncols, 0), m->rowptr, m->rowind, m->rowval,
                      which == MFX_MAT_TRAIN ? m->colptr : nullptr, which == MFX_MAT_TRAIN ? m->colind : nullptr,
                      which == MFX_MAT_TRAIN ? m->colval : nullptr),
          "mfx_set_csr");
  s.mats[which] = m;
  (void)nItems;
}

std::shared_ptr<MfxSession> MfxSession::open(const Data& data, int nUsers, int nItems, int K) {
  auto s = std::make_shared<MfxSession>();
  const char* dv = getenv("MFX_DEVICE");
  int rc = mfx_create(dv ? atoi(dv) : 0, &s->ctx);
  if (rc != MFX_OK) throw MfxError(rc, std::string("mfx_create failed (") + std::to_string(rc) + "): " + mfx_last_error(nullptr));
  s->nUsers = nUsers; s->nItems = nItems; s->K = K;
  upload(*s, MFX_MAT_TRAIN, data.trainMat, nItems);
  upload(*s, MFX_MAT_VAL, data.valMat, nItems);
  upload(*s, MFX_MAT_TEST, data.testMat, nItems);
  s->check(mfx_set_model(s->ctx, nUsers, nItems, K), "mfx_set_model");
  return s;
}

int MfxSession::which(const csr_t* m) const {
  for (int w = 0; w < 3; w++)
    if (mats[w] == m && m) return w;
  return -1;
}

// ---------------------------------------------------------------------------
// factor files (io.cpp:83-154): one row per line, "v " per value, default precision
// ---------------------------------------------------------------------------
void writeMat(const DenseF32& mat, int nrows, int ncols, const char* fileName) {
  std::ofstream op(fileName);
  if (!op.is_open()) return;
  for (int i = 0; i < nrows; i++) {
    for (int j = 0; j < ncols; j++) op << mat(i, j) << " ";
    op << std::endl;
  }
}

// io.cpp:172-184: every value as one double, row by row -- lossless for float factors
void writeMatBin(const DenseF32& mat, int nrows, int ncols, const char* fileName) {
  std::ofstream op(fileName, std::ios::binary);
  if (!op.is_open()) return;
  std::vector<double> row((size_t)ncols);
  for (int i = 0; i < nrows; i++) {
    for (int j = 0; j < ncols; j++) row[j] = mat(i, j);
    op.write((const char*)row.data(), sizeof(double) * (size_t)ncols);
  }
}
// Reads what writeMatBin wrote.  (The reference's reader, io.cpp:291-302, reads 8 bytes into each 4-byte float
// of the Eigen matrix; that overrun is not reproduced.)
bool readMatBin(DenseF32& mat, int nrows, int ncols, const char* fileName) {
  std::ifstream in(fileName, std::ios::binary);
  if (!in.is_open()) {
    std::cout << "\nCan't open file: " << fileName << std::endl;
    return false;
  }
  mat = DenseF32(nrows, ncols);
  std::vector<double> row((size_t)ncols);
  for (int i = 0; i < nrows; i++) {
    in.read((char*)row.data(), sizeof(double) * (size_t)ncols);
    if (in.gcount() != (std::streamsize)(sizeof(double) * (size_t)ncols)) return false;
    for (int j = 0; j < ncols; j++) mat(i, j) = (float)row[j];
  }
  return true;
}
static bool isBinMat(const char* name) {
  const std::string s(name);
  return s.size() > 7 && s.compare(s.size() - 7, 7, ".binmat") == 0;
}

bool readMat(DenseF32& mat, int nrows, int ncols, const char* fileName) {
  if (isBinMat(fileName)) return readMatBin(mat, nrows, ncols, fileName);
  std::cout << "\nReading ... " << fileName << " nrows: " << nrows << " ncols: " << ncols << std::endl;
  std::ifstream in(fileName);
  if (!in.is_open()) {
    std::cout << "\nCan't open file: " << fileName << std::endl;
    return false;
  }
  mat = DenseF32(nrows, ncols);
  std::string line;
  int i = 0;
  while (std::getline(in, line) && i < nrows) {
    int j = 0;
    size_t b = 0;
    while (b < line.size()) {
      size_t e = line.find(' ', b);
      if (e == std::string::npos) e = line.size();
      if (e > b) {
        if (j >= ncols) return false;
        mat(i, j++) = (float)std::stod(line.substr(b, e - b));
      }
      b = e + 1;
    }
    if (j != ncols) return false;   // the reference asserts j == ncols (io.cpp:110)
    i++;
  }
  return i == nrows;
}

bool isFileExist(const char* fileName) {
  std::ifstream f(fileName);
  return f.good();
}

// ---------------------------------------------------------------------------
// Model
// ---------------------------------------------------------------------------
Model::Model(int nUsers, int nItems, int facDim) : nUsers(nUsers), nItems(nItems), facDim(facDim) {}

Model::Model(const Params& params) {
  nUsers = params.nUsers;
  nItems = params.nItems;
  facDim = params.facDim;
  uReg = params.uReg;
  iReg = params.iReg;
  sing_a = params.uReg;
  sing_b = params.iReg;
  learnRate = params.learnRate;
  origLearnRate = params.learnRate;
  rhoRMS = params.rhoRMS;
  alpha = params.alpha;
  maxIter = params.maxIter;
  trainSeed = -1;
  // one minstd_rand0 stream, U(-0.01f, 0.01f) drawn as double: uFac, iFac, uBias, iBias
  std::default_random_engine generator(params.seed);
  const float lb = -0.01, ub = 0.01;
  std::uniform_real_distribution<double> dist(lb, ub);
  std::cout << "lb = " << lb << " ub = " << ub << std::endl;
  uFac = DenseF32(nUsers, facDim);
  for (float& x : uFac.a) x = (float)dist(generator);
  iFac = DenseF32(nItems, facDim);
  for (float& x : iFac.a) x = (float)dist(generator);
  uBias.resize(nUsers);
  for (float& x : uBias) x = (float)dist(generator);
  iBias.resize(nItems);
  for (float& x : iBias) x = (float)dist(generator);
}

Model::Model(const Params& params, int seed) : Model(params) { trainSeed = seed; }

Model::Model(const Params& params, const char* uFacName, const char* iFacName, int seed) : Model(params, seed) {
  std::cout << "\nLoading user factors: " << uFacName;
  readMat(uFac, nUsers, facDim, uFacName);
  std::cout << "\nLoading item factors: " << iFacName;
  readMat(iFac, nItems, facDim, iFacName);
}

void Model::notInBase(const char* what) { std::cerr << "\n" << what << ": method not in base class" << std::endl; }

void Model::copyScalarsFrom(const Model& o) {
  nUsers = o.nUsers; nItems = o.nItems; facDim = o.facDim; trainSeed = o.trainSeed;
  origLearnRate = o.origLearnRate; learnRate = o.learnRate; rhoRMS = o.rhoRMS; alpha = o.alpha;
  maxIter = o.maxIter; uReg = o.uReg; iReg = o.iReg; sing_a = o.sing_a; sing_b = o.sing_b; mu = o.mu;
  singularVals = o.singularVals;
}

std::string Model::modelSignature() {
  return std::to_string(nUsers) + "X" + std::to_string(nItems) + "_" + std::to_string(facDim) + "_" +
         std::to_string(uReg) + "_" + std::to_string(iReg) + "_" + std::to_string(origLearnRate);
}

void Model::display() {
  std::cout << "nUsers: " << nUsers << " nItems: " << nItems << std::endl;
  std::cout << "facDim: " << facDim << std::endl;
  std::cout << "uReg: " << uReg << " iReg: " << iReg << std::endl;
  std::cout << "learnRate: " << learnRate << std::endl;
  std::cout << "trainSeed: " << trainSeed;
}

void Model::syncHost() {
  if (!hostStale || !dev) return;
  if (uFac.rows != nUsers || uFac.cols != facDim) uFac = DenseF32(nUsers, facDim);
  if (iFac.rows != nItems || iFac.cols != facDim) iFac = DenseF32(nItems, facDim);
  dev->check(mfx_get_factors(dev->ctx, devSnap, uFac.data(), iFac.data(), MFX_ROWMAJOR), "mfx_get_factors");
  hostStale = false;
}

void Model::pushToDevice() {
  dev->check(mfx_set_factors(dev->ctx, uFac.data(), iFac.data(), MFX_ROWMAJOR), "mfx_set_factors");
}

void Model::saveFacs(std::string prefix) {
  if (getenv("MFX_NO_SAVE")) return;
  syncHost();
  std::cout << "Saving model... " << prefix << std::endl;
  const std::string sign = modelSignature();
  const std::string uName = prefix + "_uFac_" + sign + ".mat";
  writeMat(uFac, nUsers, facDim, uName.c_str());
  std::cout << "uFac Norm: " << uFac.norm() << std::endl;
  const std::string iName = prefix + "_iFac_" + sign + ".mat";
  writeMat(iFac, nItems, facDim, iName.c_str());
  std::cout << "iFac Norm: " << iFac.norm() << std::endl;
  if (getenv("MFX_SAVE_BIN")) saveBinFacs(prefix);   // lossless companion files (the text files keep 6 digits)
}

// model.cpp:131-140
void Model::saveBinFacs(std::string prefix) {
  if (getenv("MFX_NO_SAVE")) return;
  syncHost();
  const std::string sign = modelSignature();
  writeMatBin(uFac, nUsers, facDim, (prefix + "_uFac_" + sign + ".binmat").c_str());
  writeMatBin(iFac, nItems, facDim, (prefix + "_iFac_" + sign + ".binmat").c_str());
}
// model.cpp:143-159
void Model::loadBinFacs(std::string prefix) {
  const std::string sign = modelSignature();
  const std::string uName = prefix + "_uFac_" + sign + ".binmat";
  if (isFileExist(uName.c_str())) {
    std::cout << "Loading user factors: " << uName << std::endl;
    readMatBin(uFac, nUsers, facDim, uName.c_str());
  }
  const std::string iName = prefix + "_iFac_" + sign + ".binmat";
  if (isFileExist(iName.c_str())) {
    std::cout << "Loading item factors: " << iName << std::endl;
    readMatBin(iFac, nItems, facDim, iName.c_str());
  }
  if (dev && devSnap == MFX_SNAP_CURRENT) pushToDevice();
}

void Model::loadFacs(std::string prefix) {
  const std::string sign = modelSignature();
  const std::string uName = prefix + "_uFac_" + sign + ".mat";
  std::cout << "Loading user factors: " << uName << std::endl;
  if (isFileExist(uName.c_str())) {
    readMat(uFac, nUsers, facDim, uName.c_str());
    std::cout << "uFac Norm: " << uFac.norm() << std::endl;
  } else {
    std::cout << "File doesn't exist: " << uName << std::endl;
  }
  const std::string iName = prefix + "_iFac_" + sign + ".mat";
  std::cout << "Loading item factors: " << iName << std::endl;
  if (isFileExist(iName.c_str())) {
    readMat(iFac, nItems, facDim, iName.c_str());
    std::cout << "iFac Norm: " << iFac.norm() << std::endl;
  } else {
    std::cout << "File doesn't exist: " << iName << std::endl;
  }
  if (dev && devSnap == MFX_SNAP_CURRENT) pushToDevice();
}

double Model::estRating(int user, int item) {
  syncHost();
  float s = uFac(user, 0) * iFac(item, 0);
  for (int k = 1; k < facDim; k++) s = s + uFac(user, k) * iFac(item, k);
  return s;
}

void Model::attach(const Data& data) {
  if (!dev || dev->mats[MFX_MAT_TRAIN] != data.trainMat) {
    dev = MfxSession::open(data, nUsers, nItems, facDim);
    devSnap = MFX_SNAP_CURRENT;
    uint8_t dummy = 0;
    (void)dummy;
    dev->check(mfx_compute_invalid(dev->ctx, nullptr, nullptr), "mfx_compute_invalid");
  }
  pushToDevice();
  hostStale = false;
}

void Model::deviceInvalid(const Data& data, IntSet& invalidUsers, IntSet& invalidItems) {
  (void)data;
  std::vector<uint8_t> iu(nUsers), ii(nItems);
  dev->check(mfx_compute_invalid(dev->ctx, iu.data(), ii.data()), "mfx_compute_invalid");
  for (int u = 0; u < nUsers; u++) if (iu[u]) invalidUsers.insert(u);
  for (int i = 0; i < nItems; i++) if (ii[i]) invalidItems.insert(i);
}

void Model::evalDevice(const csr_t* mat, int withNorms, mfx_eval_out* out) {
  const int w = dev ? dev->which(mat) : -1;
  if (w < 0) {
    throw MfxError(-100, "Model: matrix is not part of the device session");
  }
  dev->check(mfx_eval(dev->ctx, w, devSnap, withNorms, out), "mfx_eval");
}

// model.cpp:214-251.  The masks on the device are the ones getInvalidUsersItems yields for the
// session's train matrix, i.e. what every caller in the reference passes here.
double Model::RMSE(csr_t* mat, IntSet& invalidUsers, IntSet& invalidItems) {
  (void)invalidUsers; (void)invalidItems;
  if (!mat) return NAN;
  if (!dev || dev->which(mat) < 0) {
    throw MfxError(-100, "Model::RMSE: no device session for this matrix (call a trainer first)");
  }
  mfx_eval_out o;
  evalDevice(mat, 0, &o);
  return std::sqrt(o.sse / (double)o.n);
}

// model.cpp:348-394 / :446-486: RMSE over the ratings of the given items / users only; (count, rmse)
static std::pair<int, double> filteredRMSE(Model& md, csr_t* mat, const Model::IntSet* users, const Model::IntSet* items) {
  if (!mat || !md.dev || md.dev->which(mat) < 0) {
    throw MfxError(-100, "Model::RMSE(filtered): no device session for this matrix (call a trainer first)");
  }
  std::vector<uint8_t> ku, ki;
  if (users) { ku.assign((size_t)md.nUsers, 0); for (int u : *users) if (u >= 0 && u < md.nUsers) ku[u] = 1; }
  if (items) { ki.assign((size_t)md.nItems, 0); for (int i : *items) if (i >= 0 && i < md.nItems) ki[i] = 1; }
  mfx_eval_out o;
  md.dev->check(mfx_eval_filtered(md.dev->ctx, md.dev->which(mat), md.devSnap, users ? ku.data() : nullptr,
                                  items ? ki.data() : nullptr, &o), "mfx_eval_filtered");
  return std::make_pair((int)o.n, std::sqrt(o.sse / (double)o.n));
}
std::pair<int, double> Model::RMSE(csr_t* mat, IntSet& filtItems, IntSet& invalidUsers, IntSet& invalidItems) {
  (void)invalidUsers; (void)invalidItems;
  return filteredRMSE(*this, mat, nullptr, &filtItems);
}
std::pair<int, double> Model::RMSEU(csr_t* mat, IntSet& filtUsers, IntSet& invalidUsers, IntSet& invalidItems) {
  (void)invalidUsers; (void)invalidItems;
  return filteredRMSE(*this, mat, &filtUsers, nullptr);
}

// model.cpp:191-211 (no masks).  On the train matrix the masked and unmasked sums coincide
// (invalid users/items have no train rating), which is the only use on the training path
// ("Obj b4 svd" print, modelMF.cpp:16-18).
double Model::RMSE(csr_t* mat) {
  IntSet a, b;
  return RMSE(mat, a, b);
}

double Model::objective(const Data& data, IntSet& invalidUsers, IntSet& invalidItems) {
  (void)invalidUsers; (void)invalidItems;
  mfx_eval_out o;
  evalDevice(data.trainMat, 1, &o);
  // model.cpp:1793-1810: rmse + uRegErr*uReg + iRegErr*iReg with float uReg/iReg
  return o.sse + o.unorm2 * uReg + o.inorm2 * iReg;
}

double Model::objective(const Data& data) {
  IntSet a, b;
  return objective(data, a, b);
}

// model.cpp:1471-1540
bool Model::isTerminateModel(Model& bestModel, const Data& data, int iter, int& bestIter, double& bestObj,
                             double& prevObj, double& bestValRMSE, double& prevValRMSE, IntSet& invalidUsers,
                             IntSet& invalidItems) {
  return terminateImpl(false, bestModel, data, iter, bestIter, bestObj, prevObj, bestValRMSE, prevValRMSE, invalidUsers,
                       invalidItems);
}
// model.cpp:1543-1612: the same rules on objectiveSing
bool Model::isTerminateModelSing(Model& bestModel, const Data& data, int iter, int& bestIter, double& bestObj,
                                 double& prevObj, double& bestValRMSE, double& prevValRMSE, IntSet& invalidUsers,
                                 IntSet& invalidItems) {
  return terminateImpl(true, bestModel, data, iter, bestIter, bestObj, prevObj, bestValRMSE, prevValRMSE, invalidUsers,
                       invalidItems);
}
// model.cpp:1818-1865: squared error + sum_k x_k^2 * singularVals(k) over the valid users and items
double Model::objectiveSing(const Data& data, IntSet& invalidUsers, IntSet& invalidItems) {
  (void)invalidUsers; (void)invalidItems;
  const int w = dev ? dev->which(data.trainMat) : -1;
  if (w < 0 || (int)singularVals.size() != facDim) {
    throw MfxError(-100, "Model::objectiveSing: needs a device session and facDim singular values");
  }
  mfx_eval_out o;
  dev->check(mfx_eval_weighted(dev->ctx, w, devSnap, singularVals.data(), &o), "mfx_eval_weighted");
  return o.sse + o.unorm2 + o.inorm2;
}

bool Model::terminateImpl(bool sing, Model& bestModel, const Data& data, int iter, int& bestIter, double& bestObj,
                          double& prevObj, double& bestValRMSE, double& prevValRMSE, IntSet& invalidUsers,
                          IntSet& invalidItems) {
  bool ret = false;
  rolledBack = false;
  double currObj, currValRMSE = -1;
  const int wt = dev ? dev->which(data.trainMat) : -1, wv = dev && data.valMat ? dev->which(data.valMat) : -1;
  if (!sing && baseObjective() && wt >= 0 && wv >= 0) {
    // objective(data, ...) and RMSE(valMat, ...) from one device round trip
    mfx_eval_out ot, ov;
    dev->check(mfx_eval2(dev->ctx, wt, 1, wv, 0, devSnap, &ot, &ov), "mfx_eval2");
    currObj = ot.sse + ot.unorm2 * uReg + ot.inorm2 * iReg;
    currValRMSE = std::sqrt(ov.sse / (double)ov.n);
  } else {
    currObj = sing ? objectiveSing(data, invalidUsers, invalidItems) : objective(data, invalidUsers, invalidItems);
    if (data.valMat) {
      currValRMSE = RMSE(data.valMat, invalidUsers, invalidItems);
    } else {
      // model.cpp:1481-1484 prints this and calls exit(0); a library entered through the C API must not end its caller's process:
      // MfxError(MFH_EXIT_OK) -- the CLI (mf_main.cpp) turns it back into the reference's exit status 0
      std::cerr << "\nNo validation data" << std::endl;
      throw MfxError(MFH_EXIT_OK, "No validation data");
    }
  }
  if (currObj != currObj || currValRMSE != currValRMSE) {
    std::cout << "Found nan " << std::endl;
    if (learnRate > 1e-5) {
      // *this = bestModel; learnRate = learnRate/2
      dev->check(mfx_restore_best(dev->ctx), "mfx_restore_best");
      copyScalarsFrom(bestModel);
      hostStale = true;
      learnRate = learnRate / 2;
      rolledBack = true;
      return false;
    }
    return true;
  }
  if (currValRMSE < bestValRMSE) {
    // bestModel = *this
    dev->check(mfx_snapshot_best(dev->ctx), "mfx_snapshot_best");
    bestModel.copyScalarsFrom(*this);
    bestModel.hostStale = true;
    bestValRMSE = currValRMSE;
    bestIter = iter;
  }
  if (iter - bestIter >= 100) {
    if (learnRate > 1e-5) learnRate = learnRate / 2;
  }
  if (iter - bestIter >= MF_CHANCE_ITER) {
    printf("\nNOT CONVERGED: bestIter:%d bestObj: %.10e bestValRMSE: %.10e currIter:%d currObj: %.10e "
           "currValRMSE: %.10e", bestIter, bestObj, bestValRMSE, iter, currObj, currValRMSE);
    ret = true;
  }
  if (fabs(prevObj - currObj) < MF_EPS) {
    printf("\nConverged in iteration: %d prevObj: %.10e currObj: %.10e bestValRMSE: %.10e", iter, prevObj,
           currObj, bestValRMSE);
    ret = true;
  }
  prevObj = currObj;
  prevValRMSE = currValRMSE;
  return ret;
}

// ---------------------------------------------------------------------------
// ModelMF trainers
// ---------------------------------------------------------------------------
// ---- std::shuffle of the epoch's index list (modelMF.cpp:76-81), the same permutation in less time -----------------------------
// The exact replay of ModelMF::train needs the reference's visiting order, i.e. libstdc++'s std::shuffle of the 16 M indices with
// the model's mt19937 every epoch: 0.4 s on the host at the ML-20M shape, next to 19 ms for the replay itself on the GPU.  For
// lists longer than 65 536 entries libstdc++ runs the plain loop `for i: swap(a[i], a[d(g, {0, i})])` (bits/stl_algo.h; the
// two-positions-per-draw path covers shorter lists) and spends its time on the cache misses of a[j].  The positions depend on the
// generator alone: they are drawn a block ahead -- same distribution object, same calls, same order -- their lines are requested,
// and the swaps follow in order.  A self-check against std::shuffle on first use (same seed, a list just above the threshold)
// switches this off for good on a standard library that shuffles differently.
namespace {
template <class T>
void shuffleAhead(std::vector<T>& a, std::mt19937& g) {
  const size_t n = a.size();
  const uint64_t urngrange = (uint64_t)g.max() - (uint64_t)g.min();
  if (n < 2 || urngrange / n >= n) { std::shuffle(a.begin(), a.end(), g); return; }
  std::uniform_int_distribution<unsigned long> d;
  typedef std::uniform_int_distribution<unsigned long>::param_type P;
  constexpr size_t B = 128;
  size_t j[B];
  T* p = a.data();
  for (size_t i0 = 1; i0 < n; i0 += B) {
    const size_t m = std::min(B, n - i0);
    for (size_t k = 0; k < m; k++) { j[k] = d(g, P(0, i0 + k)); __builtin_prefetch(p + j[k], 1); }
    for (size_t k = 0; k < m; k++) std::swap(p[i0 + k], p[j[k]]);
  }
}
// mt19937 and libstdc++'s uniform_int_distribution, restated for speed: the generator's state leaves and re-enters the std::mt19937
// through its stream operators (libstdc++ writes the 624 words and the position), a twist tempers all 624 outputs in one
// vectorisable loop, and the position of entry i is Lemire's multiply-shift with its rare rejection, as bits/uniform_int_dist.h
// has it for a 32-bit generator since GCC 11 (`_S_nd`).  Only ever used behind shuffleAheadIsStd(): another library version
// (GCC 10 divides) fails that check and the library's own calls are used.
struct MtBulk {
  uint32_t x[624], out[624];
  size_t p = 624;
  bool load(const std::mt19937& g) {
    std::stringstream ss;
    ss << g;
    for (int k = 0; k < 624; k++) { unsigned long v; if (!(ss >> v)) return false; x[k] = (uint32_t)v; }
    unsigned long pp;
    if (!(ss >> pp) || pp > 624) return false;
    p = pp;
    temperAll();
    return true;
  }
  bool store(std::mt19937& g) const {
    std::stringstream ss;
    for (int k = 0; k < 624; k++) ss << x[k] << ' ';
    ss << p;
    return (bool)(ss >> g);
  }
  // (twist and tempering are plain integer loops over 624 words: cloned for AVX2 where the CPU has it -- the drawing thread is the
  //  slower of the two, 3 ns per position with the baseline SSE2 code)
  __attribute__((target_clones("avx2", "default"))) void temperAll() {
    for (int k = 0; k < 624; k++) {
      uint32_t y = x[k];
      y ^= y >> 11;
      y ^= (y << 7) & 0x9d2c5680u;
      y ^= (y << 15) & 0xefc60000u;
      y ^= y >> 18;
      out[k] = y;
    }
  }
  __attribute__((target_clones("avx2", "default"))) void twist() {
    constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;
    for (int k = 0; k < 624 - 397; k++) {
      const uint32_t y = (x[k] & UP) | (x[k + 1] & LO);
      x[k] = x[k + 397] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
    }
    for (int k = 624 - 397; k < 623; k++) {
      const uint32_t y = (x[k] & UP) | (x[k + 1] & LO);
      x[k] = x[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
    }
    const uint32_t y = (x[623] & UP) | (x[0] & LO);
    x[623] = x[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
    p = 0;
    temperAll();
  }
  inline uint32_t next() {
    if (p >= 624) twist();
    return out[p++];
  }
  // d(g, param_type(0, range - 1)) of std::uniform_int_distribution<unsigned long> on std::mt19937, 1 <= range < 2^32
  inline size_t below(uint32_t range) {
    uint64_t product = (uint64_t)next() * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
      const uint32_t threshold = (0u - range) % range;
      while (low < threshold) {
        product = (uint64_t)next() * (uint64_t)range;
        low = (uint32_t)product;
      }
    }
    return (size_t)(product >> 32);
  }
};

// The same on two threads for long lists: this thread draws the positions with the library's generator and distribution objects
// (about half of the time), a helper requests their lines and swaps -- in order, block by block, through a ring of position blocks.
template <class T>
void shuffleAheadPair(std::vector<T>& a, std::mt19937& g, size_t block) {
  const size_t n = a.size();
  const uint64_t urngrange = (uint64_t)g.max() - (uint64_t)g.min();
  if (n < 2 || urngrange / n >= n) { std::shuffle(a.begin(), a.end(), g); return; }
  constexpr size_t RING = 8;
  std::vector<size_t> ring(RING * block);
  std::atomic<size_t> produced(0), consumed(0);
  const size_t nblocks = (n - 1 + block - 1) / block;
  T* p = a.data();
  double waitSwap = 0, waitDraw = 0;          // MFX_TIME_LOOP=1: which of the two threads waits for the other
  static const bool timeIt = getenv("MFX_TIME_LOOP") && atoi(getenv("MFX_TIME_LOOP")) != 0;
  std::thread helper([&] {
    for (size_t b = 0; b < nblocks; b++) {
      if (timeIt && produced.load(std::memory_order_acquire) <= b) {
        const auto w0 = std::chrono::steady_clock::now();
        while (produced.load(std::memory_order_acquire) <= b) std::this_thread::yield();
        waitSwap += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
      }
      while (produced.load(std::memory_order_acquire) <= b) std::this_thread::yield();
      const size_t* j = ring.data() + (b % RING) * block;
      const size_t i0 = 1 + b * block, m = std::min(block, n - i0);
      constexpr size_t AHEAD = 64;
      for (size_t k = 0; k < std::min(AHEAD, m); k++) __builtin_prefetch(p + j[k], 1);
      for (size_t k = 0; k < m; k++) {
        if (k + AHEAD < m) __builtin_prefetch(p + j[k + AHEAD], 1);
        std::swap(p[i0 + k], p[j[k]]);
      }
      consumed.store(b + 1, std::memory_order_release);
    }
  });
  std::uniform_int_distribution<unsigned long> d;
  typedef std::uniform_int_distribution<unsigned long>::param_type P;
  for (size_t b = 0; b < nblocks; b++) {
    if (timeIt && b - consumed.load(std::memory_order_acquire) >= RING) {
      const auto w0 = std::chrono::steady_clock::now();
      while (b - consumed.load(std::memory_order_acquire) >= RING) std::this_thread::yield();
      waitDraw += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    }
    while (b - consumed.load(std::memory_order_acquire) >= RING) std::this_thread::yield();
    size_t* j = ring.data() + (b % RING) * block;
    const size_t i0 = 1 + b * block, m = std::min(block, n - i0);
    for (size_t k = 0; k < m; k++) j[k] = d(g, P(0, i0 + k));
    produced.store(b + 1, std::memory_order_release);
  }
  helper.join();
  if (timeIt) fprintf(stderr, "[mfh] shuffle of %zu entries: the drawing thread waited %.1f ms for the swapping one, the swapping thread %.1f ms for draws\n", n, waitDraw, waitSwap);
}

// Three threads, with the restated generator and distribution (behind shuffleFastIsStd() only): one twists and tempers the
// generator's blocks of 624 words a few blocks ahead, this one turns them into swap positions (Lemire's multiply-shift, the rare
// rejection drawing again from the same stream), a third requests the positions' lines and swaps.  On the GPU box's host the two
// threads of shuffleAheadPair were level at 2.5 ns per entry each -- drawing = generating + scaling; split, and with 32-bit lists
// to swap, every stage has less to do.  The generator's state afterwards is the state of the block the last draw came from.
template <class T>
bool shuffleAheadTriple(std::vector<T>& a, std::mt19937& g, size_t block) {
  const size_t n = a.size();
  const uint64_t urngrange = (uint64_t)g.max() - (uint64_t)g.min();
  if (n < 2 || urngrange / n >= n || n >= ((size_t)1 << 32)) return false;
  constexpr size_t RAWS = 8, RING = 8;
  std::vector<MtBulk> raw(RAWS);                       // (everything that allocates happens before the helpers start)
  std::vector<uint32_t> ring(RING * block);
  if (!raw[0].load(g)) return false;
  std::atomic<size_t> rawProduced(1), rawReleased(0), produced(0), consumed(0);
  std::atomic<bool> stop(false);
  const size_t nblocks = (n - 1 + block - 1) / block;
  T* p = a.data();
  std::thread gen([&] {
    for (size_t b = 1;; b++) {                         // block b of the raw stream = block b - 1 twisted
      while (b - rawReleased.load(std::memory_order_acquire) >= RAWS) {
        if (stop.load(std::memory_order_acquire)) return;
        std::this_thread::yield();
      }
      if (stop.load(std::memory_order_acquire)) return;
      MtBulk& nx = raw[b % RAWS];
      std::memcpy(nx.x, raw[(b - 1) % RAWS].x, sizeof nx.x);
      nx.twist();
      rawProduced.store(b + 1, std::memory_order_release);
    }
  });
  std::thread swp([&] {
    for (size_t b = 0; b < nblocks; b++) {
      while (produced.load(std::memory_order_acquire) <= b) std::this_thread::yield();
      const uint32_t* j = ring.data() + (b % RING) * block;
      const size_t i0 = 1 + b * block, m = std::min(block, n - i0);
      constexpr size_t AHEAD = 64;
      for (size_t k = 0; k < std::min(AHEAD, m); k++) __builtin_prefetch(p + j[k], 1);
      for (size_t k = 0; k < m; k++) {
        if (k + AHEAD < m) __builtin_prefetch(p + j[k + AHEAD], 1);
        std::swap(p[i0 + k], p[j[k]]);
      }
      consumed.store(b + 1, std::memory_order_release);
    }
  });
  size_t rb = 0, idx = raw[0].p;                       // the next raw number: block rb, word idx
  const uint32_t* out = raw[0].out;
  auto nextRaw = [&]() -> uint32_t {
    if (idx >= 624) {                                  // block rb is used up: release it, take the next one
      rawReleased.store(rb + 1, std::memory_order_release);
      rb++;
      while (rawProduced.load(std::memory_order_acquire) <= rb) std::this_thread::yield();
      out = raw[rb % RAWS].out;
      idx = 0;
    }
    return out[idx++];
  };
  for (size_t b = 0; b < nblocks; b++) {
    while (b - consumed.load(std::memory_order_acquire) >= RING) std::this_thread::yield();
    uint32_t* j = ring.data() + (b % RING) * block;
    const size_t i0 = 1 + b * block, m = std::min(block, n - i0);
    for (size_t k = 0; k < m; k++) {
      const uint32_t range = (uint32_t)(i0 + k + 1);   // d(g, param_type(0, i)) of bits/uniform_int_dist.h, see MtBulk::below
      uint64_t product = (uint64_t)nextRaw() * (uint64_t)range;
      uint32_t low = (uint32_t)product;
      if (low < range) {
        const uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
          product = (uint64_t)nextRaw() * (uint64_t)range;
          low = (uint32_t)product;
        }
      }
      j[k] = (uint32_t)(product >> 32);
    }
    produced.store(b + 1, std::memory_order_release);
  }
  stop.store(true, std::memory_order_release);
  swp.join();
  gen.join();
  MtBulk& last = raw[rb % RAWS];                       // (never released: the generating thread has not touched it again)
  last.p = idx;
  if (!last.store(g)) throw std::runtime_error("mfhShuffle: the generator state could not be handed back");
  return true;
}
// The POSITIONS of std::shuffle alone: pos[i] = d(g, param_type(0, i)) for i = 1 .. n - 1 (pos[0] = 0), what the loop
// `for i: swap(a[i], a[pos[i]])` of bits/stl_algo.h draws for a list of n entries, and the generator left where that loop leaves it.
// Two threads: one twists and tempers the generator's blocks ahead, this one scales them (Lemire's multiply-shift with its rejection,
// as shuffleAheadTriple).  The swaps themselves -- random accesses into the list, the slowest of the three stages on a host core --
// run on the device (mfx_sgd_apply_swaps32).  false: not applicable (short list: the library pairs two positions per draw there).
bool shufflePositions(std::vector<uint32_t>& pos, size_t n, std::mt19937& g) {
  const uint64_t urngrange = (uint64_t)g.max() - (uint64_t)g.min();
  if (n < 2 || urngrange / n >= n || n >= ((size_t)1 << 32)) return false;
  // The generating thread hands the stream over in CHUNKS of 64 blocks of 624 words (a hand-off per block -- 32 000 per ML-20M epoch, each a
  // spin on an atomic with sched_yield -- cost more than the twists: twisting and tempering 20 M words takes 6 ms, scaling them 15 ms, the
  // block-by-block pipeline took 29 - 43 ms).  A chunk remembers the generator state in front of its first block, so that the state
  // behind the last word used can be rebuilt without keeping 32 000 states.
  constexpr size_t CH = 64, NCH = 8;
  struct Chunk { uint32_t startX[624]; uint32_t out[CH * 624]; };
  std::vector<Chunk> ring(NCH);
  MtBulk first;
  pos.resize(n);
  if (!first.load(g)) return false;
  std::atomic<size_t> produced(0), released(0);       // chunks
  std::atomic<bool> stop(false);
  std::thread gen([&] {
    MtBulk cur;
    std::memcpy(cur.x, first.x, sizeof cur.x);
    for (size_t c = 0;; c++) {
      while (c - released.load(std::memory_order_acquire) >= NCH) {
        if (stop.load(std::memory_order_acquire)) return;
        std::this_thread::yield();
      }
      if (stop.load(std::memory_order_acquire)) return;
      Chunk& k = ring[c % NCH];
      std::memcpy(k.startX, cur.x, sizeof cur.x);
      for (size_t b = 0; b < CH; b++) {
        cur.twist();
        std::memcpy(k.out + b * 624, cur.out, sizeof cur.out);
      }
      produced.store(c + 1, std::memory_order_release);
    }
  });
  // the next raw word: block cb of the stream (0 = what is left of the generator's current block, block b >= 1 = word range
  // [(b - 1) % CH * 624, + 624) of chunk (b - 1) / CH), word idx of it
  size_t cb = 0, idx = first.p;
  const uint32_t* out = first.out;
  auto nextBlock = [&] {
    if (cb >= 1 && cb % CH == 0) released.store(cb / CH, std::memory_order_release);     // the chunk that held blocks cb - CH + 1 .. cb is used up
    cb++;
    const size_t c = (cb - 1) / CH;
    while (produced.load(std::memory_order_acquire) <= c) std::this_thread::yield();
    out = ring[c % NCH].out + ((cb - 1) % CH) * 624;
    idx = 0;
  };
  uint32_t* j = pos.data();
  j[0] = 0;
  size_t i = 1;
  // eight positions at a time while their raw words lie in the current block and none of them needs a second look (low < range, the
  // door to the rejection test, opens for range / 2^32 of the draws: < 0.5 % at 20 M entries): a loop the compiler vectorises
  for (;;) {
    while (i + 8 <= n && idx + 8 <= 624) {
      uint64_t prod[8];
      uint32_t any = 0;
      for (int k = 0; k < 8; k++) {
        const uint32_t range = (uint32_t)(i + k + 1);
        prod[k] = (uint64_t)out[idx + k] * (uint64_t)range;
        any |= (uint32_t)((uint32_t)prod[k] < range);
      }
      if (any) break;
      for (int k = 0; k < 8; k++) j[i + k] = (uint32_t)(prod[k] >> 32);
      i += 8;
      idx += 8;
    }
    if (i >= n) break;
    // one position the careful way (also the block's last words and the list's last entries): d(g, param_type(0, i)) of
    // bits/uniform_int_dist.h, see MtBulk::below
    const uint32_t range = (uint32_t)(i + 1);
    if (idx >= 624) nextBlock();
    uint64_t product = (uint64_t)out[idx++] * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
      const uint32_t threshold = (0u - range) % range;
      while (low < threshold) {
        if (idx >= 624) nextBlock();
        product = (uint64_t)out[idx++] * (uint64_t)range;
        low = (uint32_t)product;
      }
    }
    j[i] = (uint32_t)(product >> 32);
    i++;
  }
  stop.store(true, std::memory_order_release);
  gen.join();
  // the generator as std::shuffle leaves it: the state of block cb with idx words taken
  MtBulk last;
  if (cb == 0) {
    std::memcpy(last.x, first.x, sizeof last.x);
  } else {
    std::memcpy(last.x, ring[((cb - 1) / CH) % NCH].startX, sizeof last.x);
    for (size_t b = 0; b <= (cb - 1) % CH; b++) last.twist();
  }
  last.p = idx;
  if (!last.store(g)) throw std::runtime_error("mfhShufflePositions: the generator state could not be handed back");
  return true;
}
bool shuffleAheadIsStd() {
  static const bool ok = [] {
    std::vector<size_t> x(70001), y, z;
    std::iota(x.begin(), x.end(), (size_t)0);
    y = x;
    z = x;
    std::mt19937 g1(12345), g2(12345), g3(12345);
    g1.discard(1000); g2.discard(1000); g3.discard(1000);      // (a generator in the middle of its block of 624)
    std::shuffle(x.begin(), x.end(), g1);
    shuffleAhead(y, g2);
    shuffleAheadPair(z, g3, 1000);
    return x == y && g1 == g2 && x == z && g1 == g3;          // the same permutation AND the same generator state afterwards
  }();
  return ok;
}
// ... and whether the restated generator + distribution (MtBulk) are this library's: the same, and over a range of list lengths whose
// position ranges reject often enough to exercise the rejection loop
bool shuffleFastIsStd() {
  static const bool ok = [] {
    for (size_t n : {(size_t)70001, (size_t)300007}) {
      std::vector<size_t> x(n), z;
      std::iota(x.begin(), x.end(), (size_t)0);
      z = x;
      std::mt19937 g1(777), g3(777);
      g1.discard(n % 1000); g3.discard(n % 1000);
      std::shuffle(x.begin(), x.end(), g1);
      if (!shuffleAheadTriple(z, g3, 4096) || !(x == z && g1 == g3)) return false;
      std::vector<uint32_t> w(n);                         // ... and a 32-bit list through the same draws
      std::iota(w.begin(), w.end(), 0u);
      std::mt19937 g4(777);
      g4.discard(n % 1000);
      if (!shuffleAheadTriple(w, g4, 1000) || !(g1 == g4)) return false;
      for (size_t k = 0; k < n; k++)
        if ((size_t)w[k] != x[k]) return false;
    }
    // the rejection path proper: ranges just above 2^31 reject half of the draws
    std::mt19937 g1(99), g2(99);
    MtBulk b;
    if (!b.load(g2)) return false;
    std::uniform_int_distribution<unsigned long> d;
    typedef std::uniform_int_distribution<unsigned long>::param_type P;
    for (int k = 0; k < 20000; k++) {
      const uint32_t range = k % 3 == 0 ? 0x80000001u + (uint32_t)k : (k % 3 == 1 ? 3u + (uint32_t)k : 0xfffffff0u);
      if (d(g1, P(0, (unsigned long)range - 1)) != b.below(range)) return false;
    }
    return b.store(g2) && g1 == g2;
  }();
  return ok;
}
}  // namespace
// which form mfhShuffle takes for a long list: 0 std::shuffle, 1 block-ahead with the library's generator and distribution,
// 2 block-ahead with the restated ones
int mfhShuffleForm() {
  if (!shuffleAheadIsStd()) return 0;
  return shuffleFastIsStd() ? 2 : 1;
}
// std::shuffle(a.begin(), a.end(), g) -- bit for bit, see above (MFX_STD_SHUFFLE=1: the library call itself)
namespace {
template <class T>
void mfhShuffleT(std::vector<T>& a, std::mt19937& g) {
  static const bool plain = getenv("MFX_STD_SHUFFLE") && atoi(getenv("MFX_STD_SHUFFLE")) != 0;
  static const bool single = getenv("MFX_SHUFFLE_THREADS") && atoi(getenv("MFX_SHUFFLE_THREADS")) == 1;
  static const bool slowGen = getenv("MFX_SHUFFLE_LIBGEN") && atoi(getenv("MFX_SHUFFLE_LIBGEN")) != 0;
  if (plain || !shuffleAheadIsStd()) { std::shuffle(a.begin(), a.end(), g); return; }
  if (a.size() >= (size_t)1 << 20 && !single) {
    if (!slowGen && shuffleFastIsStd() && shuffleAheadTriple(a, g, 8192)) return;
    shuffleAheadPair(a, g, 8192);
  } else {
    shuffleAhead(a, g);
  }
}
}  // namespace
// the swap positions of std::shuffle(a.begin(), a.end(), g) for a list of n entries (see shufflePositions); false when the list is
// short or this standard library's generator / distribution are not the restated ones: the caller shuffles on the host then
bool mfhShufflePositions(std::vector<uint32_t>& pos, size_t n, std::mt19937& g) {
  static const bool off = getenv("MFX_STD_SHUFFLE") && atoi(getenv("MFX_STD_SHUFFLE")) != 0;
  if (off || n < ((size_t)1 << 20) || !shuffleAheadIsStd() || !shuffleFastIsStd()) return false;
  return shufflePositions(pos, n, g);
}
void mfhShuffle(std::vector<size_t>& a, std::mt19937& g) { mfhShuffleT(a, g); }
void mfhShuffle(std::vector<uint32_t>& a, std::mt19937& g) { mfhShuffleT(a, g); }

// ---- ModelMF::trainSGDPar's stratification (modelMF.cpp:229-265, 273-304; util.cpp:1077-1107) ------------------
// The reference keeps every part as a std::unordered_set<int> and sweeps a block in that container's iteration order;
// the sets are rebuilt here by the same insertion sequence (same libstdc++ => same order) and flattened once.
namespace {
struct Strata {
  int T = 0;
  std::vector<std::vector<int>> users;   // users of each part, in the iteration order of the reference's set
  std::vector<int> itemPart;             // part of each item, -1: not in any

  template <class Set>
  void deal(std::mt19937& mt, const csr_t* trainMat, const Set& invalidUsers, const Set& invalidItems, int parts) {
    T = parts;
    std::vector<int> trainUsers, trainItems;
    for (int u = 0; u < trainMat->nrows; u++)
      if (!invalidUsers.count(u)) trainUsers.push_back(u);
    for (int item = 0; item < trainMat->ncols; item++)
      if (!invalidItems.count(item)) trainItems.push_back(item);
    std::shuffle(trainUsers.begin(), trainUsers.end(), mt);
    std::shuffle(trainItems.begin(), trainItems.end(), mt);
    // the part index advances AFTER the insert whenever i % perPart == 0 (and i != 0): the first part gets one extra
    auto partOf = [&](size_t n, std::vector<int>& out) {
      const int per = (int)n / T;
      out.resize(n);
      int cur = 0;
      for (size_t i = 0; i < n; i++) {
        out[i] = cur;
        if (i != 0 && per > 0 && (int)i % per == 0 && cur != T - 1) cur++;
      }
    };
    std::vector<int> pu, pi;
    partOf(trainUsers.size(), pu);
    partOf(trainItems.size(), pi);
    std::vector<std::unordered_set<int>> sets((size_t)T);
    for (size_t i = 0; i < trainUsers.size(); i++) sets[(size_t)pu[i]].insert(trainUsers[i]);
    users.assign((size_t)T, std::vector<int>());
    for (int t = 0; t < T; t++) users[(size_t)t].assign(sets[(size_t)t].begin(), sets[(size_t)t].end());
    itemPart.assign((size_t)trainMat->ncols, -1);
    for (size_t i = 0; i < trainItems.size(); i++) itemPart[(size_t)trainItems[i]] = pi[i];
    blockList.clear();
  }

  // sgdUpdateBlockSeq: user parts in shuffled order, each drawing one of the item parts still free
  void matching(std::mt19937& mt, std::vector<std::pair<int, int>>& seq) const {
    seq.clear();
    std::vector<int> rows((size_t)T), left((size_t)T);
    std::iota(rows.begin(), rows.end(), 0);
    std::iota(left.begin(), left.end(), 0);
    std::shuffle(rows.begin(), rows.end(), mt);
    for (int r : rows) {
      std::uniform_int_distribution<int> dis(0, (int)left.size() - 1);
      const int at = dis(mt);
      seq.emplace_back(r, left[(size_t)at]);
      left.erase(left.begin() + at);          // the free parts stay in ascending order, as the reference re-collects them
    }
  }

  // CSR positions of one epoch's visits: T rounds x T blocks, a block = its users' rows restricted to its item part.  The
  // blocks' own lists do not change from epoch to epoch (only the matching does): built once (one pass over the matrix per user
  // part), an epoch is T x T copies -- the first cut re-filtered every row T times per epoch, 0.3 s at the ML-20M shape.
  mutable std::vector<std::vector<size_t>> blockList;   // [user part * T + item part]
  void buildBlocks(const csr_t* trainMat) const {
    blockList.assign((size_t)T * T, std::vector<size_t>());
    for (int p = 0; p < T; p++)
      for (int u : users[(size_t)p])
        for (int64_t e = trainMat->rowptr[u]; e < trainMat->rowptr[u + 1]; e++) {
          const int ip = itemPart[(size_t)trainMat->rowind[e]];
          if (ip >= 0) blockList[(size_t)p * T + ip].push_back((size_t)e);
        }
  }
  void epochList(std::mt19937& mt, const csr_t* trainMat, std::vector<size_t>& out) const {
    if (blockList.size() != (size_t)T * T) buildBlocks(trainMat);
    out.clear();
    std::vector<std::pair<int, int>> seq;
    for (int k = 0; k < T; k++) {
      matching(mt, seq);
      for (const auto& blk : seq) {
        const std::vector<size_t>& l = blockList[(size_t)blk.first * T + blk.second];
        out.insert(out.end(), l.begin(), l.end());       // (filled by an OpenMP loop instead: slower, 96 -> 159 ms per iteration)
      }
    }
  }
};
}  // namespace

namespace {
// One host thread that makes the epoch orders of an exact trainer, for the whole loop; the list it shuffles is first touched (and
// so placed) by it.  (Inside the loop a shuffle takes 51 - 59 ms where it takes 41 - 48 alone; a thread per epoch or this one, the
// list touched here or by the caller: the same -- not the placement, then.)
class OrderThread {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool busy = false, quit = false;
  std::exception_ptr err;

 public:
  OrderThread() {
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        cv.wait(lk, [this] { return quit || (busy && job); });
        if (quit) return;
        std::function<void()> j = std::move(job);
        job = nullptr;
        lk.unlock();
        try { j(); } catch (...) { err = std::current_exception(); }
        lk.lock();
        busy = false;
        cv.notify_all();
      }
    });
  }
  ~OrderThread() {
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return !busy; }); quit = true; }
    cv.notify_all();
    th.join();
  }
  bool pending() { std::lock_guard<std::mutex> lk(mu); return busy; }
  void start(std::function<void()> j) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return !busy; });
    job = std::move(j);
    busy = true;
    started = true;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return !busy; });
    started = false;
    if (err) { std::exception_ptr e = err; err = nullptr; std::rethrow_exception(e); }
  }
  bool started = false;      // a job was started and not yet waited for
};
}  // namespace

void ModelMF::train(const Data& d, Model& b, IntSet& iu, IntSet& ii) { run(K_SGD, "train", d, b, iu, ii); }
void ModelMF::hogTrain(const Data& d, Model& b, IntSet& iu, IntSet& ii) { run(K_HOG, "hogTrain", d, b, iu, ii); }
void ModelMF::trainSGDPar(const Data& d, Model& b, IntSet& iu, IntSet& ii) { run(K_SGDPAR, "trainSGDPar", d, b, iu, ii); }
void ModelMF::trainUShuffle(const Data& d, Model& b, IntSet& iu, IntSet& ii) { run(K_SGDU, "trainUShuffle", d, b, iu, ii); }
void ModelMF::trainALS(const Data& d, Model& b, IntSet& iu, IntSet& ii) { run(K_ALS, "trainALS", d, b, iu, ii); }
void ModelMF::trainCCDPP(const Data& d, Model& b, IntSet& iu, IntSet& ii) { run(K_CCDPP, "trainCCDPP", d, b, iu, ii); }
void ModelMF::trainCCDPPFreqAdap(const Data& d, Model& b, IntSet& iu, IntSet& ii) {
  run(K_CCDPP_FA, "trainCCDPPFreqAdap", d, b, iu, ii);
}
void ModelMF::trainCCD(const Data& d, Model& b, IntSet& iu, IntSet& ii) { run(K_CCD, "trainCCD", d, b, iu, ii); }
void ModelMF::trainSGDParSVD(const Data& d, Model& b, IntSet& iu, IntSet& ii) { run(K_SGDPARSVD, "trainSGDParSVD", d, b, iu, ii); }

void ModelMF::run(Kind kind, const char* name, const Data& data, Model& bestModel, IntSet& invalidUsers,
                  IntSet& invalidItems) {
  std::cout << "\nModelMF::" << name << " trainSeed: " << trainSeed;
  // MFX_EXACT: replay the reference's own visiting order bit by bit -- 1: dataflow-scheduled (MFX_SGD_LEVELS, parallel),
  // 2: one lane group in list order (MFX_SGD_SERIAL, the slow statement of the same thing)
  //            0: never -- the lock-free tiled schedule (a different visiting order with thousands of ratings in flight: its
  //               test RMSE agrees with the reference's to ~1e-2, in either direction -- tests/test_parity_spread_gpu.py)
  // unset: every trainer keeps ITS OWN semantics.
  //   * ModelMF::train, trainUShuffle, trainSGDPar are sequential / deterministic in the reference (modelMF.cpp:83-105, 637-659,
  //     273-304): they replay the reference's order -- the numbers ARE the reference's -- up to MFX_EXACT_SEQ_BELOW train ratings
  //     (default 128 M -- the Netflix shape, 100 M ratings at rank 128, replays in 114 ms per epoch, bench.py secondary[C4]; the ML-20M shape replays at 1.38 G updates/s, 18 ms per call, next to the reference's own std::shuffle
  //     of the index list on the host -- 0.37 s as the library call, 0.05 s as mfhShuffle above, a thread ahead); larger matrices take
  //     the lock-free tiled schedule unless MFX_EXACT=1;
  //   * hogTrain is lock-free in the reference (:1747-1763) and the sibling models' loops sit in OpenMP-parallel block loops: they
  //     take the lock-free tiled schedule, except on small matrices (up to MFX_EXACT_BELOW = 2 M ratings), where >= 64 ratings of
  //     a workgroup in flight collide on the few hundred rows of a tile and the one-thread replay is both exact and cheap.
  const char* exactEnv = getenv("MFX_EXACT");
  const bool plainSgd = kind == K_SGD || kind == K_HOG || kind == K_SGDU || kind == K_SGDPAR || kind == K_IFW || kind == K_TMF ||
                        kind == K_TMFD || kind == K_SGDPARSVD;       // every SGD trainer, the sibling models included
  const bool sequentialSgd = kind == K_SGD || kind == K_SGDU || kind == K_SGDPAR;
  const char* belowEnv = getenv("MFX_EXACT_BELOW");
  const char* seqBelowEnv = getenv("MFX_EXACT_SEQ_BELOW");
  const int64_t exactBelow = belowEnv ? atoll(belowEnv) : 2000000;
  const int64_t exactSeqBelow = seqBelowEnv ? atoll(seqBelowEnv) : (belowEnv ? std::max<int64_t>(exactBelow, 0) : 128000000);
  const bool exact = exactEnv ? atoi(exactEnv) != 0
                              : (plainSgd && data.trainMat->nnz() <= (sequentialSgd ? exactSeqBelow : exactBelow));
  const int replayMode = (exactEnv && atoi(exactEnv) == 2) ? MFX_SGD_SERIAL : MFX_SGD_LEVELS;
  if (plainSgd) std::cout << " [" << (exact ? (replayMode == MFX_SGD_SERIAL ? "order replay, serial" : "order replay, dataflow schedule") : "lock-free tiled schedule") << "]";
  if (sequentialSgd && !exact && !exactEnv)
    std::cout << "\n[mfx] " << name << ": " << data.trainMat->nnz() << " train ratings exceed MFX_EXACT_SEQ_BELOW = " << exactSeqBelow
              << ": this run takes the lock-free tiled schedule (hogTrain's semantics), not the reference's sequential order; "
                 "MFX_EXACT=1 replays the order at any size";
  const csr_t* trainMat = data.trainMat;

  // bestModel starts as its own initialisation (main.cpp:1326-1327); it becomes the BEST snapshot
  attach(data);
  bestModel.dev = dev;
  bestModel.devSnap = MFX_SNAP_BEST;
  if (bestModel.uFac.rows == nUsers && bestModel.iFac.rows == nItems && bestModel.uFac.cols == facDim) {
    dev->check(mfx_set_factors(dev->ctx, bestModel.uFac.data(), bestModel.iFac.data(), MFX_ROWMAJOR), "set best");
    dev->check(mfx_snapshot_best(dev->ctx), "snapshot best");
    pushToDevice();
  } else {
    dev->check(mfx_snapshot_best(dev->ctx), "snapshot best");
  }
  bestModel.hostStale = true;

  std::cout << "\nObj b4 svd: " << objective(data) << " Train RMSE: " << RMSE(data.trainMat)
            << " Train nnz: " << data.trainNNZ << std::endl;
  if (kind == K_SGDPARSVD) {
    // singularVals = svdFrmSvdlibCSREig(data.trainMat, facDim, uFac, iFac, false) (modelMF.cpp:368): on the device
    const auto t0 = std::chrono::system_clock::now();
    singularVals.assign((size_t)facDim, 0.0f);
    const char* it = getenv("MFX_SVD_ITERS");
    dev->check(mfx_svd_init(dev->ctx, it ? atoi(it) : 10, std::max(10, facDim / 8), (uint32_t)trainSeed, singularVals.data()),
               "mfx_svd_init");
    hostStale = true;
    std::cout << "Rank regs: ";
    std::vector<float> regk((size_t)facDim);
    for (int k = 0; k < facDim; k++) {
      regk[k] = (sing_a + 1) / (sing_b + singularVals[k]);       // float, as the update evaluates it (:498)
      std::cout << (1.0 + sing_a) / (sing_b + singularVals[k]) << " ";
    }
    std::cout << std::endl;
    std::cout << "\nsvd duration: " << std::chrono::duration<double>(std::chrono::system_clock::now() - t0).count();
    dev->check(mfx_sgd_set_dim_reg(dev->ctx, regk.data()), "mfx_sgd_set_dim_reg");
  }

  int iter, bestIter = -1;
  double bestObj, prevObj, bestValRMSE, prevValRMSE;
  deviceInvalid(data, invalidUsers, invalidItems);   // getInvalidUsersItems + modelMF.cpp:40-45
  beforeLoop(kind, data, invalidUsers, invalidItems);
  prevObj = objective(data, invalidUsers, invalidItems);
  bestObj = prevObj;
  bestValRMSE = prevValRMSE = RMSE(data.valMat, invalidUsers, invalidItems);
  std::cout << "\nObj aftr svd: " << prevObj << " Train RMSE: " << RMSE(data.trainMat, invalidUsers, invalidItems)
            << " Val RMSE: " << bestValRMSE;
  std::cout << "\nModelMF::" << name << " trainSeed: " << trainSeed << " invalidUsers: " << invalidUsers.size()
            << " invalidItems: " << invalidItems.size() << std::endl;

  std::mt19937 mt(trainSeed);
  // every train rating belongs to a valid user and a valid item, so getUIRatings (util.cpp:722-747)
  // is the CSR-order rating list; uiRatingInds indexes it (modelMF.cpp:65-68)
  const int64_t nRatings = trainMat->nnz();
  std::vector<size_t> uiRatingInds;
  std::vector<uint64_t> userPerm;
  std::vector<size_t> validUsers;
  // (the reference's uiRatingInds is a vector<size_t>; below 2^32 ratings the same list is kept in 32 bits -- the same swaps, half the
  //  bytes for the host's shuffle and for the upload -- unless MFX_ORDER64=1)
  std::vector<uint32_t> ratingInds32;
  const bool order32 = nRatings < ((int64_t)1 << 32) && !(getenv("MFX_ORDER64") && atoi(getenv("MFX_ORDER64")) != 0);
  // (the list of K_SGD / K_HOG / K_IFW is allocated and numbered by the thread that shuffles it: orderThread, below)
  const bool listOfAll = (kind == K_SGD || kind == K_HOG || kind == K_IFW) && exact;
  // (MFX_DEVICE_SHUFFLE=0: the swaps of the epoch's std::shuffle on the host, as in round 3)
  std::vector<uint32_t> swapPos[2];      // the positions of this epoch's and of the next epoch's swaps
  int swapMake = 0, swapMade = 0;
  bool devShuffleOk = true;
  bool devShuffle = false;
  if (listOfAll && order32 && nRatings >= ((int64_t)1 << 20) && !(getenv("MFX_DEVICE_SHUFFLE") && atoi(getenv("MFX_DEVICE_SHUFFLE")) == 0) &&
      replayMode == MFX_SGD_LEVELS) {
    std::mt19937 probe(1);
    std::vector<uint32_t> pp;
    devShuffle = mfhShufflePositions(pp, (size_t)1 << 20, probe);     // (the self-checks of the restated generator pass on this library)
  }
  Strata strata;
  if (kind == K_SGDPAR && exact) {
    // modelMF.cpp:191-265: valid users / items shuffled with mt and dealt into T = omp_get_max_threads() parts
    const char* e = getenv("MFX_SGDPAR_PARTS");
    strata.deal(mt, trainMat, invalidUsers, invalidItems, e ? std::max(1, atoi(e)) : omp_get_max_threads());
  }
  if (kind == K_SGDU) {
    for (int u = 0; u < nUsers; u++)
      if (!invalidUsers.count(u)) validUsers.push_back(u);
    std::cout << "No. of valid users: " << validUsers.size() << std::endl;
  }
  std::cout << "\nTrain NNZ after removing invalid users and items: " << nRatings << std::endl;
  std::vector<int> dims(facDim);
  std::iota(dims.begin(), dims.end(), 0);
  if (kind == K_CCDPP || kind == K_CCDPP_FA) dev->check(mfx_ccdpp_begin(dev->ctx), "mfx_ccdpp_begin");
  // trainCCD: res = gk_csr_Dup(trainMat), uFac = 0 (modelMF.cpp:1509-1522)
  std::vector<uint16_t> uOrder, iOrder;
  if (kind == K_CCD) dev->check(mfx_ccd_begin(dev->ctx), "mfx_ccd_begin");

  mfx_sgd_opts o = mfx_sgd_opts();
  o.uReg = uReg; o.iReg = iReg; o.seed = (uint32_t)trainSeed; o.blocks = 0; o.own = 0; o.first = 0; o.count = 0;
  // Arithmetic of the bracket: the reference's own per trainer when its order is replayed (MFX_EXACT), hogTrain's
  // float row expressions otherwise -- on the lock-free schedule the 1e-7 relative difference between the double
  // and the float bracket is far below the effect of the visiting order, and the float one runs 1.5x faster.
  o.arith = kind == K_HOG ? MFX_ARITH_F32 : kind == K_SGDPAR ? MFX_ARITH_REF64F : MFX_ARITH_REF64;
  if (!exact && (kind == K_SGD || kind == K_SGDPAR)) o.arith = MFX_ARITH_F32;

  // Lock-free tiled schedule, first epochs (MFX_TILED_WARM = "<n>" | "hog:<n>" | "flow:<n>", n epochs; default below).  While the
  // factors are still at their +-0.01 initialisation the tiled schedule's block order is what makes it diverge where the
  // reference's shuffled loops train: an item's ratings of a tile are visited as one burst, every user of the burst is fresh
  // (p ~ 0, so the error stays at r however large q has grown) and each of them pushes q by (2 lr e)^2 q along itself
  // (tests/test_block_order.py shows the reference's own trainSGDPar order doing it).  In a uniformly shuffled
  // list the same item meets users that other items have already moved.  So the epochs that start from the initialisation run
  // in a uniformly shuffled device order -- lock-free over the whole list like hogTrain (modelMF.cpp:1747-1763: "hog"), or as
  // the sequential replay of that list ("flow") -- and the tiles take over from there; a NaN rollback to the initial model
  // (model.cpp:1486-1498) starts the count again.
  // Default flow:1 -- measured on the 2.4 M-rating fixture at the reference's rate 0.005 (scripts/warm_epochs.py, three seeds): tiles
  // from epoch 0: NaN, rollback, half the rate for the rest of the run, test RMSE 0.6141 (hogTrain's own: 0.62281); hog:1 0.6174
  // (the flat lock-free epoch on this chip loses most updates of the popular rows: a gentler start, a better model, but not the
  // reference's); flow:1 0.6225 ... 0.6233 (with the four tilings of sgd_slots.h; 0.6241 ... 0.6252 on one static tiling).
  int warmN = 1, warmMode = MFX_SGD_LEVELS;
  if (const char* e = getenv("MFX_TILED_WARM")) {
    const char* c = strchr(e, ':');
    if (c && !strncmp(e, "hog", 3)) warmMode = MFX_SGD_HOGWILD;
    warmN = std::max(0, atoi(c ? c + 1 : e));
  }
  if (kind == K_SGDPARSVD) warmN = 0;    // starts from the singular vectors, not from +-0.01
  int freshEpochs = 0;                   // epochs run since the factors were last at their initialisation
  auto tiledOrWarm = [&](mfx_sgd_opts& so) {
    so.order = MFX_ORDER_DEVICE;
    so.mode = freshEpochs < warmN ? warmMode : MFX_SGD_TILED;
  };

  double subIterDuration = 0;
  // Exact replay: the order of an epoch depends on the generator alone (nothing else in this loop draws from mt), so the order of
  // epoch e + 1 is made on a second host thread (orderThread) while the device replays epoch e and the objective is taken -- for train the
  // shuffle is the larger of the two.  orderOfEpoch(make) leaves this epoch's order in uiRatingInds (made ahead, or now);
  // orderAhead(), called once mfx_sgd_set_order has copied it, starts the next one.  An iteration that ends the loop leaves one
  // order unused; mt is local to this function.  MFX_NO_SHUFFLE_AHEAD=1: everything on the calling thread.
  double tOrder = 0, tMake = 0;         // (declared in front of nextOrder: the thread it joins on destruction writes tMake)
  OrderThread orderThread;               // (waits for its job on destruction)
  std::function<void()> makeOrder;
  if (listOfAll) {
    orderThread.start([&] {
      if (order32) { ratingInds32.resize((size_t)nRatings); std::iota(ratingInds32.begin(), ratingInds32.end(), 0u); }
      else { uiRatingInds.resize((size_t)nRatings); std::iota(uiRatingInds.begin(), uiRatingInds.end(), (size_t)0); }
    });
    orderThread.wait();
  }
  // MFX_TIME_LOOP=1: where an iteration of an exact trainer goes (stderr): waiting for the order, its upload, the rest
  static const bool timeLoop = getenv("MFX_TIME_LOOP") && atoi(getenv("MFX_TIME_LOOP")) != 0;
  auto orderOfEpoch = [&](std::function<void()> make) {
    const auto t0 = std::chrono::steady_clock::now();
    makeOrder = [make, &tMake] {
      const auto m0 = std::chrono::steady_clock::now();
      make();
      tMake = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - m0).count();
    };
    if (orderThread.started) orderThread.wait();
    else makeOrder();
    tOrder = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (timeLoop) fprintf(stderr, "[mfh] iteration %d: waited %.1f ms for the order (made in %.1f ms)\n", iter, tOrder, tMake);
  };
  auto orderAhead = [&] {
    if (iter + 1 < maxIter && !getenv("MFX_NO_SHUFFLE_AHEAD")) orderThread.start(makeOrder);
  };
  const auto loopStart = std::chrono::steady_clock::now();
  for (iter = 0; iter < maxIter; iter++) {
    auto start = std::chrono::system_clock::now();
    o.learnRate = learnRate;
    o.epoch = iter;
    switch (kind) {
      case K_SGD:
      case K_HOG:
      case K_IFW:     // ModelInvPopMF::train: the same sequential loop with the weights beforeLoop() installed
        if (exact) {
          // modelMF.cpp:76-81: std::shuffle every epoch on one thread (parBlockShuffle with one
          // OpenMP thread is a plain std::shuffle; the reference is only reproducible that way)
          // Long lists (round 4): the host draws only the POSITIONS of the shuffle -- the generator's stream, a thread ahead -- and the
          // device applies the swaps to the list it keeps from epoch to epoch (mfx_sgd_apply_swaps32: the same list, bit for bit);
          // the random swaps were the slowest stage of mfhShuffle (45 ms of a 53 ms iteration at the ML-20M shape).
          if (order32 && devShuffle)
            orderOfEpoch([&swapPos, &swapMake, &swapMade, &mt, nRatings, &devShuffleOk] {
              devShuffleOk = mfhShufflePositions(swapPos[swapMake], (size_t)nRatings, mt);
              swapMade = swapMake;
              swapMake ^= 1;
            });
          else if (order32) orderOfEpoch([&ratingInds32, &mt] { mfhShuffle(ratingInds32, mt); });
          else orderOfEpoch([&uiRatingInds, &mt] { mfhShuffle(uiRatingInds, mt); });
          const auto u0 = std::chrono::steady_clock::now();
          if (order32 && devShuffle) {
            if (!devShuffleOk) throw MfxError(-100, "ModelMF::train: the positions form of the shuffle stopped applying in the middle of a run");
            const int cur = swapMade;
            orderAhead();                     // the next epoch's positions go into the OTHER buffer while the device applies these
            if (iter == 0) dev->check(mfx_sgd_set_order32(dev->ctx, ratingInds32.data(), nRatings), "set_order32");   // 0 .. n-1, once
            dev->check(mfx_sgd_apply_swaps32(dev->ctx, swapPos[cur].data(), nRatings), "apply_swaps32");
          } else if (order32) dev->check(mfx_sgd_set_order32(dev->ctx, ratingInds32.data(), nRatings), "set_order32");   // (copied when it returns)
          else dev->check(mfx_sgd_set_order(dev->ctx, (const uint64_t*)uiRatingInds.data(), nRatings), "set_order");
          const auto u1 = std::chrono::steady_clock::now();
          o.mode = replayMode; o.order = MFX_ORDER_HOST;
          if (!(order32 && devShuffle)) orderAhead();
          dev->check(mfx_sgd_epoch(dev->ctx, &o), "mfx_sgd_epoch");
          if (timeLoop)
            fprintf(stderr, "[mfh] iteration %d: order upload %.1f ms, replay call %.1f ms\n", iter,
                    std::chrono::duration<double, std::milli>(u1 - u0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - u1).count());
          break;
        } else {
          tiledOrWarm(o);                                        // (K_IFW: the tiled kernel's weighted variant)
        }
        dev->check(mfx_sgd_epoch(dev->ctx, &o), "mfx_sgd_epoch");
        break;
      case K_SGDPAR:
        if (exact) {
          // modelMF.cpp:273-304: T rounds, each a fresh random matching user part -> item part; the blocks of a round
          // share no rows, so their sequential sweeps commute and ONE list (round by round, block by block) replayed in
          // order is the reference's parallel-for.  float diff, double bracket (:289-299).
          orderOfEpoch([&strata, &mt, trainMat, &uiRatingInds] { strata.epochList(mt, trainMat, uiRatingInds); });
          if (!uiRatingInds.empty()) {
            dev->check(mfx_sgd_set_order(dev->ctx, (const uint64_t*)uiRatingInds.data(), (int64_t)uiRatingInds.size()), "set_order");
            orderAhead();
            o.mode = replayMode; o.order = MFX_ORDER_HOST;
            dev->check(mfx_sgd_epoch(dev->ctx, &o), "mfx_sgd_epoch");
          } else {
            orderAhead();
          }
        } else {
          // default: the stratification mapped onto the chip's own strata -- user-block x item-block tiles, one L2 domain per tile
          tiledOrWarm(o);
          dev->check(mfx_sgd_epoch(dev->ctx, &o), "mfx_sgd_epoch");
        }
        break;
      case K_TMFD:        // modelPoissonDropout.cpp:170-224: K_TMF with the draws mfx_set_tmf_dropout installed
      case K_TMF:         // modelDropoutSigmoid.cpp:140-192 with the rank table beforeLoop() installed; float diff
        o.arith = MFX_ARITH_REF64F;
        if (exact) { o.mode = replayMode; o.order = MFX_ORDER_NATURAL; }
        else tiledOrWarm(o);                                    // the tiled kernel's truncated-rank variant
        dev->check(mfx_sgd_epoch(dev->ctx, &o), "mfx_sgd_epoch");
        break;
      case K_SGDPARSVD:   // modelMF.cpp:474-512 with the per-dimension regulariser set above; lock-free, coherent rows
        o.mode = exact ? replayMode : MFX_SGD_TILED;           // the tiled kernel's per-dimension-regulariser variant
        o.order = exact ? MFX_ORDER_NATURAL : MFX_ORDER_DEVICE;
        dev->check(mfx_sgd_epoch(dev->ctx, &o), "mfx_sgd_epoch");
        break;
      case K_SGDU: {
        if (exact) {
          orderOfEpoch([&validUsers, &mt, trainMat, &uiRatingInds] {
            std::shuffle(validUsers.begin(), validUsers.end(), mt);   // modelMF.cpp:635
            uiRatingInds.clear();
            for (size_t u : validUsers)
              for (int64_t e = trainMat->rowptr[u]; e < trainMat->rowptr[u + 1]; e++) uiRatingInds.push_back((size_t)e);
          });
          dev->check(mfx_sgd_set_order(dev->ctx, (const uint64_t*)uiRatingInds.data(), (int64_t)uiRatingInds.size()),
                     "set_order");
          orderAhead();
          o.mode = replayMode; o.order = MFX_ORDER_HOST;
        } else if (getenv("MFX_SGDU_USERS_KERNEL")) {
          // one group per user, the user row kept in registers over the user's ratings (closest to the reference's
          // order; 12x slower than the tiled schedule at the ML-20M shape)
          std::shuffle(validUsers.begin(), validUsers.end(), mt);   // modelMF.cpp:635
          dev->check(mfx_sgd_set_order(dev->ctx, (const uint64_t*)validUsers.data(), (int64_t)validUsers.size()),
                     "set_order");
          o.mode = MFX_SGD_USERS; o.order = MFX_ORDER_HOST;
        } else {
          // like train / hogTrain: the lock-free tiled schedule (the order of the users is then the kernel's business)
          std::shuffle(validUsers.begin(), validUsers.end(), mt);   // modelMF.cpp:635 (keeps mt where the reference has it)
          tiledOrWarm(o); o.arith = MFX_ARITH_F32;
        }
        dev->check(mfx_sgd_epoch(dev->ctx, &o), "mfx_sgd_epoch");
        break;
      }
      case K_ALS:      // modelMF.cpp:805-880
        dev->check(mfx_als_half_sweep(dev->ctx, MFX_SIDE_USERS, uReg), "mfx_als_half_sweep");
        dev->check(mfx_als_half_sweep(dev->ctx, MFX_SIDE_ITEMS, iReg), "mfx_als_half_sweep");
        break;
      case K_CCDPP:
      case K_CCDPP_FA:
        if (kind == K_CCDPP) std::shuffle(dims.begin(), dims.end(), mt);   // :1026 (not in FreqAdap, :1271)
        for (int k : dims)
          dev->check(mfx_ccdpp_rank1(dev->ctx, k, 5, uReg, iReg, iter > 0, kind == K_CCDPP_FA ? 75.0f : -1.0f),
                     "mfx_ccdpp_rank1");
        break;
      case K_CCD: {
        const uint16_t *uo = nullptr, *io = nullptr;
        if (exact) {
          // the reference draws every row's factor order from the shared mt (modelMF.cpp:1539-1540,
          // :1577-1578); on one thread that is users in order, then items in order
          uOrder.assign((size_t)trainMat->nrows * facDim, 0);
          iOrder.assign((size_t)trainMat->ncols * facDim, 0);
          auto draw = [&](std::vector<uint16_t>& out, int row) {
            std::vector<int> udims(dims);
            std::shuffle(udims.begin(), udims.end(), mt);
            for (int s = 0; s < facDim; s++) out[(size_t)row * facDim + s] = (uint16_t)udims[s];
          };
          for (int u = 0; u < nUsers; u++)
            if (!invalidUsers.count(u)) draw(uOrder, u);
          for (int item = 0; item < nItems; item++)
            if (!invalidItems.count(item) && item < trainMat->ncols) draw(iOrder, item);
          uo = uOrder.data(); io = iOrder.data();
        }
        dev->check(mfx_ccd_sweep(dev->ctx, MFX_SIDE_USERS, uReg, uo, (uint32_t)trainSeed, iter), "mfx_ccd_sweep");
        dev->check(mfx_ccd_sweep(dev->ctx, MFX_SIDE_ITEMS, iReg, io, (uint32_t)trainSeed, iter), "mfx_ccd_sweep");
        break;
      }
    }
    hostStale = true;
    // subIterDuration is only printed every MF_DISP_ITER iterations (modelMF.cpp:116-123): wait for the device just for
    // those; otherwise the evaluation below is the first thing that needs the epoch to have finished
    if (iter % MF_DISP_ITER == 0) {
      dev->check(mfx_synchronize(dev->ctx), "mfx_synchronize");
      subIterDuration = std::chrono::duration<double>(std::chrono::system_clock::now() - start).count();
    }

    if (iter % MF_OBJ_ITER == 0 || iter == maxIter - 1) {
      if (kind == K_SGDPARSVD ? isTerminateModelSing(bestModel, data, iter, bestIter, bestObj, prevObj, bestValRMSE,
                                                     prevValRMSE, invalidUsers, invalidItems)
                              : isTerminateModel(bestModel, data, iter, bestIter, bestObj, prevObj, bestValRMSE, prevValRMSE,
                                                 invalidUsers, invalidItems))
        break;
      freshEpochs = (rolledBack && bestIter < 0) ? 0 : freshEpochs + 1;
      if (iter % MF_DISP_ITER == 0) {
        std::cout << "ModelMF::" << name << " trainSeed: " << trainSeed << " Iter: " << iter
                  << " Objective: " << std::scientific << prevObj
                  << " Train RMSE: " << RMSE(data.trainMat, invalidUsers, invalidItems)
                  << " Val RMSE: " << prevValRMSE << " subIterDuration: " << subIterDuration << std::endl;
      }
      if ((iter % MF_SAVE_ITER == 0 || iter == maxIter - 1) && kind != K_SGDPAR)   // :335-338 commented out there
        bestModel.saveFacs(std::string(data.prefix));
    }
  }
  lastLoopSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - loopStart).count();
  lastIters = std::min(iter + 1, maxIter);
  if (kind != K_SGDPAR) bestModel.saveFacs(std::string(data.prefix));
  std::cout << "\nBest model validation RMSE: " << bestModel.RMSE(data.valMat, invalidUsers, invalidItems);
  if (kind == K_CCDPP || kind == K_CCDPP_FA) dev->check(mfx_ccdpp_end(dev->ctx), "mfx_ccdpp_end");
  if (kind == K_CCD) dev->check(mfx_ccd_end(dev->ctx), "mfx_ccd_end");
  if (kind == K_SGDPARSVD) dev->check(mfx_sgd_set_dim_reg(dev->ctx, nullptr), "mfx_sgd_set_dim_reg");
  afterLoop(kind);
  syncHost();
  bestModel.syncHost();
}
