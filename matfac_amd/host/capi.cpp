// capi.cpp -- C entry point that drives the host classes on in-memory matrices, so that the
// parity tests (Python) can run ModelMF::train* exactly as main() would, without text files.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>

#include "mf_model.h"
#include "model_invpop.h"
#include "model_tmf.h"
#include "mfhost.h"
#include "dataprep.h"
#include "model_bias.h"

static int mfh_train_impl(const char* method, int32_t nrows, const int64_t* tr_ptr, const int32_t* tr_ind,
                         const float* tr_val, int32_t tr_ncols, const int64_t* va_ptr, const int32_t* va_ind,
                         const float* va_val, int32_t va_ncols, const int64_t* te_ptr, const int32_t* te_ind,
                         const float* te_val, int32_t te_ncols, int32_t K, int32_t maxIter, int32_t seed,
                         float learnRate, float uReg, float iReg, const char* prefix, float* Ulast, float* Vlast,
                         float* Ubest, float* Vbest, double* stats, uint8_t* invU, uint8_t* invI) {
  std::string e, pfx = prefix ? prefix : "";
  if (!prefix) setenv("MFX_NO_SAVE", "1", 1);
  Params params(K, maxIter, K, seed, uReg, iReg, learnRate, 0.0f, 0.0f, e, e, e, e, e, e, e, e, pfx);
  Data data(csr_from_arrays(nrows, tr_ncols, tr_ptr, tr_ind, tr_val),
            csr_from_arrays(nrows, te_ncols, te_ptr, te_ind, te_val),
            csr_from_arrays(nrows, va_ncols, va_ptr, va_ind, va_val), pfx.c_str());
  params.nUsers = data.nUsers;
  params.nItems = data.nItems;
  const std::string m = method;
  std::unique_ptr<ModelMF> pm, pb;
  if (m.rfind("ifwmf:", 0) == 0) {       // "ifwmf:<rhoRMS>": ModelInvPopMF as main.cpp:1361-1366 builds it
    params.rhoRMS = (float)atof(m.c_str() + 6);
    std::vector<double> uf((size_t)data.trainMat->nrows, 0.0), itf((size_t)data.trainMat->ncols, 0.0);
    for (int u = 0; u < data.trainMat->nrows; u++)
      for (int64_t e2 = data.trainMat->rowptr[u]; e2 < data.trainMat->rowptr[u + 1]; e2++) { uf[u] += 1; itf[data.trainMat->rowind[e2]] += 1; }
    pm.reset(new ModelInvPopMF(params, params.seed, uf, itf));
    pb.reset(new ModelInvPopMF(params, params.seed, uf, itf));
  } else if (m.rfind("tmfd:", 0) == 0) {  // "tmfd:<rhoRMS>:<alpha>": ModelPoissonDropout (main.cpp:1355-1360)
    params.rhoRMS = (float)atof(m.c_str() + 5);
    const size_t c2 = m.find(':', 5);
    params.alpha = c2 == std::string::npos ? 0.0f : (float)atof(m.c_str() + c2 + 1);
    std::vector<double> uf((size_t)data.trainMat->nrows, 0.0), itf((size_t)data.trainMat->ncols, 0.0), none;
    for (int u = 0; u < data.trainMat->nrows; u++)
      for (int64_t e2 = data.trainMat->rowptr[u]; e2 < data.trainMat->rowptr[u + 1]; e2++) { uf[u] += 1; itf[data.trainMat->rowind[e2]] += 1; }
    pm.reset(new ModelPoissonDropout(params, params.seed, none, none, uf, itf));
    pb.reset(new ModelPoissonDropout(params, params.seed, none, none, uf, itf));
  } else if (m.rfind("tmf:", 0) == 0) {  // "tmf:<rhoRMS>:<alpha>": ModelDropoutSigmoid as main.cpp:1349-1354 builds it
    params.rhoRMS = (float)atof(m.c_str() + 4);
    const size_t c2 = m.find(':', 4);
    params.alpha = c2 == std::string::npos ? 0.0f : (float)atof(m.c_str() + c2 + 1);
    std::vector<double> uf((size_t)data.trainMat->nrows, 0.0), itf((size_t)data.trainMat->ncols, 0.0), none;
    for (int u = 0; u < data.trainMat->nrows; u++)
      for (int64_t e2 = data.trainMat->rowptr[u]; e2 < data.trainMat->rowptr[u + 1]; e2++) { uf[u] += 1; itf[data.trainMat->rowind[e2]] += 1; }
    pm.reset(new ModelDropoutSigmoid(params, params.seed, none, none, uf, itf));
    pb.reset(new ModelDropoutSigmoid(params, params.seed, none, none, uf, itf));
  } else {
    pm.reset(new ModelMF(params, params.seed));
    pb.reset(new ModelMF(params, params.seed));
  }
  ModelMF &model = *pm, &best = *pb;
  std::unordered_set<int> iu, ii;
  if (m.rfind("ifwmf:", 0) == 0 || m.rfind("tmf:", 0) == 0 || m.rfind("tmfd:", 0) == 0) model.train(data, best, iu, ii);
  else if (m == "ccd++") model.trainCCDPPFreqAdap(data, best, iu, ii);
  else if (m == "ccdpp") model.trainCCDPP(data, best, iu, ii);
  else if (m == "ccd") model.trainCCD(data, best, iu, ii);
  else if (m == "als") model.trainALS(data, best, iu, ii);
  else if (m == "hogsgd") model.hogTrain(data, best, iu, ii);
  else if (m == "sgdu") model.trainUShuffle(data, best, iu, ii);
  else if (m == "sgdpar") model.trainSGDPar(data, best, iu, ii);
  else if (m == "sgdparsvd") model.trainSGDParSVD(data, best, iu, ii);
  else if (m == "sgd") model.train(data, best, iu, ii);
  else return -1;
  const size_t su = sizeof(float) * (size_t)data.nUsers * K, si = sizeof(float) * (size_t)data.nItems * K;
  if (Ulast) memcpy(Ulast, model.uFac.data(), su);
  if (Vlast) memcpy(Vlast, model.iFac.data(), si);
  if (Ubest) memcpy(Ubest, best.uFac.data(), su);
  if (Vbest) memcpy(Vbest, best.iFac.data(), si);
  if (stats) {
    stats[0] = best.RMSE(data.trainMat, iu, ii);   // main.cpp:1377-1382
    stats[1] = best.RMSE(data.testMat, iu, ii);
    stats[2] = best.RMSE(data.valMat, iu, ii);
    stats[3] = model.learnRate;
    stats[4] = best.learnRate;
    stats[5] = data.nItems;
    stats[6] = model.lastLoopSeconds;
    stats[7] = model.lastIters;
  }
  if (invU) for (int u = 0; u < data.nUsers; u++) invU[u] = iu.count(u) ? 1 : 0;
  if (invI) for (int i = 0; i < data.nItems; i++) invI[i] = ii.count(i) ? 1 : 0;
  if (!prefix) unsetenv("MFX_NO_SAVE");
  return 0;
}

// no exception leaves the C entry points: an MfxError becomes its (negative) code with the message on stderr
extern "C" int mfh_train(const char* method, int32_t nrows, const int64_t* tr_ptr, const int32_t* tr_ind,
                         const float* tr_val, int32_t tr_ncols, const int64_t* va_ptr, const int32_t* va_ind,
                         const float* va_val, int32_t va_ncols, const int64_t* te_ptr, const int32_t* te_ind,
                         const float* te_val, int32_t te_ncols, int32_t K, int32_t maxIter, int32_t seed,
                         float learnRate, float uReg, float iReg, const char* prefix, float* Ulast, float* Vlast,
                         float* Ubest, float* Vbest, double* stats, uint8_t* invU, uint8_t* invI) {
  try {
    return mfh_train_impl(method, nrows, tr_ptr, tr_ind, tr_val, tr_ncols, va_ptr, va_ind, va_val, va_ncols, te_ptr, te_ind, te_val,
                          te_ncols, K, maxIter, seed, learnRate, uReg, iReg, prefix, Ulast, Vlast, Ubest, Vbest, stats, invU, invI);
  } catch (const MfxError& e) {
    fprintf(stderr, "\n%s\n", e.what());
    return e.code < 0 ? e.code : -2;
  }
}

// Text-CSR loader / writer of the host library (csr.cpp) for the CPU-side tests: two-call protocol like
// the oracle's reader (arrays may be NULL to query the sizes).  Returns 0, or -1 with the message in err.
extern "C" int mfh_csr_read_text(const char* path, int32_t* nrows, int32_t* ncols, int64_t* nnz, int64_t* rowptr,
                                 int32_t* rowind, float* rowval, int64_t* colptr, int32_t* colind, float* colval,
                                 char* err, int errcap) {
  std::string e;
  csr_t* m = csr_read_text(path, &e);
  if (!m) {
    if (err && errcap > 0) { strncpy(err, e.c_str(), errcap - 1); err[errcap - 1] = 0; }
    return -1;
  }
  *nrows = m->nrows; *ncols = m->ncols; *nnz = m->nnz();
  if (rowptr) {
    memcpy(rowptr, m->rowptr, sizeof(int64_t) * ((size_t)m->nrows + 1));
    memcpy(rowind, m->rowind, sizeof(int32_t) * (size_t)m->nnz());
    memcpy(rowval, m->rowval, sizeof(float) * (size_t)m->nnz());
  }
  if (colptr) {
    csr_create_col_index(m);
    memcpy(colptr, m->colptr, sizeof(int64_t) * ((size_t)m->ncols + 1));
    memcpy(colind, m->colind, sizeof(int32_t) * (size_t)m->nnz());
    memcpy(colval, m->colval, sizeof(float) * (size_t)m->nnz());
  }
  csr_free(&m);
  return 0;
}

extern "C" int mfh_data_shape(const char* train, const char* test, const char* val, int32_t* nUsers, int32_t* nItems,
                              int32_t* trainNNZ) {
  std::string e, tr = train, te = test, va = val, pfx = "x";
  Params params(1, 1, 1, 1, 0.f, 0.f, 0.f, 0.f, 0.f, tr, te, va, e, e, e, e, e, pfx);
  Data data(params);
  *nUsers = data.nUsers; *nItems = data.nItems; *trainNNZ = data.trainNNZ;
  return 0;
}

// factor-file round trip for the tests: bin != 0 -> writeMatBin/readMatBin, else writeMat/readMat
extern "C" int mfh_mat_write(const char* path, const float* data, int32_t n, int32_t k, int32_t bin) {
  DenseF32 m(n, k);
  memcpy(m.data(), data, sizeof(float) * (size_t)n * k);
  if (bin) writeMatBin(m, n, k, path); else writeMat(m, n, k, path);
  return 0;
}
extern "C" int mfh_mat_read(const char* path, float* data, int32_t n, int32_t k) {
  DenseF32 m;
  if (!readMat(m, n, k, path)) return -1;
  memcpy(data, m.data(), sizeof(float) * (size_t)n * k);
  return 0;
}

// the file-level data preparation (dataprep.cpp) for the tests: split an in-memory CSR into three files / write a
// synthetic matrix from given factors (row-major doubles)
extern "C" int mfh_write_train_test_val(int32_t nrows, int32_t ncols, const int64_t* rowptr, const int32_t* rowind, const float* rowval,
                                        const char* trainFile, const char* testFile, const char* valFile, float testPc, float valPc,
                                        int32_t seed) {
  csr_t* m = csr_from_arrays(nrows, ncols, rowptr, rowind, rowval);
  writeTrainTestValMat(m, trainFile, testFile, valFile, testPc, valPc, seed);
  csr_free(&m);
  return 0;
}
extern "C" int mfh_write_rand_mat_csr(const char* file, const double* U, const double* V, int32_t nUsers, int32_t nItems, int32_t facDim,
                                      int32_t seed, int32_t nnz) {
  std::vector<std::vector<double>> uFac((size_t)nUsers), iFac((size_t)nItems);
  for (int u = 0; u < nUsers; u++) uFac[(size_t)u].assign(U + (size_t)u * facDim, U + (size_t)(u + 1) * facDim);
  for (int i = 0; i < nItems; i++) iFac[(size_t)i].assign(V + (size_t)i * facDim, V + (size_t)(i + 1) * facDim);
  writeRandMatCSR(file, uFac, iFac, facDim, seed, nnz);
  return 0;
}

// ModelMFBias::train on in-memory matrices (the bias-only sibling): last and best bias vectors, stats as mfh_train
static int mfh_train_bias_impl(int32_t nrows, const int64_t* tr_ptr, const int32_t* tr_ind, const float* tr_val, int32_t tr_ncols,
                               const int64_t* va_ptr, const int32_t* va_ind, const float* va_val, int32_t va_ncols, const int64_t* te_ptr,
                               const int32_t* te_ind, const float* te_val, int32_t te_ncols, int32_t K, int32_t maxIter, int32_t seed,
                               float learnRate, float uReg, float iReg, float* ubLast, float* ibLast, float* ubBest, float* ibBest,
                               double* stats) {
  std::string e, pfx = "";
  Params params(K, maxIter, K, seed, uReg, iReg, learnRate, 0.0f, 0.0f, e, e, e, e, e, e, e, e, pfx);
  Data data(csr_from_arrays(nrows, tr_ncols, tr_ptr, tr_ind, tr_val), csr_from_arrays(nrows, te_ncols, te_ptr, te_ind, te_val),
            csr_from_arrays(nrows, va_ncols, va_ptr, va_ind, va_val), pfx.c_str());
  params.nUsers = data.nUsers;
  params.nItems = data.nItems;
  ModelMFBias model(params, params.seed), best(params, params.seed);
  std::unordered_set<int> iu, ii;
  model.train(data, best, iu, ii);
  if (ubLast) memcpy(ubLast, model.uBias.data(), sizeof(float) * (size_t)data.nUsers);
  if (ibLast) memcpy(ibLast, model.iBias.data(), sizeof(float) * (size_t)data.nItems);
  if (ubBest) memcpy(ubBest, best.uBias.data(), sizeof(float) * (size_t)data.nUsers);
  if (ibBest) memcpy(ibBest, best.iBias.data(), sizeof(float) * (size_t)data.nItems);
  if (stats) {
    stats[0] = best.RMSE(data.trainMat, iu, ii);
    stats[1] = best.RMSE(data.testMat, iu, ii);
    stats[2] = best.RMSE(data.valMat, iu, ii);
    stats[3] = model.learnRate;
    stats[4] = model.lastIters;
  }
  return 0;
}
// (the whole body behind one try: the RMSE calls of the statistics can throw MfxError too, and no exception may leave a C entry
// point; MFX_NO_SAVE is restored on every way out)
extern "C" int mfh_train_bias(int32_t nrows, const int64_t* tr_ptr, const int32_t* tr_ind, const float* tr_val, int32_t tr_ncols,
                              const int64_t* va_ptr, const int32_t* va_ind, const float* va_val, int32_t va_ncols, const int64_t* te_ptr,
                              const int32_t* te_ind, const float* te_val, int32_t te_ncols, int32_t K, int32_t maxIter, int32_t seed,
                              float learnRate, float uReg, float iReg, float* ubLast, float* ibLast, float* ubBest, float* ibBest,
                              double* stats) {
  const char* had = getenv("MFX_NO_SAVE");
  const std::string prev = had ? had : "";
  setenv("MFX_NO_SAVE", "1", 1);
  int rc;
  try {
    rc = mfh_train_bias_impl(nrows, tr_ptr, tr_ind, tr_val, tr_ncols, va_ptr, va_ind, va_val, va_ncols, te_ptr, te_ind, te_val, te_ncols, K,
                             maxIter, seed, learnRate, uReg, iReg, ubLast, ibLast, ubBest, ibBest, stats);
  } catch (const MfxError& e) {
    fprintf(stderr, "\n%s\n", e.what());
    rc = e.code < 0 ? e.code : -2;
  } catch (const std::exception& e) {
    fprintf(stderr, "\n%s\n", e.what());
    rc = -2;
  }
  if (had) setenv("MFX_NO_SAVE", prev.c_str(), 1); else unsetenv("MFX_NO_SAVE");
  return rc;
}


extern "C" int mfh_shuffle_check(int64_t n, uint32_t seed, double* secs) {
  try {
    std::vector<size_t> x((size_t)std::max<int64_t>(n, 0)), y;
    std::iota(x.begin(), x.end(), (size_t)0);
    y = x;
    std::mt19937 g1(seed), g2(seed);
    const auto t0 = std::chrono::steady_clock::now();
    std::shuffle(x.begin(), x.end(), g1);
    const auto t1 = std::chrono::steady_clock::now();
    mfhShuffle(y, g2);
    const auto t2 = std::chrono::steady_clock::now();
    if (secs) { secs[0] = std::chrono::duration<double>(t1 - t0).count(); secs[1] = std::chrono::duration<double>(t2 - t1).count(); }
    if (secs) secs[2] = (double)mfhShuffleForm();
    return x == y && g1 == g2 ? 1 : 0;
  } catch (...) {
    return -1;
  }
}

extern "C" int mfh_shuffle_positions_check(int64_t n, uint32_t seed, uint32_t* pos, double* secs) {
  try {
    if (n < 0 || n >= ((int64_t)1 << 32)) return -1;
    std::vector<size_t> x((size_t)n);
    std::iota(x.begin(), x.end(), (size_t)0);
    std::mt19937 g1(seed), g2(seed);
    g1.discard(seed % 700); g2.discard(seed % 700);
    std::vector<uint32_t> p;
    {
      std::mt19937 gw(seed + 1);                         // (a first call pays for the pages of p: the training loop reuses its two buffers)
      if (secs && !mfhShufflePositions(p, (size_t)n, gw)) return 2;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (!mfhShufflePositions(p, (size_t)n, g2)) return 2;
    if (secs) secs[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::shuffle(x.begin(), x.end(), g1);
    std::vector<uint32_t> y((size_t)n);
    std::iota(y.begin(), y.end(), 0u);
    for (size_t i = 1; i < (size_t)n; i++) {
      if (p[i] > i) return 0;
      std::swap(y[i], y[p[i]]);
    }
    if (pos) std::memcpy(pos, p.data(), sizeof(uint32_t) * (size_t)n);
    if (!(g1 == g2)) return 0;
    for (size_t k = 0; k < x.size(); k++)
      if (x[k] != (size_t)y[k]) return 0;
    return 1;
  } catch (...) {
    return -1;
  }
}

extern "C" int mfh_shuffle_check32(int64_t n, uint32_t seed, double* secs) {
  try {
    if (n < 0 || n >= ((int64_t)1 << 32)) return -1;
    std::vector<size_t> x((size_t)n);
    std::vector<uint32_t> y((size_t)n);
    std::iota(x.begin(), x.end(), (size_t)0);
    std::iota(y.begin(), y.end(), 0u);
    std::mt19937 g1(seed), g2(seed);
    const auto t0 = std::chrono::steady_clock::now();
    std::shuffle(x.begin(), x.end(), g1);
    const auto t1 = std::chrono::steady_clock::now();
    mfhShuffle(y, g2);
    const auto t2 = std::chrono::steady_clock::now();
    if (secs) { secs[0] = std::chrono::duration<double>(t1 - t0).count(); secs[1] = std::chrono::duration<double>(t2 - t1).count(); secs[2] = (double)mfhShuffleForm(); }
    if (!(g1 == g2)) return 0;
    for (size_t k = 0; k < x.size(); k++)
      if (x[k] != (size_t)y[k]) return 0;
    return 1;
  } catch (...) {
    return -1;
  }
}
