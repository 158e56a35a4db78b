// oracle.cpp -- CPU restatement of the matfac training hot path.
// TEST INFRASTRUCTURE ONLY (see oracle.h).  Compile with -ffp-contract=off so
// that every multiply/add below is the separate IEEE operation the reference's
// g++ -std=c++14 -O3 build (CMakeLists.txt:3, no -march, ISO mode) performs;
// fused operations are spelled fmaf() explicitly where the device order needs them.
#include "oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <numeric>
#include <random>
#include <sstream>
#include <string>
#include <unordered_set>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

// ---------------------------------------------------------------------------
// dot products
// ---------------------------------------------------------------------------
// The device rule for rank K (documented in include/mfx.h): L lanes per rating,
// each lane owns 4 consecutive floats per 4L-wide chunk, C chunks.
void orc_tree_shape(int K, int* L, int* C) {
  if (K <= 16) { *L = 4; *C = 1; }
  else if (K <= 32) { *L = 8; *C = 1; }
  else { *L = 16; *C = (K + 63) / 64; }
}

// model.cpp:547-549 estRating: uFac.row(u).dot(iFac.row(item)).  The rows of a
// column-major MatrixXf are strided, Eigen cannot packetise them and reduces
// coefficient by coefficient: res = a0*b0; res += ak*bk (separate mul / add).
static inline float dot_seq(const float* p, const float* q, int K) {
  float s = p[0] * q[0];
  for (int k = 1; k < K; k++) s = s + p[k] * q[k];
  return s;
}

// The order the HIP kernels use: lane j accumulates its elements with an fma
// chain starting from 0, then an xor butterfly over the L lanes (levels 1, 2, ..., L/2).
static inline float dot_tree(const float* p, const float* q, int K) {
  int L, C;
  orc_tree_shape(K, &L, &C);
  float s[16], t[16];
  for (int j = 0; j < L; j++) {
    float a = 0.0f;
    for (int c = 0; c < C; c++)
      for (int e = 0; e < 4; e++) {
        int k = c * 4 * L + 4 * j + e;
        if (k < K) a = fmaf(p[k], q[k], a);
      }
    s[j] = a;
  }
  for (int m = 1; m < L; m <<= 1) {
    for (int j = 0; j < L; j++) t[j] = s[j] + s[j ^ m];
    for (int j = 0; j < L; j++) s[j] = t[j];
  }
  return s[0];
}

static inline float dotf(const float* p, const float* q, int K, int mode) {
  return mode == ORC_DOT_TREE ? dot_tree(p, q, K) : dot_seq(p, q, K);
}
float orc_dot(const float* p, const float* q, int K, int dot_mode) {
  return dotf(p, q, K, dot_mode);
}

// ---------------------------------------------------------------------------
// RNG / shuffles
// ---------------------------------------------------------------------------
// model.cpp:2331-2362: default_random_engine(seed) (minstd_rand0 in libstdc++),
// uniform_real_distribution<double>((double)-0.01f,(double)0.01f); uFac row by
// row, then iFac (then uBias, iBias which the MF path never reads).
void orc_init_factors(int seed, int nU, int nI, int K, float* U, float* V) {
  std::default_random_engine generator(seed);
  float lb = -0.01, ub = 0.01;
  std::uniform_real_distribution<double> dist(lb, ub);
  for (int64_t u = 0; u < nU; u++)
    for (int k = 0; k < K; k++) U[u * K + k] = dist(generator);
  for (int64_t i = 0; i < nI; i++)
    for (int k = 0; k < K; k++) V[i * K + k] = dist(generator);
}

void* orc_mt_create(uint32_t seed) { return new std::mt19937(seed); }
void orc_mt_free(void* h) { delete (std::mt19937*)h; }
uint32_t orc_mt_next(void* h) { return (uint32_t)(*(std::mt19937*)h)(); }
void orc_mt_shuffle_u64(void* h, uint64_t* arr, int64_t n) {
  // modelMF.cpp:67-68,78: std::vector<size_t> + std::shuffle(.., mt)
  static_assert(sizeof(size_t) == sizeof(uint64_t), "size_t");
  std::shuffle((size_t*)arr, (size_t*)arr + n, *(std::mt19937*)h);
}
void orc_mt_shuffle_i32(void* h, int32_t* arr, int64_t n) {
  std::shuffle(arr, arr + n, *(std::mt19937*)h);  // modelMF.cpp:1026 (vector<int>)
}
// util.cpp:1047-1064.  The reference runs the per-thread shuffles concurrently on
// ONE shared mt19937 (a data race); here the blocks are shuffled in thread order,
// which is what it does with OMP_NUM_THREADS=1 and one admissible schedule otherwise.
void orc_mt_par_block_shuffle_u64(void* h, uint64_t* arr, int64_t n, int nthreads) {
  std::mt19937& mt = *(std::mt19937*)h;
  int arrSz = (int)n;
  for (int tID = 0; tID < nthreads; tID++) {
    int blockSz = arrSz / nthreads;
    size_t* start = (size_t*)arr + (int64_t)tID * blockSz;
    size_t* end = (size_t*)arr + (int64_t)(tID + 1) * blockSz;
    if ((tID + 1) * blockSz >= arrSz) end = (size_t*)arr + n;
    std::shuffle(start, end, mt);
  }
}
// util.cpp:1077-1107 sgdUpdateBlockSeq
static void block_seq(int dim, std::vector<std::pair<int, int>>& updateSeq,
                      std::mt19937& mt) {
  updateSeq.clear();
  std::vector<bool> colMask(dim, false);
  std::vector<int> rowInds(dim);
  std::iota(rowInds.begin(), rowInds.end(), 0);
  std::shuffle(rowInds.begin(), rowInds.end(), mt);
  for (int ind = 0; ind < dim; ind++) {
    int currRow = rowInds[ind];
    std::vector<int> leftCols;
    for (int k = 0; k < dim; k++)
      if (!colMask[k]) leftCols.push_back(k);
    std::uniform_int_distribution<int> dis(0, (int)leftCols.size() - 1);
    int currCol = leftCols[dis(mt)];
    updateSeq.push_back(std::make_pair(currRow, currCol));
    colMask[currCol] = true;
  }
}
void orc_mt_block_seq(void* h, int dim, int32_t* rows, int32_t* cols) {
  std::vector<std::pair<int, int>> seq;
  block_seq(dim, seq, *(std::mt19937*)h);
  for (int t = 0; t < dim; t++) { rows[t] = seq[t].first; cols[t] = seq[t].second; }
}

// ---------------------------------------------------------------------------
// CSR helpers
// ---------------------------------------------------------------------------
// gk_csr_CreateIndex(mat, GK_CSR_COL) [GKlib, not in tree]: counting sort of the
// row view into the column view; rows are visited in ascending order so the users
// of a column come out ascending (datastruct.cpp:18).
void orc_create_col_index(int32_t nrows, int32_t ncols, const int64_t* rowptr,
                          const int32_t* rowind, const float* rowval, int64_t* colptr,
                          int32_t* colind, float* colval) {
  for (int32_t j = 0; j <= ncols; j++) colptr[j] = 0;
  for (int64_t e = 0; e < rowptr[nrows]; e++) colptr[rowind[e] + 1]++;
  for (int32_t j = 0; j < ncols; j++) colptr[j + 1] += colptr[j];
  std::vector<int64_t> pos(colptr, colptr + ncols);
  for (int32_t u = 0; u < nrows; u++)
    for (int64_t e = rowptr[u]; e < rowptr[u + 1]; e++) {
      int64_t d = pos[rowind[e]]++;
      colind[d] = u;
      colval[d] = rowval[e];
    }
}

// util.cpp:511-544 getInvalidUsersItems (the ignore sets are empty on this path:
// genStats fills nothing into them that is read here) + modelMF.cpp:40-45.
void orc_invalid(int32_t nrows, int32_t ncols, const int64_t* rowptr,
                 const int32_t* rowind, int32_t nUsers, int32_t nItems, uint8_t* invU,
                 uint8_t* invI) {
  std::vector<int> uItemCount(nrows, 0), iUserCount(ncols, 0);
  for (int u = 0; u < nrows; u++)
    for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) {
      uItemCount[u] += 1;
      iUserCount[rowind[ii]] += 1;
    }
  for (int u = 0; u < nUsers; u++) invU[u] = (u >= nrows) || (uItemCount[u] == 0);
  for (int i = 0; i < nItems; i++) invI[i] = (i >= ncols) || (iUserCount[i] == 0);
}

// GKlib gk_csr_Read(GK_CSR_FMT_CSR, readvals=1, numbering=0) [not in tree]; layout
// confirmed by the in-tree writers python/convert_scipy_sparse_to_text_csr.py:19-26
// and io.cpp:769-783: one line per user, "item rating item rating ...", an empty
// line is a user without ratings, nrows = number of lines, ncols = max item + 1.
int orc_read_csr_text(const char* path, int32_t* nrows, int32_t* ncols, int64_t* nnz,
                      int64_t* rowptr, int32_t* rowind, float* rowval) {
  std::ifstream in(path);
  if (!in.is_open()) return -1;
  std::string line;
  int32_t r = 0, maxc = -1;
  int64_t e = 0;
  if (rowptr) rowptr[0] = 0;
  while (std::getline(in, line)) {
    if (!line.empty() && line[0] == '%') continue;
    const char* s = line.c_str();
    char* end;
    for (;;) {
      long c = strtol(s, &end, 10);
      if (end == s) break;
      s = end;
      float v = strtof(s, &end);
      if (end == s) return -2;  // odd token count
      s = end;
      if (rowind) { rowind[e] = (int32_t)c; rowval[e] = v; }
      if (c > maxc) maxc = (int32_t)c;
      e++;
    }
    r++;
    if (rowptr) rowptr[r] = e;
  }
  *nrows = r; *ncols = maxc + 1; *nnz = e;
  return 0;
}
int orc_write_csr_text(const char* path, int32_t nrows, const int64_t* rowptr,
                       const int32_t* rowind, const float* rowval) {
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  for (int32_t u = 0; u < nrows; u++) {
    for (int64_t e = rowptr[u]; e < rowptr[u + 1]; e++)
      fprintf(f, e + 1 < rowptr[u + 1] ? "%d %.9g " : "%d %.9g", rowind[e], rowval[e]);
    fputc('\n', f);
  }
  fclose(f);
  return 0;
}
// io.cpp:139-154 writeMat(Eigen::MatrixXf&): "v " per value, default ostream precision
int orc_write_mat(const char* path, const float* M, int nrows, int ncols) {
  std::ofstream op(path);
  if (!op.is_open()) return -1;
  for (int i = 0; i < nrows; i++) {
    for (int j = 0; j < ncols; j++) op << M[(int64_t)i * ncols + j] << " ";
    op << std::endl;
  }
  return 0;
}
// io.cpp:83-121 readMat(Eigen::MatrixXf&): split on ' ', std::stod, narrow to float
int orc_read_mat(const char* path, float* M, int nrows, int ncols) {
  std::ifstream in(path);
  if (!in.is_open()) return -1;
  std::string line, delimiter = " ";
  int i = 0;
  while (std::getline(in, line) && i < nrows) {
    int j = 0;
    size_t pos;
    while ((pos = line.find(delimiter)) != std::string::npos) {
      std::string token = line.substr(0, pos);
      if (j >= ncols) return -2;
      M[(int64_t)i * ncols + j++] = std::stod(token);
      line.erase(0, pos + delimiter.length());
    }
    if (line.length() > 0) {
      if (j >= ncols) return -2;
      M[(int64_t)i * ncols + j++] = std::stod(line);
    }
    if (j != ncols) return -2;
    i++;
  }
  return i == nrows ? 0 : -3;
}

// ---------------------------------------------------------------------------
// SGD
// ---------------------------------------------------------------------------
// One rating visit.  modelMF.cpp:91-103 (REF64), :288-299 (REF64F), :1755-1762 (F32).
static inline void sgd_update(float* p, float* q, float itemRat, int K, float learnRate,
                              float uReg, float iReg, int arith, int dot_mode) {
  if (arith == ORC_ARITH_F32) {
    // hogTrain: double r_ui_est = dot; const double diff = itemRat - r_ui_est;
    // uFac.row(u) -= learnRate*(-2.0*diff*iFac.row(item) + 2.0*uReg*uFac.row(u));
    // the double scalars meet float row expressions and are narrowed to float.
    double r_ui_est = dotf(p, q, K, dot_mode);
    const double diff = itemRat - r_ui_est;
    const float c1 = (float)(-2.0 * diff);
    const float cu = (float)(2.0 * uReg), ci = (float)(2.0 * iReg);
    for (int i = 0; i < K; i++) p[i] = p[i] - learnRate * (c1 * q[i] + cu * p[i]);
    for (int i = 0; i < K; i++) q[i] = q[i] - learnRate * (c1 * p[i] + ci * q[i]);
    return;
  }
  double diff;
  if (arith == ORC_ARITH_REF64F) {
    float r_ui_est = dotf(p, q, K, dot_mode);
    float d = itemRat - r_ui_est;
    diff = d;
  } else {
    double r_ui_est = dotf(p, q, K, dot_mode);
    diff = itemRat - r_ui_est;
  }
  for (int i = 0; i < K; i++)   // update user
    p[i] -= learnRate * (-2.0 * diff * q[i] + 2.0 * uReg * p[i]);
  for (int i = 0; i < K; i++)   // update item, sees the updated user row
    q[i] -= learnRate * (-2.0 * diff * p[i] + 2.0 * iReg * q[i]);
}

void orc_sgd_pass(int K, float* U, float* V, const int32_t* u, const int32_t* i,
                  const float* r, const uint64_t* order, int64_t n, float lr, float uReg,
                  float iReg, int arith, int dot_mode) {
  for (int64_t t = 0; t < n; t++) {
    int64_t ind = order ? (int64_t)order[t] : t;
    sgd_update(U + (int64_t)u[ind] * K, V + (int64_t)i[ind] * K, r[ind], K, lr, uReg, iReg,
               arith, dot_mode);
  }
}

// trainSGDParSVD's visit (modelMF.cpp:489-507): float dot, FLOAT diff, and the same per-dimension
// regulariser 2.0*((sing_a + 1)/(sing_b + singularVals[i])) on the user and on the item side.
// regk[i] = (sing_a + 1)/(sing_b + singularVals[i]) as the float the reference's expression yields.
void orc_sgd_pass_dimreg(int K, float* U, float* V, const int32_t* u, const int32_t* i, const float* r,
                         const uint64_t* order, int64_t n, float learnRate, const float* regk, int dot_mode) {
  for (int64_t t = 0; t < n; t++) {
    const int64_t ind = order ? (int64_t)order[t] : t;
    float* p = U + (int64_t)u[ind] * K;
    float* q = V + (int64_t)i[ind] * K;
    const float itemRat = r[ind];
    const float r_ui_est = dotf(p, q, K, dot_mode);
    const float diff = itemRat - r_ui_est;
    for (int k = 0; k < K; k++) p[k] -= learnRate * (-2.0 * diff * q[k] + 2.0 * regk[k] * p[k]);
    for (int k = 0; k < K; k++) q[k] -= learnRate * (-2.0 * diff * p[k] + 2.0 * regk[k] * q[k]);
  }
}

// Model::objectiveSing (model.cpp:1818-1865): squared error + sum_k x_k^2 * singularVals(k) over valid rows
double orc_objective_sing(int K, const float* U, const float* V, int32_t nUsers, int32_t nItems, int32_t nrows,
                          const int64_t* rowptr, const int32_t* rowind, const float* rowval, const uint8_t* invU,
                          const uint8_t* invI, const float* sing, int dot_mode, double* sse_out, double* ureg_out,
                          double* ireg_out) {
  double rmse = 0, uRegErr = 0, iRegErr = 0;
  for (int u = 0; u < nUsers; u++) {
    if (invU[u]) continue;
    const float* p = U + (int64_t)u * K;
    if (u < nrows)
      for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) {
        const int item = rowind[ii];
        if (invI[item]) continue;
        const float itemRat = rowval[ii];
        const double diff = itemRat - (double)dotf(p, V + (int64_t)item * K, K, dot_mode);
        rmse += diff * diff;
      }
    for (int k = 0; k < K; k++) uRegErr += p[k] * p[k] * sing[k];
  }
  for (int item = 0; item < nItems; item++) {
    if (invI[item]) continue;
    const float* q = V + (int64_t)item * K;
    for (int k = 0; k < K; k++) iRegErr += q[k] * q[k] * sing[k];
  }
  if (sse_out) *sse_out = rmse;
  if (ureg_out) *ureg_out = uRegErr;
  if (ireg_out) *ireg_out = iRegErr;
  return rmse + uRegErr + iRegErr;
}

// ---- ModelInvPopMF (IFWMF) ---------------------------------------------------------------
// modelInvPopMF.cpp:84-113: popularity scores of the valid users and items, each normalised to sum 1
void orc_ifw_pop(int32_t nrows, int32_t ncols, const int64_t* rowptr, const int32_t* rowind, const uint8_t* invU,
                 const uint8_t* invI, double* userFreq, double* itemFreq, double* invPopU, double* invPopI) {
  for (int u = 0; u < nrows; u++) userFreq[u] = 0;
  for (int i = 0; i < ncols; i++) itemFreq[i] = 0;
  for (int u = 0; u < nrows; u++)           // getRowColFreq, util.cpp:555-569
    for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) { userFreq[u] += 1; itemFreq[rowind[ii]] += 1; }
  int nTrainUsers = 0, nTrainItems = 0;
  for (int u = 0; u < nrows; u++) if (!invU[u]) nTrainUsers++;
  for (int i = 0; i < ncols; i++) if (!invI[i]) nTrainItems++;
  double sumPopScore = 0;
  for (int u = 0; u < nrows; u++) { invPopU[u] = 0; if (!invU[u]) { invPopU[u] = userFreq[u] / ((double)nTrainItems); sumPopScore += invPopU[u]; } }
  for (int u = 0; u < nrows; u++) if (!invU[u]) invPopU[u] = invPopU[u] / sumPopScore;
  sumPopScore = 0;
  for (int i = 0; i < ncols; i++) { invPopI[i] = 0; if (!invI[i]) { invPopI[i] = itemFreq[i] / ((double)nTrainUsers); sumPopScore += invPopI[i]; } }
  for (int i = 0; i < ncols; i++) if (!invI[i]) invPopI[i] = invPopI[i] / sumPopScore;
}
static inline float ifw_weight(int u, int item, const double* userFreq, const double* itemFreq, const double* invPopU,
                               const double* invPopI, float rhoRMS) {
  float wt = invPopI[item];                          // modelInvPopMF.cpp:161-166
  if (itemFreq[item] > userFreq[u]) wt = invPopU[u];
  wt = (1.0 / (1.0 + rhoRMS * wt));
  return wt;
}
// modelInvPopMF.cpp:152-178
void orc_sgd_pass_ifw(int K, float* U, float* V, const int32_t* u, const int32_t* i, const float* r, const uint64_t* order,
                      int64_t n, float learnRate, float uReg, float iReg, const double* userFreq, const double* itemFreq,
                      const double* invPopU, const double* invPopI, float rhoRMS, int dot_mode) {
  for (int64_t t = 0; t < n; t++) {
    const int64_t ind = order ? (int64_t)order[t] : t;
    float* p = U + (int64_t)u[ind] * K;
    float* q = V + (int64_t)i[ind] * K;
    const float itemRat = r[ind];
    const double r_ui_est = dotf(p, q, K, dot_mode);
    const double diff = itemRat - r_ui_est;
    const float wt = ifw_weight(u[ind], i[ind], userFreq, itemFreq, invPopU, invPopI, rhoRMS);
    for (int k = 0; k < K; k++) p[k] -= learnRate * (-2.0 * wt * diff * q[k] + 2.0 * uReg * p[k]);
    for (int k = 0; k < K; k++) q[k] -= learnRate * (-2.0 * wt * diff * p[k] + 2.0 * iReg * q[k]);
  }
}
// modelInvPopMF.cpp:3-55
double orc_objective_ifw(int K, const float* U, const float* V, int32_t nUsers, int32_t nItems, int32_t nrows,
                         const int64_t* rowptr, const int32_t* rowind, const float* rowval, const uint8_t* invU,
                         const uint8_t* invI, float uReg, float iReg, const double* userFreq, const double* itemFreq,
                         const double* invPopU, const double* invPopI, float rhoRMS, int dot_mode, double* wsse_out) {
  double rmse = 0, uRegErr = 0, iRegErr = 0;
  for (int u = 0; u < nUsers; u++) {
    if (invU[u]) continue;
    const float* p = U + (int64_t)u * K;
    if (u < nrows)
      for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) {
        const int item = rowind[ii];
        if (invI[item]) continue;
        const float wt = ifw_weight(u, item, userFreq, itemFreq, invPopU, invPopI, rhoRMS);
        const float itemRat = rowval[ii];
        const double diff = itemRat - (double)dotf(p, V + (int64_t)item * K, K, dot_mode);
        rmse += wt * diff * diff;
      }
    uRegErr += dotf(p, p, K, dot_mode);
  }
  uRegErr = uRegErr * uReg;
  for (int item = 0; item < nItems; item++) {
    if (invI[item]) continue;
    const float* q = V + (int64_t)item * K;
    iRegErr += dotf(q, q, K, dot_mode);
  }
  iRegErr = iRegErr * iReg;
  if (wsse_out) *wsse_out = rmse;
  return rmse + uRegErr + iRegErr;
}

// ---- ModelDropoutSigmoid (TMF) -------------------------------------------------------------
// adapDotProd (util.cpp:1067-1074) over the first `rank` dimensions; in device order the masked elements
// simply contribute nothing to their lane's chain
static inline float dotf_trunc(const float* p, const float* q, int K, int rank, int mode) {
  if (mode != ORC_DOT_TREE) {
    float prod = 0;
    for (int k = 0; k < rank; k++) prod += p[k] * q[k];
    return prod;
  }
  std::vector<float> qm(q, q + K), pm(p, p + K);
  for (int k = rank; k < K; k++) { qm[k] = 0.0f; pm[k] = 0.0f; }    // fma(0, 0, a) == a
  return dot_tree(pm.data(), qm.data(), K);
}
// modelDropoutSigmoid.cpp:158-172 (train) / :7-18 (estRating): the rank of a frequency.  The class's
// meanFreq/stdFreq are meanStdDev (util.cpp:278-294) of userFreq ++ itemFreq.
void orc_tmf_ranks(int32_t n, const double* freq, double meanFreq, double stdFreq, float rhoRMS, float alpha, int facDim,
                   int32_t* rank) {
  for (int i = 0; i < n; i++) {
    const double scaleFreq = (freq[i] - meanFreq) / stdFreq;
    const double sigmPc = 1.0 / (1.0 + exp(-rhoRMS * (scaleFreq - alpha)));
    int updMinRank = std::ceil(sigmPc * ((double)facDim));
    if (updMinRank < 1e-5) updMinRank = 1;        // train (:165-167); estRating asserts > 0 instead
    if (updMinRank > facDim) updMinRank = facDim;
    rank[i] = updMinRank;
  }
}
static inline int tmf_rank(int u, int item, const double* userFreq, const double* itemFreq, const int32_t* ru, const int32_t* ri) {
  return userFreq[u] < itemFreq[item] ? ru[u] : ri[item];       // isUMinFreq
}
// modelDropoutSigmoid.cpp:152-188 for a list of ratings
void orc_sgd_pass_tmf(int K, float* U, float* V, const int32_t* u, const int32_t* i, const float* r, const uint64_t* order,
                      int64_t n, float learnRate, float uReg, float iReg, const double* userFreq, const double* itemFreq,
                      const int32_t* ru, const int32_t* ri, int dot_mode) {
  for (int64_t t = 0; t < n; t++) {
    const int64_t ind = order ? (int64_t)order[t] : t;
    float* p = U + (int64_t)u[ind] * K;
    float* q = V + (int64_t)i[ind] * K;
    const int updMinRank = tmf_rank(u[ind], i[ind], userFreq, itemFreq, ru, ri);
    const float itemRat = r[ind];
    const float r_ui_est = dotf_trunc(p, q, K, updMinRank, dot_mode);
    const float diff = itemRat - r_ui_est;
    for (int k = 0; k < updMinRank; k++) p[k] -= learnRate * (-2.0 * diff * q[k] + 2.0 * uReg * p[k]);
    for (int k = 0; k < updMinRank; k++) q[k] -= learnRate * (-2.0 * diff * p[k] + 2.0 * iReg * q[k]);
  }
}
// Model::RMSE (model.cpp:214-251) through ModelDropoutSigmoid::estRating; also returns the squared error sum
double orc_rmse_tmf(int K, const float* U, const float* V, int32_t nUsers, int32_t nItems, int32_t nrows,
                    const int64_t* rowptr, const int32_t* rowind, const float* rowval, const uint8_t* invU,
                    const uint8_t* invI, const double* userFreq, const double* itemFreq, const int32_t* ru,
                    const int32_t* ri, int dot_mode, double* sse_out, int64_t* cnt) {
  int64_t nnz = 0;
  double rmse = 0;
  for (int u = 0; u < nUsers && u < nrows; u++) {
    if (invU[u]) continue;
    for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) {
      const int item = rowind[ii];
      if (item >= nItems || invI[item]) continue;
      const double r_ui_est = dotf_trunc(U + (int64_t)u * K, V + (int64_t)item * K, K, tmf_rank(u, item, userFreq, itemFreq, ru, ri), dot_mode);
      const double diff = rowval[ii] - r_ui_est;
      rmse += diff * diff;
      nnz++;
    }
  }
  if (sse_out) *sse_out = rmse;
  if (cnt) *cnt = nnz;
  return sqrt(rmse / nnz);
}

// ---- ModelPoissonDropout (TMF + Dropout) ---------------------------------------------------------
// initCDFRanks (modelPoissonDropout.cpp:25-47): for lambda = 1..facDim the smallest k whose Poisson(lambda) CDF over
// 0..k+1 reaches 0.99; estRating uses dimensions 0..cdfRanks[lambda-1] (:17).  factorial as the class builds it.
void orc_cdf_ranks(int facDim, int32_t* cdfRanks) {
  std::vector<double> factorial;
  factorial.push_back(1);
  for (int i = 1; i <= facDim + 1; i++) factorial.push_back(factorial.back() * ((double)i));
  for (int lambda = 1; lambda <= facDim; lambda++) {
    double cdf = std::exp(-lambda) * (std::pow(lambda, 0) / factorial[0]);
    int k = 0;
    for (k = 0; k < facDim; k++) {
      const double wt = std::exp(-lambda) * (std::pow(lambda, k + 1) / factorial[k + 1]);
      cdf += wt;
      if (cdf >= 0.99) break;
    }
    cdfRanks[lambda - 1] = k;
    if (k == facDim) cdfRanks[lambda - 1] = k - 1;
  }
}
// The draw of this build (include/mfx.h, mfx_set_tmf_dropout): hash of (seed, epoch, user, item) -> (0,1) -> inverse
// Poisson CDF by sequential search in double, clipped to [1, K] as :202-207 clip std::poisson_distribution's draw.
static inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
int32_t orc_poisson_rank(int32_t lambda, uint32_t seed, uint32_t epoch, uint32_t u, uint32_t item, int32_t K) {
  uint32_t h = mix32(seed * 0x9e3779b1U + epoch * 0x85ebca6bU + 0x2545f491U);
  h = mix32(h ^ (u * 0xc2b2ae35U + 0x27d4eb2fU));
  h = mix32(h ^ (item * 0x165667b1U + 0x9e3779b9U));
  const double x = ((double)h + 0.5) * (1.0 / 4294967296.0);
  double p = exp(-(double)lambda), F = p;
  int k = 0;
  while (x > F && k < 4 * K + 64) { k++; p = p * (double)lambda / (double)k; F += p; }
  return k < 1 ? 1 : (k > K ? K : k);
}
// modelPoissonDropout.cpp:186-221 for a list of ratings: lambda of the rarer side, updRank drawn, truncated visit
void orc_sgd_pass_tmfd(int K, float* U, float* V, const int32_t* u, const int32_t* i, const float* r, const uint64_t* order,
                       int64_t n, float learnRate, float uReg, float iReg, const double* userFreq, const double* itemFreq,
                       const int32_t* lu, const int32_t* li, uint32_t seed, uint32_t epoch, int dot_mode) {
  for (int64_t t = 0; t < n; t++) {
    const int64_t ind = order ? (int64_t)order[t] : t;
    float* p = U + (int64_t)u[ind] * K;
    float* q = V + (int64_t)i[ind] * K;
    const int lambda = userFreq[u[ind]] < itemFreq[i[ind]] ? lu[u[ind]] : li[i[ind]];
    const int updRank = orc_poisson_rank(lambda, seed, epoch, (uint32_t)u[ind], (uint32_t)i[ind], K);
    const float itemRat = r[ind];
    const float r_ui_est = dotf_trunc(p, q, K, updRank, dot_mode);
    const float diff = itemRat - r_ui_est;
    for (int k = 0; k < updRank; k++) p[k] -= learnRate * (-2.0 * diff * q[k] + 2.0 * uReg * p[k]);
    for (int k = 0; k < updRank; k++) q[k] -= learnRate * (-2.0 * diff * p[k] + 2.0 * iReg * q[k]);
  }
}

void orc_sgd_hogwild(int K, float* U, float* V, const int32_t* u, const int32_t* i,
                     const float* r, const uint64_t* order, int64_t n, float lr,
                     float uReg, float iReg, int arith, int dot_mode, int nthreads) {
#pragma omp parallel for num_threads(nthreads)
  for (int64_t t = 0; t < n; t++) {
    int64_t ind = order ? (int64_t)order[t] : t;
    sgd_update(U + (int64_t)u[ind] * K, V + (int64_t)i[ind] * K, r[ind], K, lr, uReg, iReg,
               arith, dot_mode);
  }
}

struct Strat {
  int T;
  std::vector<std::unordered_set<int>> usersPart, itemsPart;
};
// modelMF.cpp:191-203,229-265
void* orc_strat_create(void* mth, int32_t nrows, int32_t ncols, const uint8_t* invU,
                       const uint8_t* invI, int T) {
  std::mt19937& mt = *(std::mt19937*)mth;
  std::vector<int> trainUsers, trainItems;
  for (int u = 0; u < nrows; u++) if (!invU[u]) trainUsers.push_back(u);
  for (int it = 0; it < ncols; it++) if (!invI[it]) trainItems.push_back(it);
  std::shuffle(trainUsers.begin(), trainUsers.end(), mt);
  std::shuffle(trainItems.begin(), trainItems.end(), mt);
  Strat* s = new Strat;
  s->T = T;
  s->usersPart.resize(T);
  s->itemsPart.resize(T);
  int usersPerPart = (int)trainUsers.size() / T;
  int currPart = 0;
  for (int i = 0; i < (int)trainUsers.size(); i++) {
    s->usersPart[currPart].insert(trainUsers[i]);
    if (i != 0 && i % usersPerPart == 0)
      if (currPart != T - 1) currPart++;
  }
  int itemsPerPart = (int)trainItems.size() / T;
  currPart = 0;
  for (int i = 0; i < (int)trainItems.size(); i++) {
    s->itemsPart[currPart].insert(trainItems[i]);
    if (i != 0 && i % itemsPerPart == 0)
      if (currPart != T - 1) currPart++;
  }
  return s;
}
void orc_strat_free(void* h) { delete (Strat*)h; }
void orc_strat_parts(void* h, int32_t nrows, int32_t ncols, int32_t* userPart,
                     int32_t* itemPart) {
  Strat* s = (Strat*)h;
  for (int u = 0; u < nrows; u++) userPart[u] = -1;
  for (int i = 0; i < ncols; i++) itemPart[i] = -1;
  for (int t = 0; t < s->T; t++) {
    for (int u : s->usersPart[t]) userPart[u] = t;
    for (int i : s->itemsPart[t]) itemPart[i] = t;
  }
}
// modelMF.cpp:273-304
void orc_strat_epoch(void* h, void* mth, int K, float* U, float* V, const int64_t* rowptr,
                     const int32_t* rowind, const float* rowval, float lr, float uReg,
                     float iReg, int dot_mode) {
  Strat* s = (Strat*)h;
  std::mt19937& mt = *(std::mt19937*)mth;
  int maxThreads = s->T;
  std::vector<std::pair<int, int>> updateSeq;
  for (int k = 0; k < maxThreads; k++) {
    block_seq(maxThreads, updateSeq, mt);
    for (int t = 0; t < maxThreads; t++) {
      const auto& users = s->usersPart[updateSeq[t].first];
      const auto& items = s->itemsPart[updateSeq[t].second];
      for (const auto& u : users)
        for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) {
          int item = rowind[ii];
          if (items.count(item) == 0) continue;
          sgd_update(U + (int64_t)u * K, V + (int64_t)item * K, rowval[ii], K, lr, uReg,
                     iReg, ORC_ARITH_REF64F, dot_mode);
        }
    }
  }
}

// ---------------------------------------------------------------------------
// objective / RMSE
// ---------------------------------------------------------------------------
// model.cpp:1770-1815.  Sums are double; the reference's OpenMP reduction order is
// thread-dependent, this is the 1-thread order.
double orc_objective(int K, const float* U, const float* V, int32_t nUsers, int32_t nItems,
                     int32_t nrows, const int64_t* rowptr, const int32_t* rowind,
                     const float* rowval, const uint8_t* invU, const uint8_t* invI,
                     float uReg, float iReg, int dot_mode, double* sse_out,
                     double* unorm2, double* inorm2) {
  double rmse = 0, uRegErr = 0, iRegErr = 0;
  for (int u = 0; u < nUsers; u++) {
    if (invU[u]) continue;
    const float* p = U + (int64_t)u * K;
    if (u < nrows)
      for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) {
        int item = rowind[ii];
        if (invI[item]) continue;
        float itemRat = rowval[ii];
        double diff = itemRat - (double)dotf(p, V + (int64_t)item * K, K, dot_mode);
        rmse += diff * diff;
      }
    uRegErr += dotf(p, p, K, dot_mode);
  }
  if (unorm2) *unorm2 = uRegErr;
  uRegErr = uRegErr * uReg;
  for (int item = 0; item < nItems; item++) {
    if (invI[item]) continue;
    const float* q = V + (int64_t)item * K;
    iRegErr += dotf(q, q, K, dot_mode);
  }
  if (inorm2) *inorm2 = iRegErr;
  iRegErr = iRegErr * iReg;
  if (sse_out) *sse_out = rmse;
  return rmse + uRegErr + iRegErr;
}

// model.cpp:214-251
double orc_rmse(int K, const float* U, const float* V, int32_t nUsers, int32_t nItems,
                int32_t nrows, const int64_t* rowptr, const int32_t* rowind,
                const float* rowval, const uint8_t* invU, const uint8_t* invI, int dot_mode,
                double* sse_out, int64_t* cnt) {
  int64_t nnz = 0;
  double rmse = 0;
  // the reference indexes mat->rowptr[u] for every u < nUsers (model.cpp:223-231);
  // rows the matrix does not have are treated as empty here.
  for (int u = 0; u < nUsers && u < nrows; u++) {
    if (invU[u]) continue;
    for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) {
      int item = rowind[ii];
      if (item >= nItems || invI[item]) continue;
      double r_ui = rowval[ii];
      double r_ui_est = dotf(U + (int64_t)u * K, V + (int64_t)item * K, K, dot_mode);
      double diff = r_ui - r_ui_est;
      rmse += diff * diff;
      nnz++;
    }
  }
  if (sse_out) *sse_out = rmse;
  if (cnt) *cnt = nnz;
  return sqrt(rmse / nnz);
}

// ---------------------------------------------------------------------------
// ALS
// ---------------------------------------------------------------------------
// Eigen/src/Cholesky/LDLT.h (Eigen 3.3/3.4; not in the reference tree => unpinned):
// internal::ldlt_inplace<Lower>::unblocked + LDLT::_solve_impl.  Inner products run
// j = 0..k-1 sequentially (Eigen's GEMV order is an implementation detail).
void orc_ldlt_solve(int n, float* A, const float* b, float* x) {
  std::vector<int> tr(n);
  std::vector<float> temp(n);
#define M(i, j) A[(int64_t)(i) * n + (j)]
  for (int k = 0; k < n; k++) {
    int big = k;
    float bigv = std::fabs(M(k, k));
    for (int i = k + 1; i < n; i++)
      if (std::fabs(M(i, i)) > bigv) { bigv = std::fabs(M(i, i)); big = i; }
    tr[k] = big;
    if (k != big) {
      int s = n - big - 1;
      for (int j = 0; j < k; j++) std::swap(M(k, j), M(big, j));
      for (int i = 0; i < s; i++) std::swap(M(big + 1 + i, k), M(big + 1 + i, big));
      std::swap(M(k, k), M(big, big));
      for (int i = k + 1; i < big; i++) {
        float tmp = M(i, k);
        M(i, k) = M(big, i);
        M(big, i) = tmp;
      }
    }
    int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; j++) temp[j] = M(j, j) * M(k, j);
      float acc = 0;
      for (int j = 0; j < k; j++) acc += M(k, j) * temp[j];
      M(k, k) -= acc;
      for (int i = 0; i < rs; i++) {
        float a = 0;
        for (int j = 0; j < k; j++) a += M(k + 1 + i, j) * temp[j];
        M(k + 1 + i, k) -= a;
      }
    }
    float realAkk = M(k, k);
    bool pivot_is_valid = std::fabs(realAkk) > 0.0f;
    if (k == 0 && !pivot_is_valid) {
      for (int j = 0; j < n; j++) tr[j] = j;
      break;
    }
    if (rs > 0 && pivot_is_valid)
      for (int i = 0; i < rs; i++) M(k + 1 + i, k) /= realAkk;
  }
  // solve: dst = P b; L^-1; D^-1 (tolerance = numeric_limits<float>::min()); L^-T; P^T
  for (int i = 0; i < n; i++) x[i] = b[i];
  for (int k = 0; k < n; k++) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
  for (int i = 0; i < n; i++) {
    float a = x[i];
    for (int j = 0; j < i; j++) a -= M(i, j) * x[j];
    x[i] = a;
  }
  const float tolerance = std::numeric_limits<float>::min();
  for (int i = 0; i < n; i++) {
    if (std::fabs(M(i, i)) > tolerance) x[i] /= M(i, i);
    else x[i] = 0;
  }
  for (int i = n - 1; i >= 0; i--) {
    float a = x[i];
    for (int j = i + 1; j < n; j++) a -= M(j, i) * x[j];
    x[i] = a;
  }
  for (int k = n - 1; k >= 0; k--) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
#undef M
}

// modelMF.cpp:805-841 (users) / :844-880 (items): per row, full K x K Gramian of the
// rated counterpart rows (both triangles, ratings <= 0 skipped), +reg on the
// diagonal (not degree-scaled), LDLT solve.  Thread-count independent.
void orc_als_half(int side, int K, float* X, const float* Y, int32_t nX, const int64_t* ptr,
                  const int32_t* ind, const float* val, const uint8_t* invX, float reg,
                  int nthreads) {
  (void)side;
#pragma omp parallel num_threads(nthreads)
  {
    std::vector<float> YTY((size_t)K * K), b(K), sol(K);
#pragma omp for schedule(dynamic, 64)
    for (int x = 0; x < nX; x++) {
      if (invX[x]) continue;
      std::fill(YTY.begin(), YTY.end(), 0.0f);
      std::fill(b.begin(), b.end(), 0.0f);
      for (int64_t ii = ptr[x]; ii < ptr[x + 1]; ii++) {
        const float* y = Y + (int64_t)ind[ii] * K;
        float rating = val[ii];
        if (rating > 0) {
          for (int j = 0; j < K; j++) {
            for (int k = 0; k < K; k++) YTY[(size_t)j * K + k] += y[j] * y[k];
            b[j] += rating * y[j];
          }
        }
      }
      for (int j = 0; j < K; j++) YTY[(size_t)j * K + j] += reg;
      orc_ldlt_solve(K, YTY.data(), b.data(), sol.data());
      for (int j = 0; j < K; j++) X[(int64_t)x * K + j] = sol[j];
    }
  }
}

// ---------------------------------------------------------------------------
// CCD++
// ---------------------------------------------------------------------------
// modelMF.cpp:1027-1121 for one k (FreqAdap: :1272-1360).  Products are float*float,
// num/denom accumulate in double, the quotient is stored as float.
void orc_ccdpp_rank1(int K, int k, float* U, float* V, int32_t nUsers, int32_t nItems,
                     int32_t ncols, const int64_t* rowptr, const int32_t* rowind,
                     float* res_row, const int64_t* colptr, const int32_t* colind,
                     float* res_col, const uint8_t* invU, const uint8_t* invI, float uReg,
                     float iReg, int add_back, int inner, float freq_thresh, int nthreads) {
  std::vector<float> u_k(nUsers), v_k(nItems);
  for (int u = 0; u < nUsers; u++) u_k[u] = U[(int64_t)u * K + k];
  for (int i = 0; i < nItems; i++) v_k[i] = V[(int64_t)i * K + k];
  if (add_back) {  // iter > 0
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256)
    for (int u = 0; u < nUsers; u++) {
      if (invU[u]) continue;
      for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++)
        res_row[ii] += U[(int64_t)u * K + k] * V[(int64_t)rowind[ii] * K + k];
    }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256)
    for (int item = 0; item < nItems; item++) {
      if (invI[item] || item >= ncols) continue;
      for (int64_t uu = colptr[item]; uu < colptr[item + 1]; uu++)
        res_col[uu] += U[(int64_t)colind[uu] * K + k] * V[(int64_t)item * K + k];
    }
  }
  for (int subIter = 0; subIter < inner; subIter++) {
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256)
    for (int u = 0; u < nUsers; u++) {
      if (invU[u]) continue;
      double num = 0, denom = uReg, newV;
      for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) {
        int item = rowind[ii];
        num += res_row[ii] * v_k[item];
        denom += v_k[item] * v_k[item];
      }
      newV = num / denom;
      u_k[u] = newV;
    }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256)
    for (int item = 0; item < nItems; item++) {
      if (invI[item] || item >= ncols) continue;
      double num = 0, denom = iReg, newV;
      for (int64_t uu = colptr[item]; uu < colptr[item + 1]; uu++) {
        int u = colind[uu];
        num += res_col[uu] * u_k[u];
        denom += u_k[u] * u_k[u];
      }
      newV = num / denom;
      v_k[item] = newV;
      if (freq_thresh >= 0) {  // modelMF.cpp:1336-1342; itemFreq = column count
        double itemFreq = (double)(colptr[item + 1] - colptr[item]);
        if (itemFreq < freq_thresh && k > 0) v_k[item] = 0;
      }
    }
  }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256)
  for (int u = 0; u < nUsers; u++) {
    if (invU[u]) continue;
    for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++)
      res_row[ii] -= u_k[u] * v_k[rowind[ii]];
  }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256)
  for (int item = 0; item < nItems; item++) {
    if (invI[item] || item >= ncols) continue;
    for (int64_t uu = colptr[item]; uu < colptr[item + 1]; uu++)
      res_col[uu] -= u_k[colind[uu]] * v_k[item];
  }
  for (int u = 0; u < nUsers; u++) U[(int64_t)u * K + k] = u_k[u];
  for (int i = 0; i < nItems; i++) V[(int64_t)i * K + k] = v_k[i];
}

// util.cpp:847-864
static int binSearch(const int32_t* sortedArr, int key, int64_t ub, int64_t lb) {
  int64_t ind = -1;
  while (ub >= lb) {
    int64_t midP = (ub + lb) / 2;
    if (sortedArr[midP] == key) { ind = midP; break; }
    else if (sortedArr[midP] < key) lb = midP + 1;
    else ub = midP - 1;
  }
  return (int)ind;
}

// modelMF.cpp:1528-1605, sequential (the reference shares one mt across its
// OpenMP threads, so only the 1-thread order is defined).
void orc_ccd_iter(int K, float* U, float* V, int32_t nUsers, int32_t nItems, int32_t ncols,
                  const int64_t* rowptr, const int32_t* rowind, float* res_row,
                  const int64_t* colptr, const int32_t* colind, float* res_col,
                  const uint8_t* invU, const uint8_t* invI, float uReg, float iReg,
                  void* mth, uint16_t* uorder, uint16_t* iorder, int orders_given) {
  // uorder [nUsers][K] / iorder [nItems][K]: the factor order of every row.  orders_given: take them
  // from the caller; otherwise they come from std::shuffle(udims, mt) as in the reference and are
  // written out (when non-NULL) so that a test can replay them elsewhere.
  std::mt19937 dummy;
  std::mt19937& mt = mth ? *(std::mt19937*)mth : dummy;
  std::vector<int> dims(K);
  std::iota(dims.begin(), dims.end(), 0);
  auto row_order = [&](uint16_t* store, int row) {
    std::vector<int> udims(dims);
    if (orders_given) { for (int s = 0; s < K; s++) udims[s] = store[(int64_t)row * K + s]; }
    else {
      std::shuffle(udims.begin(), udims.end(), mt);
      if (store) for (int s = 0; s < K; s++) store[(int64_t)row * K + s] = (uint16_t)udims[s];
    }
    return udims;
  };
#define UF(u, k) U[(int64_t)(u) * K + (k)]
#define IF(i, k) V[(int64_t)(i) * K + (k)]
  for (int u = 0; u < nUsers; u++) {
    if (invU[u]) continue;
    std::vector<int> udims = row_order(uorder, u);
    for (const auto& k : udims) {
      double num = 0, denom = uReg, newV;
      for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) {
        int item = rowind[ii];
        num += (res_row[ii] + UF(u, k) * IF(item, k)) * IF(item, k);
        denom += IF(item, k) * IF(item, k);
      }
      newV = num / denom;
      for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) {
        int item = rowind[ii];
        double upd = (newV - UF(u, k)) * IF(item, k);
        res_row[ii] -= upd;
        int binInd = binSearch(colind, u, colptr[item + 1] - 1, colptr[item]);
        if (binInd != -1) res_col[binInd] -= upd;
      }
      UF(u, k) = newV;
    }
  }
  for (int item = 0; item < nItems; item++) {
    if (invI[item] || item >= ncols) continue;
    std::vector<int> udims = row_order(iorder, item);
    for (const auto& k : udims) {
      double num = 0, denom = iReg, newV;
      for (int64_t uu = colptr[item]; uu < colptr[item + 1]; uu++) {
        int u = colind[uu];
        num += (res_col[uu] + UF(u, k) * IF(item, k)) * UF(u, k);
        denom += UF(u, k) * UF(u, k);
      }
      newV = num / denom;
      for (int64_t uu = colptr[item]; uu < colptr[item + 1]; uu++) {
        int u = colind[uu];
        double upd = (newV - IF(item, k)) * UF(u, k);
        res_col[uu] -= upd;
        int binInd = binSearch(rowind, item, rowptr[u + 1] - 1, rowptr[u]);
        if (binInd != -1) res_row[binInd] -= upd;
      }
      IF(item, k) = newV;
    }
  }
#undef UF
#undef IF
}

// ---------------------------------------------------------------------------
// full training loops
// ---------------------------------------------------------------------------
namespace {
struct OModel {  // the fields of Model that *this = bestModel copies (model.h:24-41)
  float learnRate;
  std::vector<float> U, V;
};
}  // namespace

int orc_train(const orc_train_cfg* c, float* U, float* V, float* Ubest, float* Vbest,
              double* objTraj, double* valTraj, int32_t* bestIterOut, float* finalLR,
              uint8_t* invU, uint8_t* invI) {
  const int K = c->K, nU = c->nUsers, nI = c->nItems;
  const int64_t nnz = c->tr_rowptr[c->tr_nrows];
  const float uReg = c->uReg, iReg = c->iReg;
  const int dm = c->dot_mode;
  // modelMF.cpp:37-45
  orc_invalid(c->tr_nrows, c->tr_ncols, c->tr_rowptr, c->tr_rowind, nU, nI, invU, invI);

  OModel cur, best;
  cur.learnRate = c->learnRate;
  cur.U.assign(U, U + (size_t)nU * K);
  cur.V.assign(V, V + (size_t)nI * K);
  best = cur;  // main.cpp:1326-1327: both models built from the same params + seed

  auto objective = [&]() {
    return orc_objective(K, cur.U.data(), cur.V.data(), nU, nI, c->tr_nrows, c->tr_rowptr,
                         c->tr_rowind, c->tr_rowval, invU, invI, uReg, iReg, dm, nullptr,
                         nullptr, nullptr);
  };
  auto valrmse = [&]() {
    return orc_rmse(K, cur.U.data(), cur.V.data(), nU, nI, c->va_nrows, c->va_rowptr,
                    c->va_rowind, c->va_rowval, invU, invI, dm, nullptr, nullptr);
  };

  int bestIter = -1;
  double prevObj = objective();            // modelMF.cpp:48-50
  double bestValRMSE = valrmse(), prevValRMSE = bestValRMSE;
  (void)prevValRMSE;

  std::mt19937 mt(c->seed);                // modelMF.cpp:63 (trainSeed = params.seed)
  // getUIRatings (util.cpp:722-747): CSR order, invalid users/items removed
  std::vector<int32_t> ru, ri; std::vector<float> rr;
  std::vector<uint64_t> inds;
  const int m = c->method;
  if (m == ORC_M_SGD || m == ORC_M_HOGSGD) {
    ru.reserve(nnz); ri.reserve(nnz); rr.reserve(nnz);
    for (int u = 0; u < c->tr_nrows; u++) {
      if (invU[u]) continue;
      for (int64_t e = c->tr_rowptr[u]; e < c->tr_rowptr[u + 1]; e++) {
        if (invI[c->tr_rowind[e]]) continue;
        ru.push_back(u); ri.push_back(c->tr_rowind[e]); rr.push_back(c->tr_rowval[e]);
      }
    }
    inds.resize(ru.size());
    std::iota(inds.begin(), inds.end(), 0);
  }
  std::vector<uint64_t> validUsers;        // modelMF.cpp:620-625
  if (m == ORC_M_SGDU)
    for (int u = 0; u < nU; u++) if (!invU[u]) validUsers.push_back(u);
  Strat* strat = nullptr;
  if (m == ORC_M_SGDPAR)
    strat = (Strat*)orc_strat_create(&mt, c->tr_nrows, c->tr_ncols, invU, invI, c->nthreads);
  std::vector<int32_t> dims(K);
  std::iota(dims.begin(), dims.end(), 0);
  std::vector<float> res_row, res_col;
  if (m == ORC_M_CCDPP || m == ORC_M_CCDPP_FA || m == ORC_M_CCD) {
    res_row.assign(c->tr_rowval, c->tr_rowval + nnz);   // gk_csr_Dup, modelMF.cpp:1013
    res_col.assign(c->tr_colval, c->tr_colval + nnz);
    std::fill(cur.U.begin(), cur.U.end(), 0.0f);        // uFac.fill(0), :1020
  }

  int iter;
  for (iter = 0; iter < c->maxIter; iter++) {
    float lr = cur.learnRate;
    switch (m) {
      case ORC_M_SGD:                      // modelMF.cpp:76-105
      case ORC_M_HOGSGD:                   // modelMF.cpp:1739-1763
        if (iter % 10 == 0) std::shuffle((size_t*)inds.data(), (size_t*)inds.data() + inds.size(), mt);
        else orc_mt_par_block_shuffle_u64(&mt, inds.data(), (int64_t)inds.size(), c->nthreads);
        if (m == ORC_M_SGD)
          orc_sgd_pass(K, cur.U.data(), cur.V.data(), ru.data(), ri.data(), rr.data(),
                       inds.data(), (int64_t)inds.size(), lr, uReg, iReg, ORC_ARITH_REF64, dm);
        else
          orc_sgd_hogwild(K, cur.U.data(), cur.V.data(), ru.data(), ri.data(), rr.data(),
                          inds.data(), (int64_t)inds.size(), lr, uReg, iReg, ORC_ARITH_F32, dm,
                          c->nthreads);
        break;
      case ORC_M_SGDPAR:
        orc_strat_epoch(strat, &mt, K, cur.U.data(), cur.V.data(), c->tr_rowptr, c->tr_rowind,
                        c->tr_rowval, lr, uReg, iReg, dm);
        break;
      case ORC_M_SGDU:                     // modelMF.cpp:635-659
        std::shuffle((size_t*)validUsers.data(), (size_t*)validUsers.data() + validUsers.size(), mt);
        for (uint64_t u : validUsers)
          for (int64_t ii = c->tr_rowptr[u]; ii < c->tr_rowptr[u + 1]; ii++)
            sgd_update(cur.U.data() + u * K, cur.V.data() + (int64_t)c->tr_rowind[ii] * K,
                       c->tr_rowval[ii], K, lr, uReg, iReg, ORC_ARITH_REF64, dm);
        break;
      case ORC_M_ALS:                      // modelMF.cpp:795-882
        orc_als_half(0, K, cur.U.data(), cur.V.data(), std::min(nU, c->tr_nrows), c->tr_rowptr,
                     c->tr_rowind, c->tr_rowval, invU, uReg, c->nthreads);
        orc_als_half(1, K, cur.V.data(), cur.U.data(), std::min(nI, c->tr_ncols), c->tr_colptr,
                     c->tr_colind, c->tr_colval, invI, iReg, c->nthreads);
        break;
      case ORC_M_CCDPP:
      case ORC_M_CCDPP_FA:
        if (m == ORC_M_CCDPP) std::shuffle(dims.begin(), dims.end(), mt);  // :1026 vs :1271
        for (int k : dims)
          orc_ccdpp_rank1(K, k, cur.U.data(), cur.V.data(), nU, nI, c->tr_ncols, c->tr_rowptr,
                          c->tr_rowind, res_row.data(), c->tr_colptr, c->tr_colind,
                          res_col.data(), invU, invI, uReg, iReg, iter > 0, 5,
                          m == ORC_M_CCDPP_FA ? 75.0f : -1.0f, c->nthreads);
        break;
      case ORC_M_CCD:
        orc_ccd_iter(K, cur.U.data(), cur.V.data(), nU, nI, c->tr_ncols, c->tr_rowptr,
                     c->tr_rowind, res_row.data(), c->tr_colptr, c->tr_colind, res_col.data(),
                     invU, invI, uReg, iReg, &mt, nullptr, nullptr, 0);
        break;
    }

    // ---- Model::isTerminateModel, model.cpp:1471-1540 (OBJ_ITER = 1) ----
    bool ret = false;
    double currObj = objective();
    double currValRMSE = valrmse();
    if (objTraj) objTraj[iter] = currObj;
    if (valTraj) valTraj[iter] = currValRMSE;
    if (currObj != currObj || currValRMSE != currValRMSE) {
      if (cur.learnRate > 1e-5) {
        cur = best;                        // *this = bestModel (learnRate comes along)
        cur.learnRate = cur.learnRate / 2;
        continue;                          // return false
      } else {
        iter++;
        break;                             // return true
      }
    }
    if (currValRMSE < bestValRMSE) {
      best = cur;
      bestValRMSE = currValRMSE;
      bestIter = iter;
    }
    if (iter - bestIter >= 100)
      if (cur.learnRate > 1e-5) cur.learnRate = cur.learnRate / 2;
    if (iter - bestIter >= 500) ret = true;              // CHANCE_ITER
    if (fabs(prevObj - currObj) < 1e-5) ret = true;      // EPS
    prevObj = currObj;
    prevValRMSE = currValRMSE;
    if (ret) { iter++; break; }
  }
  if (strat) orc_strat_free(strat);
  memcpy(U, cur.U.data(), sizeof(float) * (size_t)nU * K);
  memcpy(V, cur.V.data(), sizeof(float) * (size_t)nI * K);
  memcpy(Ubest, best.U.data(), sizeof(float) * (size_t)nU * K);
  memcpy(Vbest, best.V.data(), sizeof(float) * (size_t)nI * K);
  if (bestIterOut) *bestIterOut = bestIter;
  if (finalLR) *finalLR = cur.learnRate;
  return iter;
}

// ---------------------------------------------------------------------------
// cpu_baseline timing
// ---------------------------------------------------------------------------
int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// modelMF.cpp:1746-1767 bracket: the parallel-for only (the shuffle is outside).
// colmajor = 1 stores the factors like Eigen::MatrixXf (element (r,k) at k*n + r).
double orc_time_hogwild(int K, int32_t nU, int32_t nI, float* U, float* V, const int32_t* u,
                        const int32_t* i, const float* r, int64_t n, float lr, float uReg,
                        float iReg, int nthreads, int colmajor, int epochs) {
  auto t0 = std::chrono::steady_clock::now();
  for (int ep = 0; ep < epochs; ep++) {
    if (!colmajor) {
      orc_sgd_hogwild(K, U, V, u, i, r, nullptr, n, lr, uReg, iReg, ORC_ARITH_F32, ORC_DOT_SEQ,
                      nthreads);
    } else {
      const int64_t su = nU, si = nI;
#pragma omp parallel for num_threads(nthreads)
      for (int64_t t = 0; t < n; t++) {
        float* p = U + u[t];
        float* q = V + i[t];
        float s = p[0] * q[0];
        for (int k = 1; k < K; k++) s = s + p[k * su] * q[k * si];
        double r_ui_est = s;
        const double diff = r[t] - r_ui_est;
        const float c1 = (float)(-2.0 * diff);
        const float cu = (float)(2.0 * uReg), ci = (float)(2.0 * iReg);
        for (int k = 0; k < K; k++) p[k * su] = p[k * su] - lr * (c1 * q[k * si] + cu * p[k * su]);
        for (int k = 0; k < K; k++) q[k * si] = q[k * si] - lr * (c1 * p[k * su] + ci * q[k * si]);
      }
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

// modelMF.cpp:271-309 bracket: trainSGDPar's epoch as the reference runs it -- T rounds, the T disjoint blocks of a
// round in an OpenMP parallel for (one block per thread), each block scanning its users' rows and skipping the
// items outside its item part through unordered_set::count (:279-285).  Returns seconds for `epochs` epochs.
double orc_time_strat(void* h, void* mth, int K, float* U, float* V, const int64_t* rowptr,
                      const int32_t* rowind, const float* rowval, float lr, float uReg, float iReg,
                      int epochs) {
  Strat* s = (Strat*)h;
  std::mt19937& mt = *(std::mt19937*)mth;
  const int T = s->T;
  std::vector<std::pair<int, int>> updateSeq;
  auto t0 = std::chrono::steady_clock::now();
  for (int ep = 0; ep < epochs; ep++)
    for (int k = 0; k < T; k++) {
      block_seq(T, updateSeq, mt);
#pragma omp parallel for num_threads(T) schedule(static, 1)
      for (int t = 0; t < T; t++) {
        const auto& users = s->usersPart[updateSeq[t].first];
        const auto& items = s->itemsPart[updateSeq[t].second];
        for (const auto& u : users)
          for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) {
            const int item = rowind[ii];
            if (items.count(item) == 0) continue;
            sgd_update(U + (int64_t)u * K, V + (int64_t)item * K, rowval[ii], K, lr, uReg, iReg,
                       ORC_ARITH_REF64F, ORC_DOT_SEQ);
          }
      }
    }
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

// ---------------------------------------------------------------------------
// data preparation in front of the path (io.cpp:410-459, 726-787): the RNG-driven parts
// ---------------------------------------------------------------------------
// io.cpp:413-437: colour 1 = test (nTest draws with replacement), 2 = val (nVal uncoloured ratings), 0 = train
void orc_split_colors(int64_t nnz, float testPc, float valPc, int seed, int32_t* color) {
  int n = (int)nnz;
  int nTest = testPc * n;
  int nVal = valPc * n;
  memset(color, 0, sizeof(int32_t) * (size_t)n);
  std::mt19937 mt(seed);
  std::uniform_int_distribution<int> nnzDist(0, n - 1);
  for (int i = 0; i < nTest; i++) {
    int k = nnzDist(mt);
    color[k] = 1;
  }
  int i = 0;
  while (i < nVal) {
    int k = nnzDist(mt);
    if (!color[k]) {
      color[k] = 2;
      i++;
    }
  }
}
// io.cpp:730-767: the sampled (user, item) pairs of writeRandMatCSR.  Two-call protocol: pairs == NULL returns the count;
// otherwise fills pairs[2*t] = user, pairs[2*t+1] = item, users ascending and items ascending inside a user.
int64_t orc_rand_pairs(int32_t nUsers, int32_t nItems, int seed, int32_t nnz, int32_t* pairs) {
  std::vector<std::unordered_set<int>> uItemSet(nUsers);
  std::mt19937 mt(seed);
  std::uniform_int_distribution<int> uDist(0, nUsers - 1);
  std::uniform_int_distribution<int> iDist(0, nItems - 1);
  for (int u = 0; u < nUsers; u++) {
    int item = iDist(mt);
    uItemSet[u].insert(item);
  }
  for (int item = 0; item < nItems; item++) {
    int user = uDist(mt);
    uItemSet[user].insert(item);
  }
  auto nTuples = [&]() { int64_t c = 0; for (auto& s : uItemSet) c += (int64_t)s.size(); return c; };
  int64_t nPairs = nTuples();
  while (nPairs < nnz) {
    for (int64_t i = 0; i < nnz - nPairs; i++) {
      int user = uDist(mt);
      int item = iDist(mt);
      uItemSet[user].insert(item);
    }
    nPairs = nTuples();
  }
  if (pairs) {
    int64_t t = 0;
    for (int u = 0; u < nUsers; u++) {
      std::vector<int> items(uItemSet[u].begin(), uItemSet[u].end());
      std::sort(items.begin(), items.end());
      for (int item : items) { pairs[2 * t] = u; pairs[2 * t + 1] = item; t++; }
    }
  }
  return nPairs;
}

// ---------------------------------------------------------------------------
// ModelMFBias (modelMFBias.cpp)
// ---------------------------------------------------------------------------
// :163-197: the sequential loop over the rating tuples in order[] (NULL: 0..n-1)
void orc_bias_pass(float* uBias, float* iBias, const int32_t* u, const int32_t* i, const float* r, const uint64_t* order,
                   int64_t n, float learnRate, float uReg, float iReg) {
  for (int64_t t = 0; t < n; t++) {
    const int64_t ind = order ? (int64_t)order[t] : t;
    const int user = u[ind], item = i[ind];
    const float itemRat = r[ind];
    double r_ui_est = uBias[user] + iBias[item];                  // estRating (:94-99): float sum widened
    double diff = itemRat - r_ui_est;
    uBias[user] -= learnRate * (-2.0 * diff + 2.0 * uReg * uBias[user]);
    iBias[item] -= learnRate * (-2.0 * diff + 2.0 * iReg * iBias[item]);
  }
}
// :40-91 and Model::RMSE (model.cpp:214-251) through the class's estRating.  Returns the objective; *sse, *cnt,
// *ubReg = sum uBias^2, *ibReg = sum iBias^2 over the valid users / items.
double orc_bias_eval(const float* uBias, const float* iBias, int32_t nUsers, int32_t nItems, int32_t nrows, const int64_t* rowptr,
                     const int32_t* rowind, const float* rowval, const uint8_t* invU, const uint8_t* invI, float uReg, float iReg,
                     double* sse_out, int64_t* cnt_out, double* ub_out, double* ib_out) {
  double rmse = 0, uBiasReg = 0, iBiasReg = 0;
  int64_t cnt = 0;
  for (int u = 0; u < nUsers && u < nrows; u++) {
    if (invU[u]) continue;
    for (int64_t ii = rowptr[u]; ii < rowptr[u + 1]; ii++) {
      const int item = rowind[ii];
      if (item >= nItems || invI[item]) continue;
      double diff = rowval[ii] - (double)(uBias[u] + iBias[item]);
      rmse += diff * diff;
      cnt++;
    }
  }
  for (int u = 0; u < nUsers; u++) if (!invU[u]) uBiasReg += uBias[u] * uBias[u];
  for (int item = 0; item < nItems; item++) if (!invI[item]) iBiasReg += iBias[item] * iBias[item];
  if (sse_out) *sse_out = rmse;
  if (cnt_out) *cnt_out = cnt;
  if (ub_out) *ub_out = uBiasReg;
  if (ib_out) *ib_out = iBiasReg;
  return rmse + uBiasReg * uReg + iBiasReg * iReg;
}
// model.cpp:2331-2362: the bias vectors are drawn from the SAME generator behind uFac and iFac
void orc_init_bias(int seed, int nU, int nI, int K, float* uBias, float* iBias) {
  std::default_random_engine generator(seed);
  float lb = -0.01, ub = 0.01;
  std::uniform_real_distribution<double> dist(lb, ub);
  for (int64_t t = 0; t < (int64_t)nU * K + (int64_t)nI * K; t++) (void)dist(generator);
  for (int u = 0; u < nU; u++) uBias[u] = dist(generator);
  for (int i = 0; i < nI; i++) iBias[i] = dist(generator);
}

