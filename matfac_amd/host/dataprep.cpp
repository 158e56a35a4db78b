// dataprep.cpp -- train/test/val splitter and synthetic-matrix writer (reference: io.cpp:410-459, 726-787).
// Host-side preparation only; both are driven by std::mt19937 + std::uniform_int_distribution exactly as the reference
// drives them, so that with the same libstdc++ the same seed gives the same files.
#include "dataprep.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <random>
#include <unordered_set>

void trainTestValColors(int64_t nnz, float testPc, float valPc, int seed, int* color) {
  const int n = (int)nnz;
  const int nTest = testPc * n;          // float product truncated, as `int nTest = testPc * nnz` (:414)
  const int nVal = valPc * n;
  std::fill(color, color + n, 0);
  std::mt19937 mt(seed);
  std::uniform_int_distribution<int> nnzDist(0, n - 1);
  for (int i = 0; i < nTest; i++) color[nnzDist(mt)] = 1;   // with replacement: fewer than nTest distinct ratings
  for (int i = 0; i < nVal;) {
    const int k = nnzDist(mt);
    if (!color[k]) { color[k] = 2; i++; }
  }
}

csr_t* csr_take_color(const csr_t* mat, const int* color, int which) {
  std::vector<int64_t> ptr((size_t)mat->nrows + 1, 0);
  std::vector<int32_t> ind;
  std::vector<float> val;
  for (int32_t u = 0; u < mat->nrows; u++) {
    for (int64_t e = mat->rowptr[u]; e < mat->rowptr[u + 1]; e++)
      if (color[e] == which) { ind.push_back(mat->rowind[e]); val.push_back(mat->rowval[e]); }
    ptr[(size_t)u + 1] = (int64_t)ind.size();
  }
  return csr_from_arrays(mat->nrows, mat->ncols, ptr.data(), ind.data(), val.data());
}

int csr_write_text_gk(const csr_t* m, const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  for (int32_t u = 0; u < m->nrows; u++) {
    for (int64_t e = m->rowptr[u]; e < m->rowptr[u + 1]; e++) fprintf(f, " %d %f", m->rowind[e], m->rowval[e]);
    fputc('\n', f);
  }
  fclose(f);
  return 0;
}

void writeTrainTestValMat(csr_t* mat, const char* trainFileName, const char* testFileName, const char* valFileName,
                          float testPc, float valPc, int seed) {
  const int64_t nnz = mat->nnz();
  std::vector<int> color((size_t)std::max<int64_t>(nnz, 1));
  std::cout << "nTest: " << (int)(testPc * (int)nnz) << " nVal: " << (int)(valPc * (int)nnz) << std::endl;
  trainTestValColors(nnz, testPc, valPc, seed, color.data());
  std::cout << "Partitionining matrix..." << std::endl;
  const char* names[3] = {trainFileName, testFileName, valFileName};
  for (int c = 0; c < 3; c++) {
    csr_t* part = csr_take_color(mat, color.data(), c);
    if (csr_write_text_gk(part, names[c]) != 0) std::cerr << "writeTrainTestValMat: cannot write " << names[c] << std::endl;
    csr_free(&part);
  }
}

void writeRandMatCSR(const char* opFileName, std::vector<std::vector<double>>& uFac, std::vector<std::vector<double>>& iFac,
                     int facDim, int seed, int nnz) {
  const int nUsers = (int)uFac.size(), nItems = (int)iFac.size();
  std::vector<std::unordered_set<int>> uItemSet((size_t)nUsers);
  std::mt19937 mt(seed);
  std::uniform_int_distribution<int> uDist(0, nUsers - 1), iDist(0, nItems - 1);
  for (int u = 0; u < nUsers; u++) uItemSet[(size_t)u].insert(iDist(mt));              // every user rates something
  for (int item = 0; item < nItems; item++) uItemSet[(size_t)uDist(mt)].insert(item);  // every item is rated
  auto pairs = [&]() {
    int64_t c = 0;
    for (const auto& s : uItemSet) c += (int64_t)s.size();
    return c;
  };
  for (int64_t have = pairs(); have < nnz; have = pairs())
    for (int64_t i = 0, missing = nnz - have; i < missing; i++) {
      const int user = uDist(mt);        // user first, then item: the order the reference draws them in
      const int item = iDist(mt);
      uItemSet[(size_t)user].insert(item);
    }
  std::ofstream opFile(opFileName);
  if (!opFile.is_open()) return;
  for (int u = 0; u < nUsers; u++) {
    std::vector<int> items(uItemSet[(size_t)u].begin(), uItemSet[(size_t)u].end());
    std::sort(items.begin(), items.end());
    for (int item : items) {
      double r = 0;                      // dotProd (util.cpp): k ascending, double
      for (int k = 0; k < facDim; k++) r += uFac[(size_t)u][(size_t)k] * iFac[(size_t)item][(size_t)k];
      opFile << item << " " << r << " ";
    }
    opFile << std::endl;
  }
}
