#include "csr.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

csr_t* csr_from_arrays(int32_t nrows, int32_t ncols, const int64_t* rowptr, const int32_t* rowind,
                       const float* rowval) {
  csr_t* m = new csr_t;
  m->nrows = nrows;
  m->ncols = ncols;
  const int64_t nnz = rowptr[nrows];
  m->rowptr = (int64_t*)malloc(sizeof(int64_t) * ((size_t)nrows + 1));
  m->rowind = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
  m->rowval = (float*)malloc(sizeof(float) * (size_t)(nnz ? nnz : 1));
  memcpy(m->rowptr, rowptr, sizeof(int64_t) * ((size_t)nrows + 1));
  if (nnz) {
    memcpy(m->rowind, rowind, sizeof(int32_t) * (size_t)nnz);
    memcpy(m->rowval, rowval, sizeof(float) * (size_t)nnz);
  }
  return m;
}

csr_t* csr_read_text(const char* path, std::string* err) {
  std::ifstream in(path);
  if (!in.is_open()) {
    if (err) *err = std::string("cannot open ") + path;
    return nullptr;
  }
  std::vector<int64_t> rp(1, 0);
  std::vector<int32_t> ri;
  std::vector<float> rv;
  std::string line;
  int32_t maxc = -1;
  int64_t lineno = 0;
  while (std::getline(in, line)) {
    lineno++;
    if (!line.empty() && line[0] == '%') continue;
    const char* s = line.c_str();
    char* end;
    for (;;) {
      const long c = strtol(s, &end, 10);
      if (end == s) break;
      s = end;
      const float v = strtof(s, &end);
      if (end == s || c < 0) {
        if (err) *err = std::string(path) + ": malformed line " + std::to_string(lineno);
        return nullptr;
      }
      s = end;
      ri.push_back((int32_t)c);
      rv.push_back(v);
      if (c > maxc) maxc = (int32_t)c;
    }
    rp.push_back((int64_t)ri.size());
  }
  return csr_from_arrays((int32_t)rp.size() - 1, maxc + 1, rp.data(), ri.data(), rv.data());
}

int csr_write_text(const csr_t* m, const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  for (int32_t u = 0; u < m->nrows; u++) {
    for (int64_t e = m->rowptr[u]; e < m->rowptr[u + 1]; e++)
      fprintf(f, e + 1 < m->rowptr[u + 1] ? "%d %.9g " : "%d %.9g", m->rowind[e], m->rowval[e]);
    fputc('\n', f);
  }
  fclose(f);
  return 0;
}

void csr_create_col_index(csr_t* m) {
  const int64_t nnz = m->nnz();
  free(m->colptr); free(m->colind); free(m->colval);
  m->colptr = (int64_t*)calloc((size_t)m->ncols + 1, sizeof(int64_t));
  m->colind = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
  m->colval = (float*)malloc(sizeof(float) * (size_t)(nnz ? nnz : 1));
  for (int64_t e = 0; e < nnz; e++) m->colptr[m->rowind[e] + 1]++;
  for (int32_t j = 0; j < m->ncols; j++) m->colptr[j + 1] += m->colptr[j];
  std::vector<int64_t> pos(m->colptr, m->colptr + m->ncols);
  for (int32_t u = 0; u < m->nrows; u++)
    for (int64_t e = m->rowptr[u]; e < m->rowptr[u + 1]; e++) {
      const int64_t d = pos[m->rowind[e]]++;
      m->colind[d] = u;
      m->colval[d] = m->rowval[e];
    }
}

csr_t* csr_dup(const csr_t* m) {
  csr_t* d = csr_from_arrays(m->nrows, m->ncols, m->rowptr, m->rowind, m->rowval);
  if (m->colptr) {
    const int64_t nnz = m->nnz();
    d->colptr = (int64_t*)malloc(sizeof(int64_t) * ((size_t)m->ncols + 1));
    d->colind = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    d->colval = (float*)malloc(sizeof(float) * (size_t)(nnz ? nnz : 1));
    memcpy(d->colptr, m->colptr, sizeof(int64_t) * ((size_t)m->ncols + 1));
    memcpy(d->colind, m->colind, sizeof(int32_t) * (size_t)nnz);
    memcpy(d->colval, m->colval, sizeof(float) * (size_t)nnz);
  }
  return d;
}

void csr_free(csr_t** pm) {
  if (!pm || !*pm) return;
  csr_t* m = *pm;
  free(m->rowptr); free(m->rowind); free(m->rowval);
  free(m->colptr); free(m->colind); free(m->colval);
  delete m;
  *pm = nullptr;
}
