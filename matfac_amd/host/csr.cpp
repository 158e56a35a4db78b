#include "csr.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <omp.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <string>
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

// ---------------------------------------------------------------------------
// Text CSR reader (the format gk_csr_Read(GK_CSR_FMT_CSR, readvals=1, numbering=0) takes from
// datastruct.cpp:13-15, 27-29, 62-64: one line per user, "item rating item rating ...", an empty line is a
// user without ratings, '%' starts a comment line).  The file is mapped, cut at line ends into one piece per
// thread and every piece parsed on its own; at C4/C5 sizes the serial reader was the longest step of a run.
// ---------------------------------------------------------------------------
namespace {
inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// Decimal -> float, correctly rounded (== strtof) or "don't know".  w * 10^k with w < 2^53 and |k| <= 22 is one
// correctly rounded double operation; narrowing it to float is right unless the double sits exactly on a
// float rounding midpoint, where the digits beyond the double decide -- that case goes to strtof.
bool fast_float(const char*& s, const char* e, float* out) {
  const char* p = s;
  bool neg = false;
  if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
  uint64_t w = 0;
  int nd = 0, k = 0;
  const char* d0 = p;
  while (p < e && *p >= '0' && *p <= '9') { if (w || *p != '0') { w = w * 10 + (uint64_t)(*p - '0'); nd++; } p++; }
  int digits = (int)(p - d0);
  if (p < e && *p == '.') {
    p++;
    const char* f0 = p;
    while (p < e && *p >= '0' && *p <= '9') { if (w || *p != '0') { w = w * 10 + (uint64_t)(*p - '0'); nd++; } k--; p++; }
    digits += (int)(p - f0);
  }
  if (digits == 0 || nd > 15) return false;
  if (p < e && (*p == 'e' || *p == 'E')) {
    const char* q = p + 1;
    bool eneg = false;
    if (q < e && (*q == '-' || *q == '+')) { eneg = *q == '-'; q++; }
    if (q < e && *q >= '0' && *q <= '9') {
      int ex = 0;
      while (q < e && *q >= '0' && *q <= '9') { if (ex < 10000) ex = ex * 10 + (*q - '0'); q++; }
      k += eneg ? -ex : ex;
      p = q;
    }
  }
  if (p < e && !is_ws(*p)) return false;            // "1.5x", "0x1p3", "nan(...)": let strtof decide
  double d;
  if (w == 0) d = 0.0;
  else if (k >= -22 && k <= 22) d = k < 0 ? (double)w / kPow10[-k] : (double)w * kPow10[k];
  else return false;
  if (w != 0) {
    if (d < 1e-30 || d > 1e30) return false;         // keep clear of float subnormals / overflow
    uint64_t bits;
    memcpy(&bits, &d, 8);
    if ((bits & 0x1fffffffULL) == 0x10000000ULL) return false;
  }
  const float f = (float)d;
  *out = neg ? -f : f;
  s = p;
  return true;
}

struct Piece {
  std::vector<int64_t> rowlen;
  std::vector<int32_t> ri;
  std::vector<float> rv;
  int64_t lines = 0, bad_line = 0;   // bad_line: 1-based inside the piece
  int32_t maxc = -1;
};

void parse_piece(const char* b, const char* e, Piece& P) {
  const char* s = b;
  std::string tmp;
  while (s < e) {
    const char* le = (const char*)memchr(s, '\n', (size_t)(e - s));
    const char* nx = le ? le + 1 : e;
    if (!le) le = e;
    P.lines++;
    if (s < le && *s == '%') { s = nx; continue; }
    int64_t n = 0;
    for (;;) {
      while (s < le && is_ws(*s)) s++;
      const char* t = s;
      bool neg = false;
      if (t < le && (*t == '-' || *t == '+')) { neg = *t == '-'; t++; }
      if (t >= le || *t < '0' || *t > '9') break;               // strtol found no number: the row ends here
      int64_t c = 0;
      while (t < le && *t >= '0' && *t <= '9') { if (c < ((int64_t)1 << 40)) c = c * 10 + (*t - '0'); t++; }
      s = t;
      while (s < le && is_ws(*s)) s++;
      float v;
      if (!fast_float(s, le, &v)) {
        tmp.assign(s, (size_t)(le - s));
        char* end;
        v = strtof(tmp.c_str(), &end);
        if (end == tmp.c_str()) { if (!P.bad_line) P.bad_line = P.lines; break; }
        s += end - tmp.c_str();
      }
      if (neg || c > 0x7ffffffe) { if (!P.bad_line) P.bad_line = P.lines; break; }
      P.ri.push_back((int32_t)c);
      P.rv.push_back(v);
      if ((int32_t)c > P.maxc) P.maxc = (int32_t)c;
      n++;
    }
    P.rowlen.push_back(n);
    s = nx;
  }
}
}  // namespace

csr_t* csr_read_text(const char* path, std::string* err) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) {
    if (err) *err = std::string("cannot open ") + path;
    return nullptr;
  }
  struct stat sb;
  if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
    close(fd);
    if (err) *err = std::string("cannot stat ") + path;
    return nullptr;
  }
  const size_t len = (size_t)sb.st_size;
  const char* base = nullptr;
  if (len) {
    base = (const char*)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (base == MAP_FAILED) {
      close(fd);
      if (err) *err = std::string("cannot map ") + path;
      return nullptr;
    }
  }
  close(fd);
  int T = std::max(1, omp_get_max_threads());
  T = (int)std::min<size_t>((size_t)T, len / (1 << 20) + 1);
  std::vector<size_t> cut((size_t)T + 1, len);
  cut[0] = 0;
  for (int t = 1; t < T; t++) {
    size_t c = std::max(len / T * t, cut[t - 1]);
    const char* nl = c < len ? (const char*)memchr(base + c, '\n', len - c) : nullptr;
    cut[t] = nl ? (size_t)(nl - base) + 1 : len;
  }
  std::vector<Piece> P((size_t)T);
#pragma omp parallel for schedule(static, 1) num_threads(T)
  for (int t = 0; t < T; t++) parse_piece(base + cut[t], base + cut[t + 1], P[t]);
  int64_t lines = 0;
  for (int t = 0; t < T; t++) {
    if (P[t].bad_line) {
      if (err) *err = std::string(path) + ": malformed line " + std::to_string(lines + P[t].bad_line);
      if (len) munmap((void*)base, len);
      return nullptr;
    }
    lines += P[t].lines;
  }
  std::vector<int64_t> row0((size_t)T + 1, 0), nz0((size_t)T + 1, 0);
  int32_t maxc = -1;
  for (int t = 0; t < T; t++) {
    row0[t + 1] = row0[t] + (int64_t)P[t].rowlen.size();
    nz0[t + 1] = nz0[t] + (int64_t)P[t].ri.size();
    maxc = std::max(maxc, P[t].maxc);
  }
  const int64_t nrows = row0[T], nnz = nz0[T];
  csr_t* m = new csr_t;
  m->nrows = (int32_t)nrows;
  m->ncols = maxc + 1;
  m->rowptr = (int64_t*)malloc(sizeof(int64_t) * ((size_t)nrows + 1));
  m->rowind = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
  m->rowval = (float*)malloc(sizeof(float) * (size_t)(nnz ? nnz : 1));
  m->rowptr[0] = 0;
#pragma omp parallel for schedule(static, 1) num_threads(T)
  for (int t = 0; t < T; t++) {
    int64_t e = nz0[t];
    for (size_t r = 0; r < P[t].rowlen.size(); r++) { e += P[t].rowlen[r]; m->rowptr[row0[t] + (int64_t)r + 1] = e; }
    if (!P[t].ri.empty()) {
      memcpy(m->rowind + nz0[t], P[t].ri.data(), sizeof(int32_t) * P[t].ri.size());
      memcpy(m->rowval + nz0[t], P[t].rv.data(), sizeof(float) * P[t].rv.size());
    }
  }
  if (len) munmap((void*)base, len);
  return m;
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
