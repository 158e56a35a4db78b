// synth.cpp -- synthetic rating matrices of the BASELINE.md shapes (host-side data prep).
//
// Plays the role of the reference's own synthetic workflow: ground-truth low-rank
// factors (python/genLatFacs.py:17-37), a CSR sampled from them (writeRandMatCSR,
// io.cpp:726-787) and the per-rating train/test/val colouring of writeTrainTestValMat
// (io.cpp:410-459), with the power-law user-degree / item-popularity skew SURVEY.md
// 8(d) prescribes (real MovieLens/Netflix files are not available offline).
//
// Deterministic in (shape, seed) and independent of the OpenMP thread count: every
// row draws from its own counter-based generator.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "mfhost.h"

namespace {
inline uint64_t splitmix(uint64_t& s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
inline uint64_t hash2(uint64_t a, uint64_t b) {
  uint64_t s = a * 0x9e3779b97f4a7c15ULL ^ (b + 0x632be59bd9b4e019ULL);
  return splitmix(s);
}
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() { return splitmix(s); }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
  double normal() {
    double u1 = uni(), u2 = uni();
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  }
};

struct Csr {
  int32_t nrows = 0, ncols = 0;
  std::vector<int64_t> rowptr;
  std::vector<int32_t> rowind;
  std::vector<float> rowval;
};

// Zipf-Mandelbrot weights (rank + r0)^-alpha over a random ranking of n objects.  The
// defaults in matfac_amd/synth.py are fitted to MovieLens-20M's published marginals
// (most-rated movie ~0.34 % of the ratings, median movie ~18 ratings; users 20..9254,
// median 68), so that the Hogwild conflict pattern is the real one.
std::vector<double> powerlaw(int64_t n, double alpha, double r0, uint64_t seed) {
  std::vector<std::pair<uint64_t, int64_t>> key(n);
  for (int64_t i = 0; i < n; i++) key[i] = {hash2(seed, (uint64_t)i), i};
  std::sort(key.begin(), key.end());
  std::vector<double> w(n);
  for (int64_t r = 0; r < n; r++) w[key[r].second] = 1.0 / std::pow((double)(r + 1) + r0, alpha);
  return w;
}
}  // namespace

struct mfh_synth {
  Csr mat[4];  // full, train, val, test
  int32_t nItems = 0;
};

extern "C" mfh_synth* mfh_synth_create(int32_t nU, int32_t nI, int64_t nnz, uint32_t seed, double alpha_u,
                                       double alpha_i, double noise, int32_t K0, double frac_train,
                                       double frac_val, uint32_t shard, double r0_u, double r0_i) {
  if (nU <= 0 || nI <= 1 || nnz < nU || K0 <= 0) return nullptr;
  const int64_t dmax = std::max<int64_t>(1, nI / 2);
  nnz = std::min<int64_t>(nnz, (int64_t)nU * dmax);
  mfh_synth* S = new mfh_synth;

  // ---- user degrees: floor + log-normal, exact total --------------------------
  // MovieLens-20M users: min 20, median 68, mean 144, max 9254  ==  20 + lognormal with
  // sigma ~1.38; alpha_u is that sigma, r0_u the floor as a fraction of the mean degree.
  const uint64_t useed = hash2(seed, 0x1000ULL + shard);  // per-user streams: (seed, shard)
  const double mean_deg = (double)nnz / (double)nU;
  const int64_t dmin = std::max<int64_t>(1, (int64_t)std::floor(r0_u * mean_deg));
  std::vector<double> wu(nU);
  double W = 0;
  for (int32_t u = 0; u < nU; u++) {
    Rng r(hash2(useed, 11ULL * 1315423911ULL + (uint64_t)u));
    wu[u] = std::exp(alpha_u * r.normal());
    W += wu[u];
  }
  std::vector<int64_t> deg(nU);
  int64_t tot = 0;
  const double spare = std::max(0.0, (double)nnz - (double)dmin * nU);
  for (int32_t u = 0; u < nU; u++) {
    int64_t d = dmin + (int64_t)std::floor(wu[u] / W * spare);
    d = std::min(std::max<int64_t>(d, 1), dmax);
    deg[u] = d;
    tot += d;
  }
  {  // fix the remainder in a fixed pseudo-random user order
    std::vector<std::pair<uint64_t, int32_t>> ord(nU);
    for (int32_t u = 0; u < nU; u++) ord[u] = {hash2(useed ^ 0x5bd1e995, (uint64_t)u), u};
    std::sort(ord.begin(), ord.end());
    while (tot != nnz) {
      bool moved = false;
      for (auto& o : ord) {
        if (tot == nnz) break;
        int32_t u = o.second;
        if (tot < nnz && deg[u] < dmax) { deg[u]++; tot++; moved = true; }
        else if (tot > nnz && deg[u] > 1) { deg[u]--; tot--; moved = true; }
      }
      if (!moved) break;
    }
  }
  // ---- item popularity cdf ---------------------------------------------------
  std::vector<double> wi = powerlaw(nI, alpha_i, r0_i, hash2(seed, 23));
  std::vector<double> cdf(nI);
  {
    double c = 0, Wi = 0;
    for (double x : wi) Wi += x;
    for (int32_t i = 0; i < nI; i++) { c += wi[i] / Wi; cdf[i] = c; }
    cdf[nI - 1] = 1.0;
  }
  // ---- ground-truth factors --------------------------------------------------
  const double sig = 1.0 / std::sqrt(std::sqrt((double)K0));
  std::vector<float> P((size_t)nU * K0), Q((size_t)nI * K0);
#pragma omp parallel for schedule(static)
  for (int32_t u = 0; u < nU; u++) {
    Rng r(hash2(useed ^ 0xA5A5A5A5ULL, (uint64_t)u));
    for (int k = 0; k < K0; k++) P[(size_t)u * K0 + k] = (float)(r.normal() * sig);
  }
#pragma omp parallel for schedule(static)
  for (int32_t i = 0; i < nI; i++) {
    Rng r(hash2(seed ^ 0x3C3C3C3CULL, (uint64_t)i));
    for (int k = 0; k < K0; k++) Q[(size_t)i * K0 + k] = (float)(r.normal() * sig);
  }

  Csr& F = S->mat[0];
  F.nrows = nU;
  F.ncols = nI;
  F.rowptr.assign((size_t)nU + 1, 0);
  for (int32_t u = 0; u < nU; u++) F.rowptr[u + 1] = F.rowptr[u] + deg[u];
  F.rowind.resize((size_t)tot);
  F.rowval.resize((size_t)tot);
  std::vector<int8_t> color((size_t)tot);

#pragma omp parallel
  {
    std::vector<uint8_t> seen((size_t)nI, 0);
    std::vector<std::pair<double, int32_t>> keys;
#pragma omp for schedule(dynamic, 64)
    for (int32_t u = 0; u < nU; u++) {
      Rng r(hash2(useed, 1000003ULL + (uint64_t)u));
      const int64_t d = deg[u];
      int32_t* out = F.rowind.data() + F.rowptr[u];
      if (d * 8 <= nI) {  // rejection sampling from the popularity cdf
        int64_t got = 0;
        while (got < d) {
          const double x = r.uni();
          int32_t it = (int32_t)(std::lower_bound(cdf.begin(), cdf.end(), x) - cdf.begin());
          if (it >= nI) it = nI - 1;
          if (!seen[it]) { seen[it] = 1; out[got++] = it; }
        }
        for (int64_t t = 0; t < d; t++) seen[out[t]] = 0;
      } else {  // weighted sampling without replacement (exponential keys), heavy rows only
        keys.resize((size_t)nI);
        for (int32_t i = 0; i < nI; i++) {
          double x = r.uni();
          if (x < 1e-300) x = 1e-300;
          keys[i] = {-std::log(x) / wi[i], i};
        }
        std::nth_element(keys.begin(), keys.begin() + (d - 1), keys.end());
        for (int64_t t = 0; t < d; t++) out[t] = keys[t].second;
      }
      std::sort(out, out + d);
      float* val = F.rowval.data() + F.rowptr[u];
      int8_t* col = color.data() + F.rowptr[u];
      for (int64_t t = 0; t < d; t++) {
        double dot = 0;
        for (int k = 0; k < K0; k++) dot += (double)P[(size_t)u * K0 + k] * (double)Q[(size_t)out[t] * K0 + k];
        double x = 3.5 + dot + noise * r.normal();
        x = std::round(x * 2.0) / 2.0;
        val[t] = (float)std::min(5.0, std::max(0.5, x));
        const double c = r.uni();
        col[t] = c < frac_train ? 0 : (c < frac_train + frac_val ? 1 : 2);
      }
      if (d > 0) col[0] = 0;  // every user keeps at least one train rating
    }
  }
  // ---- split -----------------------------------------------------------------
  for (int k = 0; k < 3; k++) {
    Csr& M = S->mat[1 + k];
    M.nrows = nU;
    M.rowptr.assign((size_t)nU + 1, 0);
    for (int32_t u = 0; u < nU; u++) {
      int64_t c = 0;
      for (int64_t e = F.rowptr[u]; e < F.rowptr[u + 1]; e++) c += color[e] == k;
      M.rowptr[u + 1] = M.rowptr[u] + c;
    }
    M.rowind.resize((size_t)M.rowptr[nU]);
    M.rowval.resize((size_t)M.rowptr[nU]);
    int32_t maxc = -1;
#pragma omp parallel for schedule(static) reduction(max : maxc)
    for (int32_t u = 0; u < nU; u++) {
      int64_t w = M.rowptr[u];
      for (int64_t e = F.rowptr[u]; e < F.rowptr[u + 1]; e++)
        if (color[e] == k) {
          M.rowind[w] = F.rowind[e];
          M.rowval[w] = F.rowval[e];
          if (F.rowind[e] > maxc) maxc = F.rowind[e];
          w++;
        }
    }
    M.ncols = maxc + 1;  // gk_csr_Read: ncols = max index + 1
    S->nItems = std::max(S->nItems, M.ncols);  // datastruct.cpp:91
  }
  return S;
}

extern "C" void mfh_synth_free(mfh_synth* s) { delete s; }

extern "C" int mfh_synth_shape(const mfh_synth* s, int which, int32_t* nrows, int32_t* ncols, int64_t* nnz) {
  if (!s || which < 0 || which > 3) return -1;
  const Csr& m = s->mat[which];
  if (nrows) *nrows = m.nrows;
  if (ncols) *ncols = m.ncols;
  if (nnz) *nnz = m.rowptr.empty() ? 0 : m.rowptr.back();
  return 0;
}

extern "C" int mfh_synth_copy(const mfh_synth* s, int which, int64_t* rowptr, int32_t* rowind, float* rowval) {
  if (!s || which < 0 || which > 3) return -1;
  const Csr& m = s->mat[which];
  if (rowptr) memcpy(rowptr, m.rowptr.data(), sizeof(int64_t) * m.rowptr.size());
  if (rowind) memcpy(rowind, m.rowind.data(), sizeof(int32_t) * m.rowind.size());
  if (rowval) memcpy(rowval, m.rowval.data(), sizeof(float) * m.rowval.size());
  return 0;
}

extern "C" int32_t mfh_synth_nitems(const mfh_synth* s) { return s ? s->nItems : 0; }
