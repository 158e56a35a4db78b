// Host-side check of the CCD++ trip lists (matfac_amd/csrc/mfx_internal.h: mfx_trips_layout / mfx_trips_append), compiled
// by tests/test_trips_cpu.py with hipcc and run on the CPU: no device code is launched.
//   usage: trips_check <E> <nwg> <gpw> < segments            (segments: lines "b e", memory order)
// Prints OK and statistics, or the first violated property.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mfx_internal.h"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const int E = atoi(argv[1]), nwg = atoi(argv[2]), gpw = atoi(argv[3]);
  std::vector<MfxSeg> segs;
  long long b, e;
  while (scanf("%lld %lld", &b, &e) == 2) segs.push_back(MfxSeg{b, e, (int32_t)segs.size()});
  std::vector<int4> trips;
  std::vector<int32_t> gptr;
  int64_t max_end = 0;
  mfx_trips_layout(segs, 0, segs.size(), nwg, gpw, E, trips, gptr, &max_end);
  gptr.push_back((int32_t)trips.size());
  // the allocation-pad invariant (mfx_trips_fit): the reported max_end is the true one, it lies less than E entries behind the last
  // segment, and arrays of 2- and 4-byte entries that end with the last segment hold every loaded trip inside their pad
  {
    long long n = 0, true_end = 0;
    for (const MfxSeg& g : segs) n = g.e > n ? g.e : n;
    for (const int4& r : trips) { const long long t0 = ((long long)(unsigned)r.x << 2); true_end = t0 + E > true_end ? t0 + E : true_end; }
    if (max_end != true_end) { printf("FAIL: max_end %lld, trips load up to %lld\n", (long long)max_end, true_end); return 1; }
    if (max_end - n >= E) { printf("FAIL: trips load %lld entries behind the last segment (E = %d)\n", (long long)max_end - n, E); return 1; }
    if (!mfx_trips_fit(max_end, n, 2) || !mfx_trips_fit(max_end, n, 4)) { printf("FAIL: a trip leaves the allocation pad (%lld behind %lld)\n", (long long)max_end - n, n); return 1; }
    if (mfx_trips_fit(n + (long long)MFX_ALLOC_PAD / 4 + 1, n, 4)) { printf("FAIL: mfx_trips_fit accepts a read behind the pad\n"); return 1; }
  }
  if ((int)gptr.size() != nwg * gpw + 1) { printf("FAIL: %zu range starts for %d groups\n", gptr.size() - 1, nwg * gpw); return 1; }
  std::vector<int> seen(segs.size(), 0);
  std::vector<long long> covered(segs.size(), 0);
  long long wg_min = 1LL << 60, wg_max = 0;
  for (int w = 0; w < nwg; w++) {
    long long wt = gptr[(size_t)(w + 1) * gpw] - gptr[(size_t)w * gpw];
    wg_min = wt < wg_min ? wt : wg_min;
    wg_max = wt > wg_max ? wt : wg_max;
  }
  for (size_t g = 0; g + 1 < gptr.size(); g++) {
    if (gptr[g] > gptr[g + 1]) { printf("FAIL: range %zu runs backwards\n", g); return 1; }
    int expect_i = 0, cur = -1;
    for (int n = gptr[g]; n < gptr[g + 1]; n++) {
      const int4 r = trips[(size_t)n];
      const int a = r.y & 31, i = (r.y >> 5) & 31, len = (r.y >> 10) & 0x7ff, last = (r.y & MFX_TRIP_LAST) != 0, s = r.z;
      if (s < 0 || s >= (int)segs.size()) { printf("FAIL: trip %d names segment %d\n", n, s); return 1; }
      if (i == 0) { if (cur >= 0) { printf("FAIL: segment %d not closed before %d\n", cur, s); return 1; } cur = s; expect_i = 0; seen[(size_t)s]++; }
      if (s != cur || i != expect_i) { printf("FAIL: trip %d: segment %d step %d, expected %d step %d\n", n, s, i, cur, expect_i); return 1; }
      const long long t0 = ((long long)(unsigned)r.x << 2), seg_b = segs[(size_t)s].b, seg_e = segs[(size_t)s].e;
      if (t0 % MFX_TRIP_ALIGN != 0) { printf("FAIL: trip %d starts at %lld, not a multiple of %d\n", n, t0, (int)MFX_TRIP_ALIGN); return 1; }
      if (len != seg_e - seg_b || t0 - (long long)E * i + a != seg_b) { printf("FAIL: trip %d does not decode to segment [%lld, %lld)\n", n, seg_b, seg_e); return 1; }
      // entries of this trip inside the segment: rel = E*i - a + p for p in [0, E)
      long long in = 0;
      for (int p = 0; p < E; p++) { const long long rel = (long long)E * i - a + p; if (rel >= 0 && rel < len) in++; }
      covered[(size_t)s] += in;
      const bool is_last = t0 + E >= seg_e;
      if (last != is_last) { printf("FAIL: trip %d last flag %d, expected %d\n", n, last, (int)is_last); return 1; }
      if (last) cur = -1;
      expect_i++;
    }
    if (cur >= 0) { printf("FAIL: range %zu ends inside segment %d\n", g, cur); return 1; }
  }
  for (size_t k = 0; k < segs.size(); k++) {
    if (seen[k] != 1) { printf("FAIL: segment %zu appears %d times\n", k, seen[k]); return 1; }
    if (covered[k] != segs[k].e - segs[k].b) { printf("FAIL: segment %zu: %lld of %lld entries covered\n", k, covered[k], segs[k].e - segs[k].b); return 1; }
  }
  printf("OK trips %zu segments %zu workgroup trips min %lld max %lld\n", trips.size(), segs.size(), wg_min, wg_max);
  return 0;
}
