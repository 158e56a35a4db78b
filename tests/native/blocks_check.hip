// Host-side check of the CCD++ block plan (matfac_amd/csrc/ccd_blocks.h: mfx_blocks_region), compiled by tests/test_trips_cpu.py
// with hipcc and run on the CPU: no device code is launched.  The device loop is restated here as plain loops: a live record
// (first slot, end mask) of trip t gives every end lane l the slot first + popcount(mask below l) and the lanes (previous end, l].
//   usage: blocks_check <r0> <r1> <nwg> < pieces            (pieces: lines "b e", padded positions, ascending)
// Prints OK and statistics, REFUSED when the plan does not lay out, or the first violated property.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ccd_blocks.h"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const long long r0 = atoll(argv[1]), r1 = atoll(argv[2]);
  const int nwg = atoi(argv[3]);
  std::vector<MfxPiece> pc;
  long long b, e;
  while (scanf("%lld %lld", &b, &e) == 2) pc.push_back(MfxPiece{b, e});
  MfxBlockPlan plan;
  plan.nslots = 7;                                  // a second region continues the slot numbering of the first
  std::vector<int32_t> first(pc.size()), cnt(pc.size());
  if (!mfx_blocks_region(pc.data(), pc.size(), r0, r1, nwg, 3, plan, first.data(), cnt.data())) { printf("REFUSED\n"); return 0; }
  const long long ntr = (r1 - r0) / MFX_BLK_E, T0 = r0 / MFX_BLK_E;
  // the residual order inside a trip is a bijection with its inverse, and a quad of entries stays a quad in memory
  for (long long p = 0; p < 512; p++) {
    const long long t = mfx_blk_mem_of(r0 + p);
    if (t / 128 != (r0 + p) / 128 || mfx_blk_entry_of(t) != r0 + p) { printf("FAIL: mfx_blk_mem_of / mfx_blk_entry_of at %lld\n", p); return 1; }
    if ((p & 3) == 0 && mfx_blk_mem_of(r0 + p + 3) != t + 3) { printf("FAIL: a quad is not contiguous in memory at %lld\n", p); return 1; }
  }
  // every trip of the region appears in exactly one live record; dead records are (-1, 0); slot -> lanes
  std::vector<int> seen((size_t)ntr, 0);
  std::vector<long long> slot_lo((size_t)plan.nslots, -1), slot_hi((size_t)plan.nslots, -1);   // global lane range of each slot
  long long wmin = 1LL << 60, wmax = 0;
  if (plan.wg_t0.size() != plan.wg_n.size() || plan.wg_t0.size() != plan.wg_rec.size() || plan.wg_t0.size() != plan.wg_stride.size()) { printf("FAIL: table sizes\n"); return 1; }
  for (size_t w = 0; w < plan.wg_t0.size(); w++) {
    const int wn = plan.wg_n[w], s4 = mfx_blk_steps4(wn), steps = (wn + MFX_BLK_GPW - 1) / MFX_BLK_GPW;
    if (plan.wg_tag[w] != 3) { printf("FAIL: tag\n"); return 1; }
    wmin = wn < wmin ? wn : wmin;
    wmax = wn > wmax ? wn : wmax;
    long long live = 0;
    for (int g = 0; g < MFX_BLK_GPW; g++)
      for (int i = 0; i < s4; i++) {
        const int2 r = plan.rec[(size_t)(plan.wg_rec[w] + (long long)g * s4 + i)];
        const long long t = (long long)plan.wg_t0[w] + g + (long long)MFX_BLK_GPW * plan.wg_stride[w] * i - T0;      // what the loop loads at this step
        const bool should = i < (wn - g + MFX_BLK_GPW - 1) / MFX_BLK_GPW;          // the loop's own count of a group's live steps
        if (!should) {
          if (r.x != -1 || r.y != 0) { printf("FAIL: record past a group's last trip is live (wg %zu group %d step %d)\n", w, g, i); return 1; }
          continue;
        }
        if (i >= steps) { printf("FAIL: live step beyond the workgroup's steps\n"); return 1; }
        if (t < 0 || t >= ntr) { printf("FAIL: live record for trip %lld outside the region\n", t); return 1; }
        if (r.x < 0 || !(r.y & 0x8000) || (r.y & ~0xffff)) { printf("FAIL: live record (%d, %x)\n", r.x, r.y); return 1; }
        seen[(size_t)t]++;
        live++;
        int prev = -1, rank = 0;
        for (int l = 0; l < 16; l++)
          if (r.y & (1 << l)) {
            const long long s = (long long)r.x + rank;
            if (s < 7 || s >= plan.nslots) { printf("FAIL: slot %lld outside [7, %lld)\n", s, (long long)plan.nslots); return 1; }
            if (slot_lo[(size_t)s] != -1) { printf("FAIL: slot %lld written twice\n", s); return 1; }
            slot_lo[(size_t)s] = (T0 + t) * 16 + prev + 1;
            slot_hi[(size_t)s] = (T0 + t) * 16 + l;
            prev = l;
            rank++;
          }
      }
    if (live != wn) { printf("FAIL: workgroup %zu has %lld live records for %d trips\n", w, live, wn); return 1; }
  }
  for (long long t = 0; t < ntr; t++)
    if (seen[(size_t)t] != 1) { printf("FAIL: trip %lld handled %d times\n", t, seen[(size_t)t]); return 1; }
  for (long long s = 7; s < plan.nslots; s++)
    if (slot_lo[(size_t)s] < 0) { printf("FAIL: slot %lld never written\n", s); return 1; }
  // a piece's slots are consecutive and cover exactly its lanes, in order
  for (size_t k = 0; k < pc.size(); k++) {
    long long lane = pc[k].b / MFX_BLK_EPL;
    for (int c = 0; c < cnt[k]; c++) {
      const long long s = (long long)first[k] + c;
      if (s < 7 || s >= plan.nslots || slot_lo[(size_t)s] != lane) { printf("FAIL: piece %zu slot %d does not start at lane %lld\n", k, c, lane); return 1; }
      lane = slot_hi[(size_t)s] + 1;
      if (c + 1 < cnt[k] && (lane & 15) != 0) { printf("FAIL: piece %zu continues inside a trip\n", k); return 1; }
    }
    if (lane != pc[k].e / MFX_BLK_EPL) { printf("FAIL: piece %zu covered to lane %lld, ends at %lld\n", k, lane, (long long)(pc[k].e / MFX_BLK_EPL)); return 1; }
  }
  printf("OK trips %lld pieces %zu slots %lld workgroups %zu trips min %lld max %lld\n", ntr, pc.size(), (long long)plan.nslots - 7, plan.wg_t0.size(), wmin, wmax);
  return 0;
}
