// ccd_blocks.h -- the CCD++ passes as a segmented reduction over LINE-ALIGNED blocks (round 4; the trip lists of rounds 2 and 3 are
// what this replaces).
//
// A pass forms, for every row (column), (num, den) = (sum res*o, sum o*o) over its entries, o = other[id] -- float products, double
// accumulation (modelMF.cpp:1069-1070, 1085-1086).  The trip lists gave every segment its own aligned 128-entry trips and masked what
// lay outside: a 208-entry row took 2.6 trips, i.e. the vector memory pipe moved 1.6 x the entries it needed (masked lanes cost the
// pipe what live ones do), the lines at segment borders were requested by both neighbours, and a third of every step's instructions
// were spent on masks.  Here the VIEW IS PADDED instead:
//   * every piece (a row of the row view; a (strip, column) piece of the column view) starts on a multiple of 8 entries -- the eight
//     consecutive entries ONE LANE loads -- and is filled up to the next multiple with entries whose residual is 0 and whose id is the
//     +0.0 slot behind the gathered vector: they add exact zeros (C4: 2 % more entries in the row view, 4 % in the column view);
//   * a region (the row view; one strip) is a whole number of 128-entry TRIPS, trip t = entries [128 t, 128 t + 128): sixteen lanes,
//     two 16-byte residual loads and one 16-byte load of eight 16-bit ids each.  Every line is loaded once, by one instruction, with all
//     lanes live, no compare and no select (the residuals of a trip are stored quad-interleaved for that: mfx_blk_mem_of);
//   * piece boundaries fall BETWEEN lanes.  A trip's record is (first slot, 16-bit mask of the lanes that end a piece; lane 15 always
//     set: what runs on into the next trip is a piece of its own).  The lane sums go through an inclusive scan over the sixteen lanes
//     (4 DPP levels); an end lane takes the prefix of the previous end lane (ds_bpermute) and stores the difference to
//     slot = first + popcount(mask below it).  A row's pieces are consecutive slots; the finishing kernels add them in order
//     (a fixed association) and divide.
// Trips carry no state from one to the next, so they can be dealt freely: a step of a workgroup is a CHUNK of 64 consecutive trips
// (group g: trip g of it -- 48 KB of consecutive lines), and the workgroups take the chunks of a region round-robin.
// The differences of prefixes are doubles over at most 128 float products: the cancellation error is 1e-16 of the trip's sum, nine
// orders below the float the quotient is rounded to.
#ifndef MFX_CCD_BLOCKS_H_
#define MFX_CCD_BLOCKS_H_

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "mfx_internal.h"

constexpr int MFX_BLK_EPL = 8;                     // entries per lane and trip
constexpr int MFX_BLK_E = 16 * MFX_BLK_EPL;        // entries per trip
constexpr int MFX_BLK_GPW = 64;                    // 16-lane groups per workgroup
// Entries that must be readable (valid ids, finite residuals) behind the last trip of an array: the loop prefetches two steps
// ahead and rounds a group's steps up to four, i.e. it reads at most 64 * 6 trips past the end of its window.
constexpr int64_t MFX_BLK_SLACK = (int64_t)MFX_BLK_GPW * 6 * MFX_BLK_E;

// Where the RESIDUAL of (logical) entry p of a padded view lies: inside every trip the sixteen lanes' first quads come first, then
// their second quads, so that each of a lane's two 16-byte loads is part of 256 contiguous bytes of its group.  Ids stay in entry
// order (a lane takes its eight with one load); the elementwise residual kernels walk the residuals in memory order and map back.
__host__ __device__ static inline int64_t mfx_blk_mem_of(int64_t p) {
  return (p & ~(int64_t)127) | (((p >> 2) & 1) << 6) | (((p >> 3) & 15) << 2) | (p & 3);
}
__host__ __device__ static inline int64_t mfx_blk_entry_of(int64_t t) {
  return (t & ~(int64_t)127) | (((t >> 2) & 15) << 3) | (((t >> 6) & 1) << 2) | (t & 3);
}
static inline void mfx_blocks_free(MfxBlocks& b) { dev_free(b.rec); dev_free(b.wg_t0); dev_free(b.wg_n); dev_free(b.wg_rec); dev_free(b.wg_tag); dev_free(b.wg_stride); b.nwg = 0; b.nslots = 0; }

struct MfxPiece { int64_t b, e; };               // padded positions [b, e): multiples of 8, ascending inside a region
struct MfxBlockPlan {
  std::vector<int2> rec;
  std::vector<int32_t> wg_t0, wg_n, wg_tag, wg_stride;
  std::vector<int64_t> wg_rec;
  int64_t nslots = 0;
};
__host__ __device__ static inline int mfx_blk_steps4(int wn) { return ((wn + MFX_BLK_GPW - 1) / MFX_BLK_GPW + 3) & ~3; }
// One region [r0, r1) (multiples of 128) with its pieces, dealt to nwg workgroups with (nearly) equal numbers of trips.
// first[k], cnt[k]: the slots of piece k (consecutive).  Returns false when a piece is not 8-aligned / out of order / outside.
static inline bool mfx_blocks_region(const MfxPiece* pc, size_t np, int64_t r0, int64_t r1, int nwg, int tag, MfxBlockPlan& plan,
                                     int32_t* first, int32_t* cnt) {
  if (r0 % MFX_BLK_E || r1 % MFX_BLK_E || r1 < r0) return false;
  const int64_t ntr = (r1 - r0) / MFX_BLK_E, T0 = r0 / MFX_BLK_E;
  if (ntr == 0) return np == 0;
  std::vector<uint16_t> mask((size_t)ntr, (uint16_t)0x8000);
  int64_t prev_e = r0;
  for (size_t k = 0; k < np; k++) {
    if (pc[k].b % MFX_BLK_EPL || pc[k].e % MFX_BLK_EPL || pc[k].b < prev_e || pc[k].e <= pc[k].b || pc[k].e > r1) return false;
    prev_e = pc[k].e;
    const int64_t endlane = pc[k].e / MFX_BLK_EPL - 1;
    mask[(size_t)((endlane >> 4) - T0)] |= (uint16_t)(1u << (endlane & 15));
  }
  std::vector<int64_t> base((size_t)ntr + 1);
  base[0] = plan.nslots;
  for (int64_t t = 0; t < ntr; t++) base[(size_t)t + 1] = base[(size_t)t] + __builtin_popcount(mask[(size_t)t]);
  if (base[(size_t)ntr] >= ((int64_t)1 << 28)) return false;        // slots are 16 bytes behind a 32-bit buffer offset
  for (size_t k = 0; k < np; k++) {
    const int64_t fl = pc[k].b / MFX_BLK_EPL, el = pc[k].e / MFX_BLK_EPL - 1;
    const int64_t tf = (fl >> 4) - T0;
    first[k] = (int32_t)(base[(size_t)tf] + __builtin_popcount(mask[(size_t)tf] & ((1u << (fl & 15)) - 1u)));
    cnt[k] = (int32_t)((el >> 4) - (fl >> 4) + 1);
  }
  plan.nslots = base[(size_t)ntr];
  // The workgroups take the region's CHUNKS of 64 trips (one step of a workgroup: 48 KB of consecutive lines) round-robin -- workgroup
  // w: chunks w, w + nwg, ... -- so that all of them read inside ONE window that moves through the region, like a grid-stride loop
  // (a contiguous window per workgroup meant 512 to 1 000 separate DRAM streams: 4.5 TB/s where the elementwise residual kernels,
  // which do walk one front, reach 5.7).  The loop addresses a region with 32-bit byte offsets: a region of 2^30 entries or more keeps
  // contiguous windows.
  const int64_t nch = (ntr + MFX_BLK_GPW - 1) / MFX_BLK_GPW;
  nwg = (int)std::max<int64_t>(1, std::min<int64_t>(nwg, nch));
  const bool interleave = (r1 - r0) + MFX_BLK_SLACK < ((int64_t)1 << 30) && !getenv("MFX_CCD_CONTIG");      // (the knob: tests)
  for (int w = 0; w < nwg; w++) {
    int64_t c0, nc, stride;             // first chunk, chunks, chunks between steps
    if (interleave) { c0 = w; nc = (nch - w + nwg - 1) / nwg; stride = nwg; }
    else { c0 = nch * w / nwg; nc = nch * (w + 1) / nwg - c0; stride = 1; }
    const int64_t last_chunk = c0 + stride * (nc - 1);
    const int last_trips = (int)std::min<int64_t>(MFX_BLK_GPW, ntr - last_chunk * MFX_BLK_GPW);
    const int wn = (int)((nc - 1) * MFX_BLK_GPW + last_trips), s4 = mfx_blk_steps4(wn);
    plan.wg_t0.push_back((int32_t)(T0 + c0 * MFX_BLK_GPW));
    plan.wg_n.push_back(wn);
    plan.wg_stride.push_back((int32_t)stride);
    plan.wg_tag.push_back(tag);
    plan.wg_rec.push_back((int64_t)plan.rec.size());
    for (int g = 0; g < MFX_BLK_GPW; g++)
      for (int i = 0; i < s4; i++) {
        const int64_t t = (c0 + stride * i) * MFX_BLK_GPW + g;
        int2 r;
        if (i < nc && t < ntr) { r.x = (int32_t)base[(size_t)t]; r.y = mask[(size_t)t]; }
        else { r.x = -1; r.y = 0; }
        plan.rec.push_back(r);
      }
  }
  return true;
}
static inline int mfx_blocks_upload(mfx_ctx* ctx, const MfxBlockPlan& p, MfxBlocks* out) {
  mfx_blocks_free(*out);
  int rc;
  if ((rc = dev_alloc(ctx, &out->rec, p.rec.size() + 16)) || (rc = dev_alloc(ctx, &out->wg_t0, p.wg_t0.size())) ||
      (rc = dev_alloc(ctx, &out->wg_n, p.wg_n.size())) || (rc = dev_alloc(ctx, &out->wg_rec, p.wg_rec.size())) ||
      (rc = dev_alloc(ctx, &out->wg_tag, p.wg_tag.size())) || (rc = dev_alloc(ctx, &out->wg_stride, p.wg_stride.size())))
    return rc;
  if (!p.rec.empty()) HIPCHK(hipMemcpy(out->rec, p.rec.data(), sizeof(int2) * p.rec.size(), hipMemcpyHostToDevice));
  if (!p.wg_t0.empty()) {
    HIPCHK(hipMemcpy(out->wg_t0, p.wg_t0.data(), sizeof(int32_t) * p.wg_t0.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out->wg_n, p.wg_n.data(), sizeof(int32_t) * p.wg_n.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out->wg_rec, p.wg_rec.data(), sizeof(int64_t) * p.wg_rec.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out->wg_tag, p.wg_tag.data(), sizeof(int32_t) * p.wg_tag.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out->wg_stride, p.wg_stride.data(), sizeof(int32_t) * p.wg_stride.size(), hipMemcpyHostToDevice));
  }
  out->nwg = (int)p.wg_t0.size();
  out->nslots = p.nslots;
  return MFX_OK;
}

// Diagnostic builds (results are WRONG on purpose): bit 0 no gathers, bit 1 float sums, bit 2 no scan / no lane exchange
#ifndef MFX_BLK_EXP
#define MFX_BLK_EXP 0
#endif
#ifdef __HIPCC__
// eight consecutive ids of a lane
template <typename IdxT> struct MfxIds8;
template <> struct MfxIds8<uint16_t> {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  struct raw { u4 a; };
  static __device__ __forceinline__ raw load(const uint16_t* p) { return raw{__builtin_nontemporal_load((const u4*)p)}; }
  static __device__ __forceinline__ int get(const raw& v, int q) { return (int)((q & 1) ? (v.a[q >> 1] >> 16) : (v.a[q >> 1] & 0xffffu)); }
};
template <> struct MfxIds8<int32_t> {
  typedef int i4 __attribute__((ext_vector_type(4)));
  struct raw { i4 a, b; };
  static __device__ __forceinline__ raw load(const int32_t* p) { return raw{*(const i4*)p, *(const i4*)(p + 4)}; }
  static __device__ __forceinline__ int get(const raw& v, int q) { return q < 4 ? v.a[q] : v.b[q - 4]; }
};

// row_shr:N inside the row of 16 lanes, lanes without a source read 0
template <int N>
__device__ __forceinline__ double mfx_blk_shr(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x110 + N, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x110 + N, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double mfx_blk_scan16(double v) {
  v += mfx_blk_shr<1>(v);
  v += mfx_blk_shr<2>(v);
  v += mfx_blk_shr<4>(v);
  v += mfx_blk_shr<8>(v);
  return v;
}
__device__ __forceinline__ double mfx_blk_from_lane(double x, int byte_addr) {
  const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(x));
  return __hiloint2double(hi, lo);
}
typedef int mfx_desc4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mfx_desc4 mfx_blk_desc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  return mfx_desc4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// The (num, den) pair of a piece, stored by inline assembly ON PURPOSE (the compiler's vmcnt counting does not see it: with a store
// pending it would turn every later wait for a load into vmcnt(0); loads complete in order among themselves, so its counting
// stays safe -- see DESIGN 3.4).  A lane without a piece passes an offset behind the buffer: the hardware drops the store.
// (s_nop 1: on gfx940 and later a store of more than 64 bits needs TWO wait states before a VALU may overwrite its data registers, and
// the hazard recognizer does not look inside an asm statement.  With `s_nop 0` -- enough before gfx940 -- the first version of this
// loop stored pairs whose low words had already been overwritten: 41 of 1 500 rows off by up to 13 ulp, tests/test_ccd_gpu.py.)
__device__ __forceinline__ void mfx_blk_store(mfx_desc4 rs, uint32_t off, double a, double b) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 v = {(unsigned)__double2loint(a), (unsigned)__double2hiint(a), (unsigned)__double2loint(b), (unsigned)__double2hiint(b)};
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(off), "s"(rs) : "memory");
}

// the updated residuals of a lane (fused sweep), unseen by the compiler's counting like the store above
__device__ __forceinline__ void mfx_blk_store_res(char* base, uint32_t off, float __attribute__((ext_vector_type(4))) a, float __attribute__((ext_vector_type(4))) b) {
  asm volatile("global_store_dwordx4 %0, %1, %3\n\tglobal_store_dwordx4 %0, %2, %3 offset:256\n\ts_nop 1" ::"v"(off), "v"(a), "v"(b), "s"(base) : "memory");
}

// One 16-lane group (lane j) of a workgroup's window -- group g takes the window's trips g, g + 64, ...  rec: the WORKGROUP's records
// (wave-uniform; group g's s4 records start at g * s4).
// res / ind: the padded view AT THE WINDOW'S FIRST TRIP (wave-uniform pointers: the loads take the scalar-base + 32-bit-offset form;
// a window is far below 4 GB); whole trips, MFX_BLK_SLACK valid entries behind the last.  other: the gathered vector with +0.0 at the
// index the padding entries carry.  part: [nslots][2] doubles.
// FUSE (the first sweep of a factor, ccd.hip: mfx_ccdpp_rank1): the residual update res = (res - a0 * o0[id]) + a1 * o1[id] -- the
// subtract of the finished factor and the add-back of the new one, modelMF.cpp:1095-1116 and :1032-1056, both roundings kept -- is
// applied to the entries on the way: they are stored back and the sums are formed over the NEW values.  (a0, a1) = pair[id8[lane]]: a
// lane's eight entries lie in one row / column (id8: the view's id per eight entries, at the window like res), pair[] holds both
// factors' values per row / column; the id is loaded two steps ahead with the data, the pair gathered one step ahead (a hit: ~26
// consecutive lanes share a row).  o0, o1: the two gathered vectors of the update.  Saves the update's own sweep over the view
// (C4: 0.6 GB read per view and factor).
template <typename IdxT, bool FUSE = false, typename Id8T = int32_t>
__device__ __forceinline__ void mfx_ccd_block_loop(const int2* __restrict__ rec, int wn, int wstride, int g, const float* __restrict__ res,
                                                   const IdxT* __restrict__ ind, const float* other, int j, double* __restrict__ part,
                                                   uint32_t part_bytes, const float* o0 = nullptr, const float* o1 = nullptr,
                                                   const Id8T* __restrict__ id8 = nullptr, const float2* __restrict__ pair = nullptr) {
  const int steps = (wn + MFX_BLK_GPW - 1) / MFX_BLK_GPW, s4 = mfx_blk_steps4(wn);   // of the workgroup (its last one is empty for groups >= wn % 64)
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef typename MfxIds8<IdxT>::raw raw_t;
  struct Data { f4 r0, r1; raw_t x; float2 ap; uint32_t tb; int id; };
  const mfx_desc4 rs_part = mfx_blk_desc(part, part_bytes);
  uint32_t eo = (uint32_t)(g * MFX_BLK_E + MFX_BLK_EPL * j);          // entry offset inside the window
  const uint32_t STRIDE = (uint32_t)MFX_BLK_GPW * MFX_BLK_E * (uint32_t)wstride;   // entries from one step of a group to its next
  // A step past the group's last trip (the prefetch of the step after the last; the empty last step) loads the group's LAST trip
  // again -- a hit.  Unclamped it read the first trips of the NEXT window, which that workgroup had loaded at its start and the
  // caches had long dropped: with the steps rounded up to four that was 15 % (rows) and 29 % (columns) more HBM traffic.
  const uint32_t eo_last = eo + STRIDE * (uint32_t)max((wn - g + MFX_BLK_GPW - 1) / MFX_BLK_GPW - 1, 0);
  auto data = [&]() {                                                 // the loads of the next step not yet requested
    Data d;
    if (FUSE) d.id = (int)id8[eo >> 3];       // FIRST: the next step gathers with it while the rest of this trip is still on its way
    // the residuals of a trip are stored QUAD-INTERLEAVED (mfx_blk_mem_of): the lanes' first quads, then their second quads -- each
    // load is 256 contiguous bytes per group.  (In entry order a lane's two quads are 32 bytes apart: both loads touched every
    // line of the trip, half of it each -- measured 11 % of the pass.)  nt: a stream, nothing of it is read twice.
    const uint32_t tb = (eo & ~127u) * 4u + 16u * (uint32_t)j;
    d.r0 = __builtin_nontemporal_load((const f4*)((const char*)res + (size_t)tb));
    d.r1 = __builtin_nontemporal_load((const f4*)((const char*)res + (size_t)tb + 256));
    d.x = MfxIds8<IdxT>::load((const IdxT*)((const char*)ind + (size_t)(eo * (uint32_t)sizeof(IdxT))));
    if (FUSE) d.tb = tb;
    eo = min(eo + STRIDE, eo_last);
    return d;
  };
  const unsigned below_me = (1u << j) - 1u;
  const int row_addr = (int)((threadIdx.x & 48u) << 2);             // byte address of lane 0 of this row for ds_bpermute
  double pnum = 0.0, pden = 0.0;
  uint32_t poff = 0xfffffff0u;                                      // pending store (none)
  // records: lane j & 3 of the group loads the record of step base + (j & 3); a step takes its own with a DPP row_newbcast
#define MFX_BLK_STEP(S, DC, DM, DN)                                                                                           \
  {                                                                                                                           \
    if (poff != 0xfffffff0u) mfx_blk_store(rs_part, poff, pnum, pden);   /* (EXEC-masked: an all-lanes store of mostly dropped lanes costs the memory pipe a full one) */ \
    DN = data();                                                                                                              \
    if (FUSE) DM.ap = pair[DM.id];                                      /* the NEXT step's pair: its id came in with the step before */ \
    const int base = __builtin_amdgcn_mov_dpp(ra.x, 0x150 + (S), 0xF, 0xF, true);                                             \
    const unsigned mask = (unsigned)__builtin_amdgcn_mov_dpp(ra.y, 0x150 + (S), 0xF, 0xF, true);                              \
    double num, den;                                                                                                          \
    if (FUSE) {                                                                                                               \
      f4 n0, n1;                                                                                                              \
      _Pragma("unroll") for (int q = 0; q < 8; q++) {                                                                         \
        const int id = MfxIds8<IdxT>::get(DC.x, q);                                                                           \
        const float rq = q < 4 ? DC.r0[q] : DC.r1[q - 4];                                                                     \
        const float nv = (rq - DC.ap.x * o0[id]) + DC.ap.y * o1[id];                                                          \
        if (q < 4) n0[q] = nv; else n1[q - 4] = nv;                                                                           \
      }                                                                                                                       \
      if (base >= 0) mfx_blk_store_res((char*)res, DC.tb, n0, n1);      /* (a step past the group's last trip holds that trip AGAIN) */ \
      DC.r0 = n0; DC.r1 = n1;                                                                                                 \
    }                                                                                                                         \
    {                                                                                                                         \
      float o[8];                                                                                                             \
      _Pragma("unroll") for (int q = 0; q < 8; q++) o[q] = (MFX_BLK_EXP & 1) ? __int_as_float(MfxIds8<IdxT>::get(DC.x, q) | 0x3f800000) : other[MfxIds8<IdxT>::get(DC.x, q)]; \
      if (MFX_BLK_EXP & 2) {                                                                                                  \
        float fn = DC.r0[0] * o[0], fd = o[0] * o[0];                                                                         \
        _Pragma("unroll") for (int q = 1; q < 8; q++) { fn += (q < 4 ? DC.r0[q] : DC.r1[q - 4]) * o[q]; fd += o[q] * o[q]; }  \
        num = (double)fn; den = (double)fd;                                                                                   \
      } else {                                                                                                                \
      num = (double)(DC.r0[0] * o[0]);                                                                                        \
      den = (double)(o[0] * o[0]);                                                                                            \
      _Pragma("unroll") for (int q = 1; q < 8; q++) {                                                                         \
        num += (double)((q < 4 ? DC.r0[q] : DC.r1[q - 4]) * o[q]);                                                            \
        den += (double)(o[q] * o[q]);                                                                                         \
      } }                                                                                                                     \
    }                                                                                                                         \
    if (!(MFX_BLK_EXP & 4)) { num = mfx_blk_scan16(num);                                                                      \
    den = mfx_blk_scan16(den); }                                                                                              \
    const unsigned below = mask & below_me;                                                                                   \
    const int prev = 31 - __builtin_clz(below | 1u);              /* below == 0: lane 0, not used */                           \
    const double qn = (MFX_BLK_EXP & 4) ? 0.0 : mfx_blk_from_lane(num, row_addr + 4 * prev), qd = (MFX_BLK_EXP & 4) ? 0.0 : mfx_blk_from_lane(den, row_addr + 4 * prev); \
    pnum = below ? num - qn : num;                                                                                            \
    pden = below ? den - qd : den;                                                                                            \
    poff = ((mask >> j) & 1u) ? (uint32_t)(base + __builtin_popcount(below)) * 16u : 0xfffffff0u;                             \
    __builtin_amdgcn_sched_barrier(0);   /* keep the steps apart: merged, the next step's unpacking is hoisted in front of this   \
                                            step's sums and the wave waits (vmcnt(0)) for the loads it has just issued */        \
  }
  uint32_t ro = (uint32_t)(g * s4 + (j & 3));
  int2 ra = rec[ro];
  // the loads run TWO steps ahead (four register sets rotating through the 4-step body).  With the steps kept apart by the
  // sched_barrier the plain loop needs 58 registers -- eight waves per SIMD; without it the compiler merged the steps, used 65 / 82
  // and waited for loads it had just issued.  (One step ahead was measured too: 5 % slower.)
  Data d0 = data(), d1 = data(), d2, d3;
  if (FUSE) d0.ap = pair[d0.id];
  int n = 0;
  for (; n + 4 <= steps; n += 4) {
    MFX_BLK_STEP(0, d0, d1, d2)
    MFX_BLK_STEP(1, d1, d2, d3)
    MFX_BLK_STEP(2, d2, d3, d0)
    ro += n + 4 < s4 ? 4u : 0u;
    const int2 rn = rec[ro];                                      // the next four records: a step (and seven other waves) ahead of their use
    MFX_BLK_STEP(3, d3, d0, d1)
    ra = rn;
  }
  if (n < steps) {                                                // the last one to three steps (workgroup-uniform branches)
    MFX_BLK_STEP(0, d0, d1, d2)
    if (n + 1 < steps) {
      MFX_BLK_STEP(1, d1, d2, d3)
      if (n + 2 < steps) MFX_BLK_STEP(2, d2, d3, d0)
    }
  }
  if (poff != 0xfffffff0u) mfx_blk_store(rs_part, poff, pnum, pden);
#undef MFX_BLK_STEP
}
#endif  // __HIPCC__

#endif
