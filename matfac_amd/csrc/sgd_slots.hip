// sgd_slots.hip -- MFX_SGD_TILED: XCD-tiled Hogwild with workgroup-owned item rows in LDS.
//
// Replaces the inner loops of ModelMF::train / hogTrain / trainSGDPar (modelMF.cpp:83-105,
// 1747-1763, 273-304) for the parallel default path.
//
// Two facts of the chip shape the schedule (measurements in DESIGN.md section 3.1):
//   (1) the 8 per-XCD L2 caches are not coherent with each other, so a row that is updated
//       from several XCDs in one launch exists in several diverging copies;
//   (2) lock-free updates of one row that are in flight together overwrite each other; with
//       ~10^4 ratings in flight that is every update of a popular item but one.
// Schedule: users and items are hashed into 8 blocks each (64 tiles).  In round r the
// workgroups that find themselves on XCD x (HW_REG_XCC_ID) take only tile (x, (x+r) mod 8):
// every user row is touched through one L2 per round (trainSGDPar's stratification with the
// chip's own 8 strata).  Inside a tile the ratings are grouped item-major into SLOTS of
// <= CAP_R ratings over <= ROWS items; a workgroup pulls a slot, stages the slot's item rows
// in LDS, visits the slot's ratings in a fresh pseudo-random order (Feistel permutation keyed by
// seed, epoch, slot) and applies the item-side update with an LDS atomic add on a fixed-point copy of the
// row -- no item update is lost and the 64 groups of the workgroup always read the freshest row.  The user
// side stays a lock-free read-modify-write through the XCD's L2 (sc1 loads bypass the CU's L1).
// An item with more than CAP_R/2 ratings in a tile gets ONE slot of its own, however long.
// Slots are pulled from per-tile counters and a final sweep launch drains whatever is left, so
// "every rating exactly once per epoch" holds for ANY workgroup->XCD placement.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sgd_common.h"
#include "sgd_slots.h"


static SlotState* state(mfx_ctx* ctx) { return (SlotState*)ctx->slots; }

void mfx_slots_free_internal(mfx_ctx* ctx) {
  SlotState* st = state(ctx);
  if (!st) return;
  for (SlotList& s : st->side) {
    dev_free(s.rec); dev_free(s.slot_beg); dev_free(s.slot_ibeg); dev_free(s.slot_items);
    dev_free(s.tile_slot); dev_free(s.ctr);
  }
  delete st;
  ctx->slots = nullptr;
}

// ---------------------------------------------------------------------------
// slot lists (once per train matrix and rank shape)
// ---------------------------------------------------------------------------
template <typename T>
static int up(mfx_ctx* ctx, T** dst, const std::vector<T>& v) {
  int rc = dev_alloc(ctx, dst, v.size());
  if (rc) return rc;
  if (!v.empty()) HIPCHK(hipMemcpyAsync(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, ctx->stream));
  return MFX_OK;
}

static int build_slots(mfx_ctx* ctx, SlotList* S, int rows, int side) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const int64_t nnz = m.nnz;
  // the device builder (setup.hip) makes the same lists; this host version is the readable statement of
  // them, the cross-check of tests/test_setup_gpu.py (MFX_SLOTS_HOST=1) and the path for nnz >= 2^31
  if (nnz > 0 && nnz < ((int64_t)1 << 31) && !getenv("MFX_SLOTS_HOST")) return mfx_slots_build_device(ctx, S, rows, side);
  std::vector<int32_t> ru((size_t)nnz), ri((size_t)nnz);
  std::vector<float> rv((size_t)nnz);
  if (nnz) {
    HIPCHK(hipMemcpy(ru.data(), m.rowid, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ri.data(), m.rowind, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rv.data(), m.rowval, sizeof(float) * (size_t)nnz, hipMemcpyDeviceToHost));
  }
  // own[] = index on the owned side (grouping key), oth[] = index on the lock-free side
  const std::vector<int32_t>& own = side == 0 ? ri : ru;
  const std::vector<int32_t>& oth = side == 0 ? ru : ri;
  const int32_t nown = side == 0 ? m.ncols : m.nrows;
  // ratings of each tile (stable in CSR order)
  std::vector<int64_t> tstart(NTILE + 1, 0);
  for (int64_t e = 0; e < nnz; e++) tstart[slot_user_block(ru[e]) * 8 + mfx_item_block(ri[e]) + 1]++;
  for (int t = 0; t < NTILE; t++) tstart[t + 1] += tstart[t];
  std::vector<int64_t> byt((size_t)nnz);
  {
    std::vector<int64_t> pos(tstart.begin(), tstart.end() - 1);
    for (int64_t e = 0; e < nnz; e++) byt[pos[slot_user_block(ru[e]) * 8 + mfx_item_block(ri[e])]++] = e;
  }
  std::vector<int32_t> rec((size_t)nnz * 4), slot_ibeg(1, 0), slot_items, tile_slot(NTILE + 1, 0);
  std::vector<int64_t> slot_beg(1, 0);
  int32_t nslots = 0;
  std::vector<int32_t> cnt((size_t)std::max(nown, 1), 0);
  std::vector<int32_t> items;
  int64_t out = 0;
  for (int t = 0; t < NTILE; t++) {
    tile_slot[t] = nslots;
    const int64_t b = tstart[t], e = tstart[t + 1];
    items.clear();
    for (int64_t x = b; x < e; x++) {
      const int32_t it = own[byt[x]];
      if (cnt[it]++ == 0) items.push_back(it);
    }
    // items by descending number of ratings in this tile, ties by id
    std::sort(items.begin(), items.end(), [&](int32_t a, int32_t c) { return cnt[a] != cnt[c] ? cnt[a] > cnt[c] : a < c; });
    // ratings item-major: offsets per item, then a stable scatter
    std::vector<int64_t> off(items.size() + 1, 0);
    for (size_t k = 0; k < items.size(); k++) off[k + 1] = off[k] + cnt[items[k]];
    std::vector<int64_t> sorted((size_t)(e - b));
    {
      std::vector<int64_t> where((size_t)items.size());
      for (size_t k = 0; k < items.size(); k++) { where[k] = off[k]; cnt[items[k]] = (int32_t)k; }  // cnt := rank
      for (int64_t x = b; x < e; x++) sorted[where[cnt[own[byt[x]]]]++] = byt[x];
    }
    // cut into slots
    int cur_r = 0, cur_i = 0;
    auto close = [&]() {
      if (cur_r == 0) return;
      slot_beg.push_back(out);
      slot_ibeg.push_back((int32_t)slot_items.size());
      nslots++;
      cur_r = cur_i = 0;
    };
    for (size_t k = 0; k < items.size(); k++) {
      const int64_t n = off[k + 1] - off[k];
      const int32_t it = items[k];
      if (n > CAP_R / 2) {
        // popular in this tile: ONE slot of its own, however long.  (Cutting it into several
        // concurrently processed slots and summing their deltas overshoots: measured NaN in the
        // first epoch -- each replica makes the whole step.  One owner keeps it sequential; the
        // slot list is longest-first so this pole starts first.)
        close();
        slot_items.push_back(it);
        for (int64_t x = 0; x < n; x++) {
          const int64_t src = sorted[off[k] + x];
          rec[4 * out] = oth[src]; rec[4 * out + 1] = 0; memcpy(&rec[4 * out + 2], &rv[src], 4); rec[4 * out + 3] = it;
          out++;
        }
        cur_r = (int)std::min<int64_t>(n, 1 << 30); cur_i = 1;
        close();
      } else {
        if (cur_r + n > CAP_R || cur_i == rows) close();
        slot_items.push_back(it);
        for (int64_t x = 0; x < n; x++) {
          const int64_t src = sorted[off[k] + x];
          rec[4 * out] = oth[src]; rec[4 * out + 1] = cur_i; memcpy(&rec[4 * out + 2], &rv[src], 4); rec[4 * out + 3] = it;
          out++;
        }
        cur_r += (int)n; cur_i++;
      }
    }
    close();
    for (int32_t it : items) cnt[it] = 0;
  }
  tile_slot[NTILE] = nslots;
  dev_free(S->rec); dev_free(S->slot_beg); dev_free(S->slot_ibeg); dev_free(S->slot_items);
  dev_free(S->tile_slot);
  int rc;
  if ((rc = up(ctx, &S->rec, rec))) return rc;
  if ((rc = up(ctx, &S->slot_beg, slot_beg))) return rc;
  if ((rc = up(ctx, &S->slot_ibeg, slot_ibeg))) return rc;
  if ((rc = up(ctx, &S->slot_items, slot_items))) return rc;
  if ((rc = up(ctx, &S->tile_slot, tile_slot))) return rc;
  if (!S->ctr && (rc = dev_alloc(ctx, &S->ctr, (size_t)NTILE))) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (getenv("MFX_DEBUG")) {
    int64_t mx = 0, small = 0;
    for (size_t k = 0; k + 1 < slot_beg.size(); k++) {
      mx = std::max(mx, slot_beg[k + 1] - slot_beg[k]);
      small += (slot_beg[k + 1] - slot_beg[k]) < 256;
    }
    fprintf(stderr, "[mfx] slots (%s rows owned): %zu for %lld ratings, longest %lld, <256 ratings: %lld, row refs %zu\n",
            side == 0 ? "item" : "user", (size_t)nslots, (long long)nnz, (long long)mx, (long long)small, slot_items.size());
  }
  S->nslots = nslots;
  S->nnz = nnz;
  S->rows = rows;
  S->built = true;
  return MFX_OK;
}

// ---------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------
// owned rows per slot: 16 KB of LDS per workgroup, at most 64 and at least 8 rows
template <int LD>
struct SlotRows { static constexpr int value = 4096 / LD > 64 ? 64 : (4096 / LD < 8 ? 8 : 4096 / LD); };

__device__ __forceinline__ int64_t slot_perm(int64_t t, int64_t R, uint32_t k0, uint32_t k1) {
  int bits = 2;
  while (((int64_t)1 << bits) < R) bits++;
  const int ab = bits / 2;
  return mfx_perm_index(t, R, ab, bits - ab, k0, k1);
}

// Item rows of a slot live in LDS as Q = round(q * 2^24) (int32).  A visit reads q = float(Q) * 2^-24,
// computes q_new with the reference's arithmetic and adds round((q_new - q) * 2^24) with ds_add_u32:
// no item update is lost and every group reads the freshest row.  (ds_add_f32 was measured 7x slower
// than the whole rest of the kernel; the integer add is free.)  Resolution 6e-8, range +-128; a slot
// whose staged rows are not finite or exceed the range falls back to float rows with plain stores, and a
// row that leaves the range while being updated is written back as NaN, so Model::isTerminateModel's NaN
// guard (model.cpp:1486-1498) still sees a diverged model.
constexpr float FIX_SCALE = 16777216.0f, FIX_INV = 1.0f / 16777216.0f, FIX_MAX = 127.0f;

// entry S*G+g of the chunk: row_share broadcast of the transposed chunk (L == 16) or a cross-lane read
template <int L, int S>
__device__ __forceinline__ int slot_take(int v, int g) {
  if constexpr (L == 16) return __builtin_amdgcn_update_dpp(0, v, 0x150 + S, 0xF, 0xF, false);   // row_share:S
  else return __shfl(v, S * (64 / L) + g, 64);
}

// The L steps of one 64-rating chunk, unrolled by template recursion (the DPP controls are immediates).
// FIX: the slot's rows are in the fixed-point representation (decided per slot, so per chunk it is a template
// argument, not a select per element).
template <int L, int C, int ARITH, bool OWN_U, bool FIX, int S>
struct SlotSteps {
  static __device__ __forceinline__ void run(const Rows<3>& Um, int* q_lds, int tx, int ty, int tz, int g, int j,
                                             int nvalid, float lr, float uReg, float iReg,
                                             float4v (&pn)[C], int64_t& pen, float4v (&pnn)[C], int64_t& penn) {
    constexpr int G = 64 / L;
    constexpr int LD = 4 * L * C;
    const int e = S * G + g;
    const int li = slot_take<L, S>(ty, g);
    const float r = __builtin_bit_cast(float, slot_take<L, S>(tz, g));
    // rows of steps S+1 and S+2 are already requested; take S, shift, request S+2
    float4v p[C];
    const int64_t pe = pen;
#pragma unroll
    for (int c = 0; c < C; c++) { p[c] = pn[c]; pn[c] = pnn[c]; }
    pen = penn;
    if constexpr (S + 2 < L) {
      const int un = slot_take<L, S + 2>(tx, g);
      if (e + 2 * G < nvalid) {
        penn = (int64_t)un * LD + 4 * j;
#pragma unroll
        for (int c = 0; c < C; c++) pnn[c] = Um.ld(penn + c * 4 * L);
      }
    }
    if (e < nvalid) {
      int* qrow = q_lds + li * LD + 4 * j;
      float4v q[C];
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int4 qi = *(const int4*)(qrow + c * 4 * L);
        if (FIX) q[c] = float4v{(float)qi.x * FIX_INV, (float)qi.y * FIX_INV, (float)qi.z * FIX_INV, (float)qi.w * FIX_INV};
        else q[c] = __builtin_bit_cast(float4v, qi);
      }
      // p = the row from global memory, q = the owned row; the reference updates the USER row first
      // and the item row with the updated user row (modelMF.cpp:94-103) whichever side is owned
      const float est = group_dot<L, C>(p, q);
      if constexpr (FIX && !OWN_U && ARITH == MFX_ARITH_F32) {
        // hogTrain's arithmetic (modelMF.cpp:1755-1762) with the item step taken as a delta: the new item
        // row would be q - t, t = lr*(c1*p' + ci*q); the owner copy receives round(-t * 2^24) directly
        const float c1 = -2.0f * (r - est), cu = 2.0f * uReg, ci = 2.0f * iReg;
#pragma unroll
        for (int c = 0; c < C; c++) {
#pragma unroll
          for (int x = 0; x < 4; x++) p[c][x] = upd_f32(p[c][x], q[c][x], c1, cu, lr);
          Um.st(pe + c * 4 * L, p[c]);
#pragma unroll
          for (int x = 0; x < 4; x++) {
            const float t = lr * (c1 * p[c][x] + ci * q[c][x]);
            atomicAdd(qrow + c * 4 * L + x, __float2int_rn(t * -FIX_SCALE));   // ds_add_u32
          }
        }
      } else {
        float4v q0[C];
#pragma unroll
        for (int c = 0; c < C; c++) q0[c] = q[c];
        if (OWN_U) sgd_axpys<C, ARITH>(q, p, r, est, lr, uReg, iReg);
        else sgd_axpys<C, ARITH>(p, q, r, est, lr, uReg, iReg);
#pragma unroll
        for (int c = 0; c < C; c++) {
          Um.st(pe + c * 4 * L, p[c]);
          if (FIX) {
#pragma unroll
            for (int x = 0; x < 4; x++)
              atomicAdd(qrow + c * 4 * L + x, __float2int_rn((q[c][x] - q0[c][x]) * FIX_SCALE));   // ds_add_u32
          } else {
            *(int4*)(qrow + c * 4 * L) = __builtin_bit_cast(int4, q[c]);
          }
        }
      }
    }
    if constexpr (S + 1 < L)
      SlotSteps<L, C, ARITH, OWN_U, FIX, S + 1>::run(Um, q_lds, tx, ty, tz, g, j, nvalid, lr, uReg, iReg, pn, pen, pnn, penn);
  }
};

// C = 1 (K <= 64): 2 workgroups per CU (<= 64 VGPRs); wider ranks keep 1 workgroup per CU and get 128 VGPRs
template <int L, int C, int ARITH, bool SWEEP, bool OWN_U>
__global__ __launch_bounds__(WG, (4 * L * C <= 64 ? 8 : 4)) void sgd_slots_kernel(const int4* __restrict__ rec,
                                                           const int64_t* __restrict__ slot_beg,
                                                           const int32_t* __restrict__ slot_ibeg,
                                                           const int32_t* __restrict__ slot_items,
                                                           const int32_t* __restrict__ tile_slot, unsigned* ctr,
                                                           int round, float* Oth, float* Own, uint32_t obytes,
                                                           float lr, float uReg, float iReg, uint32_t k0,
                                                           uint32_t k1) {
  constexpr int G = 64 / L;
  constexpr int LD = 4 * L * C;
  constexpr int LD4 = LD / 4;
  constexpr int ROWS = SlotRows<4 * L * C>::value;
  __shared__ __attribute__((aligned(16))) int q_lds[ROWS * LD];
  __shared__ int s_slot, s_bad;
  // Oth: the lock-free side (user rows when item rows are owned, and vice versa); Own: staged in LDS
  const Rows<3> Um(Oth, obytes);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane / L, j = lane % L;
  const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7);  // HW_REG_XCC_ID[3:0]
  // round r: XCD x takes user block x*SUB + r%SUB and item block (x + r/SUB) mod 8
  const int t_first = SWEEP ? 0 : (xcc * SUB + round % SUB) * 8 + ((xcc + round / SUB) & 7);
  const int t_last = SWEEP ? NTILE - 1 : t_first;
  int4* q4 = (int4*)q_lds;
  for (int tile = t_first; tile <= t_last; tile++) {
    const int s0 = tile_slot[tile], ns = tile_slot[tile + 1] - s0;
    if (SWEEP) {  // nothing left in this tile (the normal case): do not queue on its counter
      if (tid == 0) s_slot = (int)__hip_atomic_load(&ctr[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int seen = s_slot;
      __syncthreads();
      if (seen >= ns) continue;
    }
    for (;;) {
      if (tid == 0) { s_slot = (int)atomicAdd(&ctr[tile], 1u); s_bad = 0; }
      __syncthreads();
      const int sl = s_slot;
      __syncthreads();
      if (sl >= ns) break;
      const int slot = s0 + sl;
      const int64_t rb = slot_beg[slot], R = slot_beg[slot + 1] - rb;
      const int ib = slot_ibeg[slot], ni = slot_ibeg[slot + 1] - ib;
      // stage the slot's item rows (as floats first, to decide the representation)
      bool mybad = false;
      for (int x = tid; x < ni * LD4; x += WG) {
        const int row = x / LD4, c4 = x % LD4;
        const float4v v = *(const float4v*)(Own + (int64_t)slot_items[ib + row] * LD + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; e++) mybad |= !(__builtin_fabsf(v[e]) <= FIX_MAX);   // also true for NaN
        q4[row * LD4 + c4] = __builtin_bit_cast(int4, v);
      }
      if (mybad) s_bad = 1;
      __syncthreads();
      const bool fix = s_bad == 0;
      if (fix) {
        for (int x = tid; x < ni * LD4; x += WG) {
          const float4v v = __builtin_bit_cast(float4v, q4[x]);
          q4[x] = make_int4(__float2int_rn(v.x * FIX_SCALE), __float2int_rn(v.y * FIX_SCALE),
                            __float2int_rn(v.z * FIX_SCALE), __float2int_rn(v.w * FIX_SCALE));
        }
      }
      __syncthreads();
      const uint32_t ks0 = k0 + (uint32_t)slot * 0x632be5abU, ks1 = k1 ^ mfx_mix32((uint32_t)slot + 77U);
      for (int64_t cb = (int64_t)wave * 64; cb < R; cb += (WG / 64) * 64) {
        const int64_t t = cb + lane;
        const bool ok = t < R;
        int4 rc4 = make_int4(0, 0, 0, 0);
        if (ok) rc4 = rec[rb + slot_perm(t, R, ks0, ks1)];
        const int nvalid = (int)(R - cb < 64 ? R - cb : 64);
        // L == 16: a group is one DPP row.  Transpose the chunk once (3 ds_bpermute) so that lane s of row g
        // holds entry s*G+g; every step then takes its entry with a row_share broadcast (no LDS traffic).
        int tx = rc4.x, ty = rc4.y, tz = rc4.z;
        if (L == 16) {
          const int src = (lane & 15) * G + (lane >> 4);
          tx = __shfl(rc4.x, src, 64); ty = __shfl(rc4.y, src, 64); tz = __shfl(rc4.z, src, 64);
        }
        // software pipeline: the lock-free rows of steps s+1 and s+2 are requested before step s is computed
        float4v pn[C], pnn[C];
        int64_t pen = (int64_t)slot_take<L, 0>(tx, g) * LD + 4 * j;
        int64_t penn = (int64_t)slot_take<L, 1>(tx, g) * LD + 4 * j;
        if (g < nvalid) {
#pragma unroll
          for (int c = 0; c < C; c++) pn[c] = Um.ld(pen + c * 4 * L);
        }
        if (G + g < nvalid) {
#pragma unroll
          for (int c = 0; c < C; c++) pnn[c] = Um.ld(penn + c * 4 * L);
        }
        if (fix) SlotSteps<L, C, ARITH, OWN_U, true, 0>::run(Um, q_lds, tx, ty, tz, g, j, nvalid, lr, uReg, iReg, pn, pen, pnn, penn);
        else SlotSteps<L, C, ARITH, OWN_U, false, 0>::run(Um, q_lds, tx, ty, tz, g, j, nvalid, lr, uReg, iReg, pn, pen, pnn, penn);
      }
      __syncthreads();
      // write the item rows back (this workgroup is their only owner during the round)
      for (int x = tid; x < ni * LD4; x += WG) {
        const int row = x / LD4, c4 = x % LD4;
        const int4 qi = q4[x];
        float* vrow = Own + (int64_t)slot_items[ib + row] * LD + 4 * c4;
        float4v v;
        if (fix) {
          // new row = staged fp32 row + the accumulated fixed-point delta: an untouched row is
          // written back bit for bit, a touched one carries only the rounding of its updates
          const float4v v0 = *(const float4v*)vrow;
          const int lim = 0x7f000000;   // |q| beyond ~127: the row left the fixed-point range => diverged
          const float nanv = __builtin_nanf("");
          const int qa[4] = {qi.x, qi.y, qi.z, qi.w};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int q0i = __float2int_rn(v0[e] * FIX_SCALE);
            v[e] = (qa[e] > lim || qa[e] < -lim) ? nanv : v0[e] + (float)(qa[e] - q0i) * FIX_INV;
          }
        } else {
          v = __builtin_bit_cast(float4v, qi);
        }
        *(float4v*)vrow = v;
      }
      __syncthreads();
    }
  }
}

// test hook: materialise the visiting order of the last epoch (slot by slot)
__global__ void slots_epoch_list_kernel(const int4* __restrict__ rec, const int64_t* __restrict__ slot_beg,
                                        int64_t nslots, int own_u, uint32_t k0, uint32_t k1, int32_t* __restrict__ eu,
                                        int32_t* __restrict__ ei, float* __restrict__ er) {
  for (int64_t slot = blockIdx.x; slot < nslots; slot += gridDim.x) {
    const int64_t rb = slot_beg[slot], R = slot_beg[slot + 1] - rb;
    const uint32_t ks0 = k0 + (uint32_t)slot * 0x632be5abU, ks1 = k1 ^ mfx_mix32((uint32_t)slot + 77U);
    for (int64_t t = threadIdx.x; t < R; t += blockDim.x) {
      const int4 r = rec[rb + slot_perm(t, R, ks0, ks1)];
      eu[rb + t] = own_u ? r.w : r.x; ei[rb + t] = own_u ? r.x : r.w; er[rb + t] = __builtin_bit_cast(float, r.z);
    }
  }
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------
static int env_blocks() {
  static int b = -1;
  if (b < 0) { const char* e = getenv("MFX_SGD_BLOCKS"); b = e ? atoi(e) : 0; if (b < 0) b = 0; }
  return b;
}

template <int L, int C, int ARITH, bool OWN_U>
static int launch_slots(mfx_ctx* ctx, SlotList* S, const mfx_sgd_opts* o, int blocks, uint32_t k0, uint32_t k1) {
  float* oth = OWN_U ? ctx->V : ctx->U;
  float* own = OWN_U ? ctx->U : ctx->V;
  const uint64_t ob = (uint64_t)(OWN_U ? ctx->nI : ctx->nU) * ctx->ld * 4;
  HIPCHK(hipMemsetAsync(S->ctr, 0, NTILE * sizeof(unsigned), ctx->stream));
  for (int round = 0; round < NUB; round++) {
    ProfScope ps(ctx, MFX_K_SGD);
    hipLaunchKernelGGL((sgd_slots_kernel<L, C, ARITH, false, OWN_U>), dim3(blocks), dim3(WG), 0, ctx->stream,
                       (const int4*)S->rec, S->slot_beg, S->slot_ibeg, S->slot_items, S->tile_slot, S->ctr, round,
                       oth, own, (uint32_t)ob, o->learnRate, o->uReg, o->iReg, k0, k1);
  }
  {
    ProfScope ps(ctx, MFX_K_SGD_SWEEP);
    hipLaunchKernelGGL((sgd_slots_kernel<L, C, ARITH, true, OWN_U>), dim3(256), dim3(WG), 0, ctx->stream,
                       (const int4*)S->rec, S->slot_beg, S->slot_ibeg, S->slot_items, S->tile_slot, S->ctr, -1,
                       oth, own, (uint32_t)ob, o->learnRate, o->uReg, o->iReg, k0, k1);
  }
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

template <int L, int C>
static int launch_arith(mfx_ctx* ctx, SlotList* S, int side, const mfx_sgd_opts* o, int blocks, uint32_t k0,
                        uint32_t k1) {
#define MFX_SIDE(A) \
  (side ? launch_slots<L, C, A, true>(ctx, S, o, blocks, k0, k1) : launch_slots<L, C, A, false>(ctx, S, o, blocks, k0, k1))
  switch (o->arith) {
    case MFX_ARITH_REF64: return MFX_SIDE(MFX_ARITH_REF64);
    case MFX_ARITH_REF64F: return MFX_SIDE(MFX_ARITH_REF64F);
    default: return MFX_SIDE(MFX_ARITH_F32);
  }
#undef MFX_SIDE
}

int mfx_launch_sgd_tiled(mfx_ctx* ctx, const mfx_sgd_opts* o) {
  // which side is owned this epoch: o->own = 0 item rows (default), 1 user rows, 2 alternate
  const int side = o->own == 1 ? 1 : o->own == 2 ? (o->epoch & 1) : 0;
  const uint64_t ob = (uint64_t)(side ? ctx->nI : ctx->nU) * ctx->ld * 4;
  NEED(ob < (1ull << 32), MFX_E_ARG, "MFX_SGD_TILED: a factor matrix exceeds 4 GiB (buffer addressing)");
  SlotState* st = state(ctx);
  if (!st) { st = new SlotState; ctx->slots = st; }
  SlotList* S = &st->side[side];
  const int rows = std::min(64, std::max(8, 4096 / ctx->ld));
  if (!S->built || S->rows != rows || S->nnz != ctx->mat[MFX_MAT_TRAIN].nnz) {
    int rc = build_slots(ctx, S, rows, side);
    if (rc) return rc;
  }
  const uint32_t k0 = mfx_mix32(o->seed ^ 0x3c6ef372U) + (uint32_t)o->epoch * 0x9e3779b9U;
  const uint32_t k1 = mfx_mix32(o->seed * 0x85ebca6bU + 0xdaa66d2bU) ^ mfx_mix32((uint32_t)o->epoch + 0x1b873593U);
  st->last_k0 = k0; st->last_k1 = k1; st->last_side = side;
  int blocks = env_blocks();
  // `blocks` counts 256-thread workgroups (include/mfx.h); this kernel runs WG-thread ones
  if (blocks <= 0) blocks = o->blocks > 0 ? std::min(o->blocks, 8192) : std::max(8, std::min(2048, ctx->nU / 64));
  blocks = std::max(8, blocks * 256 / WG);
  ctx->elist_n = -1;   // the visiting order is not materialised; mfx_debug_epoch_list rebuilds it on demand
  const int L = ctx->L, C = ctx->C;
  if (L == 4) return launch_arith<4, 1>(ctx, S, side, o, blocks, k0, k1);
  if (L == 8) return launch_arith<8, 1>(ctx, S, side, o, blocks, k0, k1);
  switch (C) {
    case 1: return launch_arith<16, 1>(ctx, S, side, o, blocks, k0, k1);   // (8 lanes x 2 chunks was tried: 64 VGPRs do not hold it, 2.5x slower)
    case 2: return launch_arith<16, 2>(ctx, S, side, o, blocks, k0, k1);
    case 3: return launch_arith<16, 3>(ctx, S, side, o, blocks, k0, k1);
    case 4: return launch_arith<16, 4>(ctx, S, side, o, blocks, k0, k1);
    case 5: return launch_arith<16, 5>(ctx, S, side, o, blocks, k0, k1);
    case 6: return launch_arith<16, 6>(ctx, S, side, o, blocks, k0, k1);
    case 7: return launch_arith<16, 7>(ctx, S, side, o, blocks, k0, k1);
    case 8: return launch_arith<16, 8>(ctx, S, side, o, blocks, k0, k1);
  }
  return mfx_fail(ctx, MFX_E_ARG, "sgd tiled: unsupported rank shape L=%d C=%d", L, C);
}

// fills ctx->eu/ei/er with the order the last tiled epoch visited (test hook)
int mfx_slots_materialise_order(mfx_ctx* ctx) {
  SlotState* st = state(ctx);
  NEED(st && st->side[st->last_side].built, MFX_E_STATE, "no tiled epoch has run");
  SlotList* S = &st->side[st->last_side];
  const int blocks = (int)std::min<int64_t>(std::max<int64_t>(S->nslots, 1), 4096);
  hipLaunchKernelGGL(slots_epoch_list_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const int4*)S->rec,
                     S->slot_beg, S->nslots, st->last_side, st->last_k0, st->last_k1, ctx->eu, ctx->ei, ctx->er);
  HIPCHK(hipGetLastError());
  ctx->elist_n = S->nnz;
  return MFX_OK;
}

static uint64_t fnv1a(const void* p, size_t n, uint64_t h = 1469598103934665603ULL) {
  const uint8_t* b = (const uint8_t*)p;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ULL; }
  return h;
}
template <typename T>
static int digest_dev(mfx_ctx* ctx, const T* dev, size_t n, uint64_t* out) {
  std::vector<T> h(n);
  if (n) HIPCHK(hipMemcpy(h.data(), dev, sizeof(T) * n, hipMemcpyDeviceToHost));
  *out = fnv1a(h.data(), sizeof(T) * n);
  return MFX_OK;
}
extern "C" int mfx_debug_slots_digest(mfx_ctx* ctx, int64_t counts[4], uint64_t sums[5]) {
  if (!ctx) return MFX_E_ARG;
  SlotState* st = state(ctx);
  NEED(st && st->side[st->last_side].built && counts && sums, MFX_E_STATE, "mfx_debug_slots_digest: no tiled epoch has run");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const SlotList& S = st->side[st->last_side];
  std::vector<int32_t> ib((size_t)S.nslots + 1);
  HIPCHK(hipMemcpy(ib.data(), S.slot_ibeg, sizeof(int32_t) * ib.size(), hipMemcpyDeviceToHost));
  const int64_t refs = ib[(size_t)S.nslots];
  counts[0] = S.nslots; counts[1] = S.nnz; counts[2] = refs; counts[3] = S.rows;
  int rc;
  if ((rc = digest_dev(ctx, S.rec, (size_t)S.nnz * 4, &sums[0]))) return rc;
  if ((rc = digest_dev(ctx, S.slot_beg, (size_t)S.nslots + 1, &sums[1]))) return rc;
  if ((rc = digest_dev(ctx, S.slot_ibeg, (size_t)S.nslots + 1, &sums[2]))) return rc;
  if ((rc = digest_dev(ctx, S.slot_items, (size_t)refs, &sums[3]))) return rc;
  if ((rc = digest_dev(ctx, S.tile_slot, (size_t)NTILE + 1, &sums[4]))) return rc;
  return MFX_OK;
}
