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
// Slots are pulled from per-tile counters and eight drain launches (the same diagonals, keyed on the workgroup
// index instead of the XCD) take whatever is left, so "every rating exactly once per epoch" and "one owner per
// item row" hold for ANY workgroup->XCD placement.
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
  auto drop = [](SlotList& s) {
    dev_free(s.rec); dev_free(s.slot_beg); dev_free(s.slot_ibeg); dev_free(s.slot_items); dev_free(s.attr); dev_free(s.visit);
    dev_free(s.tile_slot); dev_free(s.ublk); dev_free(s.iblk); dev_free(s.rowver); dev_free(s.slot_need);
    if (!s.shared_ctr) {
      dev_free(s.ctr);
      if (s.abort_host) (void)hipHostFree(s.abort_host);
    }
  };
  for (auto& sd : st->side)
    for (SlotList& s : sd) drop(s);
  for (SlotList& s : st->parts) drop(s);
  dev_free(st->pu); dev_free(st->pi); dev_free(st->pv);
  delete st;
  ctx->slots = nullptr;
}

__global__ __launch_bounds__(WG) void xcc_probe_kernel(unsigned* __restrict__ count) {
  if (threadIdx.x == 0) atomicAdd(&count[__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7], 1u);   // HW_REG_XCC_ID[3:0]
}
bool mfx_xcc_ids_populated(mfx_ctx* ctx, int blocks, int* rc) {
  *rc = MFX_OK;
  SlotState* st = state(ctx);
  if (st->xcc_probe_blocks == blocks) return st->xcc_probe_ok;
  unsigned* d = nullptr;
  unsigned h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if ((*rc = dev_alloc(ctx, &d, (size_t)8))) return false;
  auto fail = [&](hipError_t e) { if (e != hipSuccess) { *rc = mfx_fail(ctx, MFX_E_HIP, "XCC probe: %s", hipGetErrorString(e)); } return e != hipSuccess; };
  if (fail(hipMemsetAsync(d, 0, sizeof h, ctx->stream))) { dev_free(d); return false; }
  hipLaunchKernelGGL(xcc_probe_kernel, dim3(blocks), dim3(WG), 0, ctx->stream, d);
  if (fail(hipGetLastError()) || fail(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream)) || fail(hipStreamSynchronize(ctx->stream))) {
    dev_free(d);
    return false;
  }
  dev_free(d);
  bool ok = true;
  for (int x = 0; x < 8; x++) ok = ok && h[x] >= 8;
  if (getenv("MFX_DEBUG"))
    fprintf(stderr, "[mfx] XCC ids of a %d-workgroup grid: %u %u %u %u %u %u %u %u -> one-launch epoch %s\n", blocks, h[0], h[1], h[2], h[3], h[4],
            h[5], h[6], h[7], ok ? "on" : "off");
  st->xcc_probe_blocks = blocks;
  st->xcc_probe_ok = ok;
  return ok;
}

// one thread per slot: the user block of its tile into the presence mask of every owned row it stages ...
__global__ void need_presence_kernel(const int32_t* __restrict__ tile_slot, const int32_t* __restrict__ slot_ibeg, const int32_t* __restrict__ slot_items,
                                     int64_t nslots, unsigned* __restrict__ pres) {
  for (int64_t sl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; sl < nslots; sl += (int64_t)gridDim.x * blockDim.x) {
    int t = 0;
    while (t + 1 < NTILE && tile_slot[t + 1] <= sl) t++;
    const unsigned bit = 1u << (t / 8);
    for (int e = slot_ibeg[sl]; e < slot_ibeg[sl + 1]; e++) atomicOr(&pres[slot_items[e]], bit);
  }
}
// ... and, per (slot, row) entry, how many tiles with that row come earlier in round order (round of tile (ub, ib) = (ib - ub) mod 8)
__global__ void need_count_kernel(const int32_t* __restrict__ tile_slot, const int32_t* __restrict__ slot_ibeg, const int32_t* __restrict__ slot_items,
                                  int64_t nslots, const unsigned* __restrict__ pres, uint8_t* __restrict__ need) {
  for (int64_t sl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; sl < nslots; sl += (int64_t)gridDim.x * blockDim.x) {
    int t = 0;
    while (t + 1 < NTILE && tile_slot[t + 1] <= sl) t++;
    const int ub = t / 8, ib = t % 8, r = (ib - ub) & 7;
    for (int e = slot_ibeg[sl]; e < slot_ibeg[sl + 1]; e++) {
      const unsigned p = pres[slot_items[e]];
      int n = 0;
      for (int u2 = 0; u2 < 8; u2++) n += ((p >> u2) & 1u) && ((ib - u2) & 7) < r;
      need[e] = (uint8_t)n;
    }
  }
}
int mfx_slots_build_needs(mfx_ctx* ctx, SlotList* S, int64_t nown) {
  NEED(SUB == 1 && NUB == 8, MFX_E_STATE, "the one-launch epoch: round of a tile = (item block - user block) mod 8 needs 8 user blocks");
  int rc;
  if (S->slot_need && S->rowver && S->rowver_n == nown) return MFX_OK;
  dev_free(S->slot_need); dev_free(S->rowver);
  S->rowver_n = 0;
  int32_t refs = 0;
  HIPCHK(hipMemcpyAsync(&refs, S->slot_ibeg + S->nslots, sizeof refs, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if ((rc = dev_alloc(ctx, &S->slot_need, (size_t)std::max(refs, 1))) || (rc = dev_alloc(ctx, &S->rowver, (size_t)std::max<int64_t>(nown, 1)))) return rc;
  HIPCHK(hipMemsetAsync(S->rowver, 0, sizeof(unsigned) * (size_t)std::max<int64_t>(nown, 1), ctx->stream));      // (the presence masks, for now)
  const int grid = (int)std::min<int64_t>(std::max<int64_t>((S->nslots + 255) / 256, 1), 4096);
  hipLaunchKernelGGL(need_presence_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t*)S->tile_slot, (const int32_t*)S->slot_ibeg,
                     (const int32_t*)S->slot_items, S->nslots, S->rowver);
  hipLaunchKernelGGL(need_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t*)S->tile_slot, (const int32_t*)S->slot_ibeg,
                     (const int32_t*)S->slot_items, S->nslots, (const unsigned*)S->rowver, S->slot_need);
  HIPCHK(hipGetLastError());
  S->rowver_n = nown;
  return MFX_OK;
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

static int build_slots(mfx_ctx* ctx, SlotList* S, int rows, int side, const RatingView& view) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const int64_t nnz = view.n;
  // the device builder (setup.hip) makes the same lists; this host version is the readable statement of
  // them, the cross-check of tests/test_setup_gpu.py (MFX_SLOTS_HOST=1) and the path for nnz >= 2^31
  if (nnz > 0 && nnz < ((int64_t)1 << 31) && !getenv("MFX_SLOTS_HOST")) return mfx_slots_build_device(ctx, S, rows, side, view);
  std::vector<int32_t> ru((size_t)nnz), ri((size_t)nnz);
  std::vector<float> rv((size_t)nnz);
  if (nnz) {
    HIPCHK(hipMemcpy(ru.data(), view.u, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ri.data(), view.i, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rv.data(), view.r, sizeof(float) * (size_t)nnz, hipMemcpyDeviceToHost));
  }
  // own[] = index on the owned side (grouping key), oth[] = index on the lock-free side
  const std::vector<int32_t>& own = side == 0 ? ri : ru;
  const std::vector<int32_t>& oth = side == 0 ? ru : ri;
  const int32_t nown = side == 0 ? m.ncols : m.nrows;
  // ratings of each tile (stable in CSR order); the blocks are balanced over this view's ratings (mfx_slots_block_tables)
  {
    int rc = mfx_slots_block_tables(ctx, S, view);
    if (rc) return rc;
  }
  const std::vector<uint8_t>&ub = S->h_ublk, &ib = S->h_iblk;
  std::vector<int64_t> tstart(NTILE + 1, 0);
  for (int64_t e = 0; e < nnz; e++) tstart[(int)ub[(size_t)ru[e]] * 8 + (int)ib[(size_t)ri[e]] + 1]++;
  for (int t = 0; t < NTILE; t++) tstart[t + 1] += tstart[t];
  std::vector<int64_t> byt((size_t)nnz);
  {
    std::vector<int64_t> pos(tstart.begin(), tstart.end() - 1);
    for (int64_t e = 0; e < nnz; e++) byt[pos[(int)ub[(size_t)ru[e]] * 8 + (int)ib[(size_t)ri[e]]]++] = e;
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
  dev_free(S->slot_need); dev_free(S->rowver); S->rowver_n = 0;      // (of the lists that go)
  dev_free(S->visit); dev_free(S->attr); dev_free(S->rec); dev_free(S->slot_beg); dev_free(S->slot_ibeg); dev_free(S->slot_items);
  dev_free(S->tile_slot);
  int rc;
  if ((rc = up(ctx, &S->rec, rec))) return rc;
  if ((rc = up(ctx, &S->slot_beg, slot_beg))) return rc;
  if ((rc = up(ctx, &S->slot_ibeg, slot_ibeg))) return rc;
  if ((rc = up(ctx, &S->slot_items, slot_items))) return rc;
  if ((rc = up(ctx, &S->tile_slot, tile_slot))) return rc;
  if (!S->ctr) {
    if ((rc = dev_alloc(ctx, &S->ctr, (size_t)CTR_WORDS))) return rc;
    HIPCHK(hipMemsetAsync(S->ctr, 0, CTR_WORDS * sizeof(unsigned), ctx->stream));      // [NTILE + 2]: the sticky abort flag
  }
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


// weight bits (var 1) or rank (var 2) of every rating in slot order, from the installed attribute tables
__global__ void slots_attr_kernel(const int4* __restrict__ rec, int64_t nnz, int var, const float2* __restrict__ ua,
                                  const float2* __restrict__ ia, float rho, const int2* __restrict__ tu, const int2* __restrict__ ti,
                                  const int32_t* __restrict__ du, const int32_t* __restrict__ di, const double* __restrict__ dexp,
                                  uint32_t seed, uint32_t epoch, int K, int32_t* __restrict__ attr) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x) {
    const int4 r = rec[t];                 // owned item rows: x = user, w = item
    if (var == 1) { attr[t] = __float_as_int(mfx_ifw_weight(ua[r.x], ia[r.w], rho)); continue; }
    const int2 a = tu[r.x], b = ti[r.w];
    int rank = mfx_tmf_rank(a, b);
    if (du) {                              // ModelPoissonDropout: a fresh Poisson(lambda) draw per visit, i.e. per epoch
      const int lam = __int_as_float(a.x) < __int_as_float(b.x) ? du[r.x] : di[r.w];
      rank = mfx_poisson_rank(lam, dexp[lam], mfx_draw_hash(seed, epoch, (uint32_t)r.x, (uint32_t)r.w), K);
    }
    attr[t] = rank;
  }
}

int mfx_launch_sgd_tiled(mfx_ctx* ctx, const mfx_sgd_opts* o) {
  // which side is owned this epoch: o->own = 0 item rows (default), 1 user rows, 2 alternate
  const int side = o->own == 1 ? 1 : o->own == 2 ? (o->epoch & 1) : 0;
  const uint64_t ob = (uint64_t)(side ? ctx->nI : ctx->nU) * ctx->ld * 4;
  NEED(ob < (1ull << 32), MFX_E_ARG, "MFX_SGD_TILED: a factor matrix exceeds 4 GiB (buffer addressing)");
  SlotState* st = state(ctx);
  if (!st) { st = new SlotState; ctx->slots = st; }
  const DevCSR& tm = ctx->mat[MFX_MAT_TRAIN];
  RatingView view;
  view.u = tm.rowid; view.i = tm.rowind; view.r = tm.rowval; view.n = tm.nnz;
  // which tiling this epoch runs on (sgd_slots.h): epochs take the tilings in turn; the test hooks that compare with a restated
  // order (ONE_GROUP) and the item parts of the multi-GPU rotation stay on tiling 0
  static const int n_tilings = [] {
    const char* e = getenv("MFX_SGD_TILINGS");
    return std::max(1, std::min(MAX_TILINGS, e ? atoi(e) : 4));
  }();
  const int tiling = (o->item_part != 0 || (o->flags & MFX_SGD_F_ONE_GROUP)) ? 0 : (int)((uint32_t)o->epoch % (uint32_t)n_tilings);
  SlotList* S = &st->side[side][tiling];
  S->tiling = tiling;
  {
    // one set of slot counters and ONE sticky abort flag per side: the tilings are used one epoch at a time on one stream, and a
    // drain that gave up must be reported by the next epoch whichever tiling that runs on
    SlotList* S0 = &st->side[side][0];
    if (!S0->ctr) {
      int rc;
      if ((rc = dev_alloc(ctx, &S0->ctr, (size_t)CTR_WORDS))) return rc;
      HIPCHK(hipMemsetAsync(S0->ctr, 0, CTR_WORDS * sizeof(unsigned), ctx->stream));
    }
    if (!S0->abort_host) {
      HIPCHK(hipHostMalloc((void**)&S0->abort_host, sizeof(unsigned), hipHostMallocDefault));
      *S0->abort_host = 0;
    }
    if (S != S0) { S->ctr = S0->ctr; S->abort_host = S0->abort_host; S->shared_ctr = true; }
  }
  const int part = o->item_part - 1;
  if (o->item_part != 0) {
    // one item part of the train matrix (the multi-GPU rotation): its ratings come from the part-grouped copy
    NEED(ctx->item_parts > 1 && part >= 0 && part < ctx->item_parts, MFX_E_ARG, "MFX_SGD_TILED: item_part %d without mfx_sgd_set_item_parts (%d parts)",
         o->item_part, ctx->item_parts);
    NEED(side == 0, MFX_E_ARG, "MFX_SGD_TILED: item parts need own = 0 (item rows owned)");
    if (st->nparts != ctx->item_parts || !st->pu) {
      for (SlotList& s : st->parts) {
        dev_free(s.rec); dev_free(s.slot_beg); dev_free(s.slot_ibeg); dev_free(s.slot_items); dev_free(s.attr); dev_free(s.visit);
        dev_free(s.tile_slot); dev_free(s.ctr); dev_free(s.ublk); dev_free(s.iblk); dev_free(s.rowver); dev_free(s.slot_need);
        if (s.abort_host) (void)hipHostFree(s.abort_host);
      }
      st->parts.clear();
      st->visit_list = nullptr;
      int rc = mfx_slots_group_by_part(ctx, st, ctx->item_parts);
      if (rc) return rc;
      st->parts.resize((size_t)ctx->item_parts);
    }
    S = &st->parts[(size_t)part];
    view.u = st->pu + st->poff[(size_t)part]; view.i = st->pi + st->poff[(size_t)part]; view.r = st->pv + st->poff[(size_t)part];
    view.n = st->poff[(size_t)part + 1] - st->poff[(size_t)part];
    if (view.n == 0) return MFX_OK;           // this rank's users hold no rating of the part
  }
  const int rows = slot_rows_for(ctx->ld);
  if (!S->built || S->rows != rows || S->nnz != view.n) {
    int rc = build_slots(ctx, S, rows, side, view);
    if (rc) return rc;
  }
  // sibling models: the weight / rank of every rating rides next to its record
  const int var = ctx->ifw ? 1 : (ctx->tmf_u ? 2 : (ctx->dimreg ? 3 : 0));
  NEED(var == 0 || side == 0, MFX_E_ARG, "MFX_SGD_TILED: the SGD variants need own = 0 (item rows owned)");
  if ((var == 1 || var == 2) && (S->var != var || S->attr_gen != ctx->var_gen || !S->attr || (var == 2 && ctx->tmfd_u))) {
    int rc;
    if (!S->attr && (rc = dev_alloc(ctx, &S->attr, (size_t)S->nnz))) return rc;
    const float2 *ua = nullptr, *ia = nullptr;
    float rho = 0.0f;
    mfx_ifw_tables(ctx, &ua, &ia, &rho);
    hipLaunchKernelGGL(slots_attr_kernel, dim3(4096), dim3(256), 0, ctx->stream, (const int4*)S->rec, S->nnz, var, ua, ia, rho,
                       ctx->tmf_u, ctx->tmf_i, ctx->tmfd_u, ctx->tmfd_i, ctx->tmfd_exp, ctx->tmfd_seed, (uint32_t)o->epoch, ctx->K,
                       S->attr);
    HIPCHK(hipGetLastError());
    S->attr_gen = ctx->var_gen;
  }
  S->var = var;
  {
    // 16 waves = 64 ratings of one slot in flight at K > 32, 256 at K <= 16 (4 lanes per rating).  On a small matrix that is a
    // large share of the lock-free rows of a tile (a 3000 x 2000 matrix: 375 users per tile): updates of one row overlap and
    // are lost and the run ends 2e-2 above the reference's RMSE.  One wave per 16 x (ratings per wave step) lock-free rows of a
    // tile -- 16 waves from 1 024 rows per tile (K > 32) resp. 4 096 (K <= 16) on; scripts/nan_check.py: 0.687 -> 0.674 vs 0.669.
    const int64_t rows_per_tile = (side ? ctx->nI : ctx->nU) / NUB;
    const char* e = getenv("MFX_SGD_WAVES");
    S->active_waves = e ? std::max(1, std::min(WG / 64, atoi(e)))
                        : (int)std::max<int64_t>(1, std::min<int64_t>(WG / 64, rows_per_tile / (16 * (64 / ctx->L))));
  }
  if (!S->abort_host) {
    HIPCHK(hipHostMalloc((void**)&S->abort_host, sizeof(unsigned), hipHostMallocDefault));
    *S->abort_host = 0;
  }
  {
    const int rc = mfx_slots_check_abort(ctx);      // what the copies that have ARRIVED say; the flag is sticky, a later check sees the rest
    if (rc) return rc;
  }
  if ((o->flags & MFX_SGD_F_COUNT_VISITS) && !S->visit) {
    int rc;
    if ((rc = dev_alloc(ctx, &S->visit, (size_t)S->nnz))) return rc;
    HIPCHK(hipMemsetAsync(S->visit, 0, sizeof(unsigned) * (size_t)S->nnz, ctx->stream));
  }
  if (o->flags & MFX_SGD_F_COUNT_VISITS) st->visit_list = S;
  const uint32_t k0 = mfx_mix32(o->seed ^ 0x3c6ef372U) + (uint32_t)o->epoch * 0x9e3779b9U;
  const uint32_t k1 = mfx_mix32(o->seed * 0x85ebca6bU + 0xdaa66d2bU) ^ mfx_mix32((uint32_t)o->epoch + 0x1b873593U);
  st->last_k0 = k0; st->last_k1 = k1; st->last_side = side; st->last_part = o->item_part != 0 ? part : -1;
  st->last_tiling = tiling;
  int blocks = env_blocks();
  // `blocks` counts 256-thread workgroups (include/mfx.h); this kernel runs WG-thread ones
  if (blocks <= 0) blocks = o->blocks > 0 ? std::min(o->blocks, 8192) : std::max(8, std::min(2048, ctx->nU / 64));
  blocks = std::max(8, blocks * 256 / WG);
  ctx->elist_n = -1;   // the visiting order is not materialised; mfx_debug_epoch_list rebuilds it on demand
  const int L = ctx->L, C = ctx->C;
  if (L == 4) return mfx_slots_launch_4x1(ctx, S, side, o, blocks, k0, k1);
  if (L == 8) return mfx_slots_launch_8x1(ctx, S, side, o, blocks, k0, k1);
  switch (C) {
    case 1: return mfx_slots_launch_16x1(ctx, S, side, o, blocks, k0, k1);   // (8 lanes x 2 chunks was tried: 64 VGPRs do not hold it, 2.5x slower)
    case 2: return mfx_slots_launch_16x2(ctx, S, side, o, blocks, k0, k1);
    case 3: return mfx_slots_launch_16x3(ctx, S, side, o, blocks, k0, k1);
    case 4: return mfx_slots_launch_16x4(ctx, S, side, o, blocks, k0, k1);
    case 5: return mfx_slots_launch_16x5(ctx, S, side, o, blocks, k0, k1);
    case 6: return mfx_slots_launch_16x6(ctx, S, side, o, blocks, k0, k1);
    case 7: return mfx_slots_launch_16x7(ctx, S, side, o, blocks, k0, k1);
    case 8: return mfx_slots_launch_16x8(ctx, S, side, o, blocks, k0, k1);
  }
  return mfx_fail(ctx, MFX_E_ARG, "sgd tiled: unsupported rank shape L=%d C=%d", L, C);
}

// Reports (once) a drain that gave up at its grid barrier.  Called behind the stream synchronisations of the C entry points
// (mfx_synchronize, mfx_eval*, mfx_get_factors) -- there the copy of the last epoch's flag has arrived -- and at the start of
// a tiled epoch.  The device flag is sticky: it is cleared here, when it is reported, and nowhere else.
int mfx_slots_check_abort(mfx_ctx* ctx) {
  SlotState* st = state(ctx);
  if (!st) return MFX_OK;
  for (size_t k = 0; k < 2 * MAX_TILINGS + st->parts.size(); k++) {
    SlotList& S = k < 2 * MAX_TILINGS ? st->side[k / MAX_TILINGS][k % MAX_TILINGS] : st->parts[k - 2 * MAX_TILINGS];
    if (!S.abort_host || !*(volatile unsigned*)S.abort_host) continue;
    // reported ONCE: copies of the sticky flag queued by epochs that are still in flight would set the host word again behind
    // this point (the start-of-epoch check does not synchronise), so the stream is drained first -- an error path, taken once --
    // and both words are cleared behind everything that could still write them
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (S.ctr) HIPCHK(hipMemsetAsync(S.ctr + NTILE + 2, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *S.abort_host = 0;
    return mfx_fail(ctx, MFX_E_HIP, "MFX_SGD_TILED: the drain of an epoch gave up at its grid barrier (no progress for 2 s; is the device "
                    "shared with another resident kernel?) -- that epoch may have left ratings unvisited");
  }
  return MFX_OK;
}

extern "C" int mfx_sgd_set_item_parts(mfx_ctx* ctx, int nparts) {
  if (!ctx) return MFX_E_ARG;
  NEED(nparts >= 0 && nparts <= 1024, MFX_E_ARG, "mfx_sgd_set_item_parts: nparts=%d", nparts);
  ctx->item_parts = nparts >= 2 ? nparts : 0;       // the part lists are (re)built by the next epoch that names a part
  return MFX_OK;
}

extern "C" int mfx_debug_raise_drain_abort(mfx_ctx* ctx) {
  if (!ctx) return MFX_E_ARG;
  SlotState* st = state(ctx);
  NEED(st && st->last().built && st->last().ctr, MFX_E_STATE, "mfx_debug_raise_drain_abort: no tiled epoch has run");
  HIPCHK(hipSetDevice(ctx->device));
  const unsigned one = 1;
  HIPCHK(hipMemcpyAsync(st->last().ctr + NTILE + 2, &one, sizeof one, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MFX_OK;
}

// fills ctx->eu/ei/er with the order the last tiled epoch visited (test hook)
int mfx_slots_materialise_order(mfx_ctx* ctx) {
  SlotState* st = state(ctx);
  NEED(st && st->last().built, MFX_E_STATE, "no tiled epoch has run");
  SlotList* S = &st->last();
  const int blocks = (int)std::min<int64_t>(std::max<int64_t>(S->nslots, 1), 4096);
  hipLaunchKernelGGL(slots_epoch_list_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const int4*)S->rec,
                     S->slot_beg, S->nslots, st->last_side, st->last_k0, st->last_k1, ctx->eu, ctx->ei, ctx->er);
  HIPCHK(hipGetLastError());
  ctx->elist_n = S->nnz;
  return MFX_OK;
}

extern "C" int mfx_debug_visit_counts(mfx_ctx* ctx, uint32_t* counts, int64_t cap, int64_t* n) {
  if (!ctx) return MFX_E_ARG;
  SlotState* st = state(ctx);
  NEED(n, MFX_E_ARG, "mfx_debug_visit_counts: n NULL");
  NEED(st && st->visit_list && st->visit_list->built && st->visit_list->visit, MFX_E_STATE,
       "mfx_debug_visit_counts: no tiled epoch has run with MFX_SGD_F_COUNT_VISITS");
  SlotList& S = *st->visit_list;
  *n = S.nnz;
  if (!counts) return MFX_OK;
  NEED(cap >= S.nnz, MFX_E_ARG, "mfx_debug_visit_counts: cap too small");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(counts, S.visit, sizeof(unsigned) * (size_t)S.nnz, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(S.visit, 0, sizeof(unsigned) * (size_t)S.nnz));
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
extern "C" int mfx_debug_tile_blocks(mfx_ctx* ctx, uint8_t* user_block, int64_t n_users, uint8_t* item_block, int64_t n_items) {
  if (!ctx) return MFX_E_ARG;
  SlotState* st = state(ctx);
  NEED(st && st->last().built && user_block && item_block, MFX_E_STATE, "mfx_debug_tile_blocks: no tiled epoch has run");
  const SlotList& S = st->last();
  NEED(n_users == (int64_t)S.h_ublk.size() && n_items == (int64_t)S.h_iblk.size(), MFX_E_ARG,
       "mfx_debug_tile_blocks: the tables hold %zu users and %zu items", S.h_ublk.size(), S.h_iblk.size());
  memcpy(user_block, S.h_ublk.data(), S.h_ublk.size());
  memcpy(item_block, S.h_iblk.data(), S.h_iblk.size());
  return MFX_OK;
}
extern "C" int mfx_debug_slots_digest(mfx_ctx* ctx, int64_t counts[4], uint64_t sums[5]) {
  if (!ctx) return MFX_E_ARG;
  SlotState* st = state(ctx);
  NEED(st && st->last().built && counts && sums, MFX_E_STATE, "mfx_debug_slots_digest: no tiled epoch has run");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const SlotList& S = st->last();
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
