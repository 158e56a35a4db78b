// sgd_slots.h -- slot lists of MFX_SGD_TILED, shared by the kernel (sgd_slots.hip) and the device-side
// builder (setup.hip).
#ifndef MFX_SGD_SLOTS_H_
#define MFX_SGD_SLOTS_H_

#include <vector>

#include "mfx_internal.h"

// Slot shape (compile-time; scripts/exp_slots.sh builds variants): ratings per slot, LDS words of owned rows per workgroup and
// the most rows a slot stages.
#ifndef MFX_SLOT_CAP
#define MFX_SLOT_CAP 1024
#endif
#ifndef MFX_SLOT_WORDS
#define MFX_SLOT_WORDS 4096
#endif
#ifndef MFX_SLOT_ROWS
#define MFX_SLOT_ROWS 64
#endif
constexpr int CAP_R = MFX_SLOT_CAP;   // ratings per slot
// owned rows per slot: MFX_SLOT_WORDS * 4 bytes of LDS per workgroup, at most MFX_SLOT_ROWS and at least 8 rows
constexpr int slot_rows_for(int ld) {
  return MFX_SLOT_WORDS / ld > MFX_SLOT_ROWS ? MFX_SLOT_ROWS : (MFX_SLOT_WORDS / ld < 8 ? 8 : MFX_SLOT_WORDS / ld);
}
// User blocks per XCD (MFX_NUB): 8*SUB user blocks x 8 item blocks = 64*SUB tiles, 8*SUB rounds per epoch.
#ifndef MFX_SUB
#define MFX_SUB 1
#endif
constexpr int SUB = MFX_SUB, NUB = 8 * SUB, NTILE = NUB * 8;
__host__ __device__ static inline int slot_user_block(int32_t u) {
  return (int)(mfx_mix32((uint32_t)u * 0x9e3779b1U + 0x1234567U) & (uint32_t)(NUB - 1));
}
constexpr int WG = 1024;      // threads per workgroup: 16 waves = 64 ratings in flight on one slot
// SlotList::ctr: [0, NTILE) slots pulled per tile | [NTILE] barrier counter of the drain | [NTILE + 1] abort flag | [NTILE + 2] the
// STICKY abort flag (not cleared per epoch) | [NTILE + 3] "the rounds left something" (slots_left_kernel)
constexpr int CTR_WORDS = NTILE + 4;

struct SlotList {
  int32_t* rec = nullptr;          // int4 per rating: other-side index, local owned index, rating bits, owned index
  int64_t* slot_beg = nullptr;     // [nslots+1] rating range of a slot
  int32_t* slot_ibeg = nullptr;    // [nslots+1] range into slot_items
  int32_t* slot_items = nullptr;   // global item ids of every slot
  int32_t* tile_slot = nullptr;    // [NTILE+1] slot range of a tile
  uint8_t *ublk = nullptr, *iblk = nullptr;   // user block / item block of every row (mfx_slots_block_tables)
  std::vector<uint8_t> h_ublk, h_iblk;        // their host copies (host builder, mfx_debug_tile_blocks)
  unsigned* ctr = nullptr;         // [NTILE] slot counters + [2] barrier counter and abort flag of the drain
  unsigned* abort_host = nullptr;  // pinned copy of the abort flag of the previous epoch's drain
  unsigned* visit = nullptr;       // [nnz] visits per rating record (MFX_SGD_F_COUNT_VISITS), or NULL
  // the one-launch epoch (PERSIST): finished visits of every owned row in this epoch, and for every (slot, row) entry of
  // slot_items the number of visits that precede this slot's in round order (mfx_slots_build_needs)
  unsigned* rowver = nullptr;
  uint8_t* slot_need = nullptr;
  int64_t rowver_n = 0;
  int32_t* attr = nullptr;         // sibling models: per rating (slot order) weight bits (var 1) or rank (var 2)
  int var = 0;                     // which of them attr holds (0: none)
  uint64_t attr_gen = 0;           // ctx->var_gen it was computed for
  int64_t nslots = 0, nnz = 0;
  int rows = 0;                    // owned rows per slot the lists were built for
  int active_waves = WG / 64;      // waves of a workgroup that work on a slot (fewer when a tile has few lock-free rows)
  int tiling = 0;                  // which dealing of the rows into blocks this list was built on (mfx_slots_block_tables)
  bool shared_ctr = false;         // ctr / abort_host are tiling 0's (ONE sticky abort flag per side, whichever tiling ran)
  bool built = false;
};
// Tilings.  ONE dealing of the users and items into blocks, used epoch after epoch, is a structure the shuffled loops of the
// reference do not have: a user's ratings always arrive grouped by the same 8 item blocks, in the same cyclic order, and the
// model that is evaluated at the end of an epoch has just been fitted to the last round's pairing.  Measured with the oracle's
// SEQUENTIAL pass over such lists (tests/tools/order_effect.py, the 2.4 M-rating fixture, lr 0.005, no concurrency at all): test
// RMSE at the best validation epoch 0.62303 for uniformly shuffled epochs, 0.62500 for one static 8 x 8 tiling (+2e-3: what
// round 3 called "the block order"), 0.62395 with the rounds of an epoch in a fresh random order, 0.62338 / 0.62303 / 0.62305
// with 2 / 4 / 8 different tilings taken in turn (and rounds in random order), 0.62319 with a fresh tiling every epoch.  So the
// epochs cycle through MFX_SGD_TILINGS (default 4) independently dealt tilings -- each balanced over the ratings like the
// first -- and walk their rounds in a per-epoch random order; tiling 0 is the deterministic dealing of round 3.
constexpr int MAX_TILINGS = 8;
// The ratings a slot list is built from: the whole train matrix (CSR arrays) or one ITEM PART of it (mfx_sgd_set_item_parts:
// the multi-GPU rotation runs an epoch as nparts sub-epochs, each on the ratings whose item i has i % nparts == part).
struct RatingView { const int32_t* u = nullptr; const int32_t* i = nullptr; const float* r = nullptr; int64_t n = 0; };
// side 0: item rows owned (slots item-major), side 1: user rows owned (slots user-major)
struct SlotState {
  SlotList side[2][MAX_TILINGS];
  uint32_t last_k0 = 0, last_k1 = 0;
  int last_side = 0, last_part = -1;       // last_part >= 0: the last epoch ran on parts[last_part]
  int last_tiling = 0;
  SlotList* visit_list = nullptr;          // the list of the last epoch run with MFX_SGD_F_COUNT_VISITS (mfx_debug_visit_counts)
  int xcc_probe_blocks = -1;               // mfx_xcc_ids_populated: the grid size probed last and what it found
  bool xcc_probe_ok = false;
  // item parts: the train ratings grouped by part (COO copy, part p = [poff[p], poff[p+1])) and one slot list per part
  int nparts = 0;
  int32_t *pu = nullptr, *pi = nullptr;
  float* pv = nullptr;
  std::vector<int64_t> poff;
  std::vector<SlotList> parts;
  SlotList& last() { return last_part >= 0 ? parts[(size_t)last_part] : side[last_side][last_tiling]; }
};

// position t of a slot's visiting order (Feistel permutation of [0, R) keyed per epoch and slot)
__device__ __forceinline__ int64_t slot_perm(int64_t t, int64_t R, uint32_t k0, uint32_t k1) {
  int bits = 2;
  while (((int64_t)1 << bits) < R) bits++;
  const int ab = bits / 2;
  return mfx_perm_index(t, R, ab, bits - ab, k0, k1);
}

// Does a grid of `blocks` 1024-thread workgroups put at least 8 workgroups on each of the eight XCC ids (HW_REG_XCC_ID)?  Probed once per
// context and grid size with a kernel that does nothing else; the one-launch epoch needs it (a partition mode that reports one
// id would leave seven neighbours that never finish).
bool mfx_xcc_ids_populated(mfx_ctx* ctx, int blocks, int* rc);
// S->slot_need and S->rowver for the one-launch epoch: an owned row is visited by one slot in every tile of its block's column of
// tiles in which it has ratings, in round order r = (item block - user block) mod 8; slot_need = how many of them come before.
int mfx_slots_build_needs(mfx_ctx* ctx, SlotList* S, int64_t nown);
// the kernel launchers, one per rank shape (sgd_slots_inst_<L>x<C>.hip)
#define MFX_SLOTS_DECL(LL, CC) \
  int mfx_slots_launch_##LL##x##CC(mfx_ctx* ctx, SlotList* S, int side, const mfx_sgd_opts* o, int blocks, uint32_t k0, uint32_t k1);
MFX_SLOTS_DECL(4, 1) MFX_SLOTS_DECL(8, 1) MFX_SLOTS_DECL(16, 1) MFX_SLOTS_DECL(16, 2) MFX_SLOTS_DECL(16, 3) MFX_SLOTS_DECL(16, 4)
MFX_SLOTS_DECL(16, 5) MFX_SLOTS_DECL(16, 6) MFX_SLOTS_DECL(16, 7) MFX_SLOTS_DECL(16, 8)
#undef MFX_SLOTS_DECL

// The 8*SUB user blocks and 8 item blocks the tiles are made of, BALANCED over the ratings of `view`: rows in descending order of
// their rating count, each onto the block with the fewest ratings so far (ties: the lower row id first, the lower block).  The
// round of an epoch ends when its slowest tile does -- with hashed blocks the 8 tiles of a diagonal differed by +-12 % at C2
// (the popular items of a block), measured as 11 % of every launch spent waiting for one XCD (scripts/slot_times.py).  Rows the
// view has no rating of keep the hash.  MFX_SGD_BLOCKS_HASH=1: the hash for every row (the layout of rounds 1 and 2).
// Fills S->ublk / S->iblk (device) and the host copies (setup.hip).
int mfx_slots_block_tables(mfx_ctx* ctx, SlotList* S, const RatingView& view);
// builds S on the device from the ratings of `view` (setup.hip); same lists as the host builder in sgd_slots.hip
int mfx_slots_build_device(mfx_ctx* ctx, SlotList* S, int rows, int side, const RatingView& view);
// the train ratings grouped by item part (i % nparts), stable in CSR order: fills st->pu/pi/pv/poff (setup.hip)
int mfx_slots_group_by_part(mfx_ctx* ctx, SlotState* st, int nparts);

#endif
