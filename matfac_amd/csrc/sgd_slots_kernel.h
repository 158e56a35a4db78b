// sgd_slots_kernel.h -- the MFX_SGD_TILED kernel and its launch templates.  Included by one small translation
// unit per rank shape (sgd_slots_inst_*.hip) so that the shapes compile in parallel; see sgd_slots.hip for the
// schedule and the slot lists.
#ifndef MFX_SGD_SLOTS_KERNEL_H_
#define MFX_SGD_SLOTS_KERNEL_H_

#include <algorithm>

#include "sgd_common.h"
#include "sgd_slots.h"

// Diagnostic builds (scripts/exp_bound.sh; results are WRONG on purpose): what the round time does when one resource is
// taken out.  1: no user-row stores  2: no LDS atomic adds  3: 16 extra vector multiplies per step  4: user rows are
// never loaded  5: 32 extra multiplies.  8: no visit counts -- with MFX_SGD_F_COUNT_VISITS every round workgroup leaves
// {start, end (100 MHz wall clock), XCD | slots pulled << 8, ratings} in the visit buffer (scripts/slot_times.py).
// 0 (the product): nothing of this is compiled.
#ifndef MFX_EXP
#define MFX_EXP 0
#endif

// ---------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------
// owned rows per slot (sgd_slots.h): 16 KB of LDS per workgroup, at most 64 and at least 8 rows
template <int LD>
struct SlotRows { static constexpr int value = slot_rows_for(LD); };


// Item rows of a slot live in LDS as Q = round(q * 2^24) (int32).  A visit reads q = float(Q) * 2^-24,
// computes q_new with the reference's arithmetic and adds round((q_new - q) * 2^24) with ds_add_u32:
// no item update is lost and every group reads the freshest row.  (ds_add_f32 was measured 7x slower
// than the whole rest of the kernel; the integer add is free.)  Resolution 6e-8, range +-128; a slot
// whose staged rows are not finite or exceed the range falls back to float rows with plain stores, and a
// row that leaves the range while being updated is written back as NaN, so Model::isTerminateModel's NaN
// guard (model.cpp:1486-1498) still sees a diverged model.
constexpr float FIX_SCALE = 16777216.0f, FIX_INV = 1.0f / 16777216.0f, FIX_MAX = 127.0f;
// round(x) for the fixed-point deltas: v_cvt_rpi_i32_f32 = floor(x + 0.5), ONE conversion instead of v_rndne_f32 +
// v_cvt_i32_f32 (conversions issue at half rate; scripts/valu_probe.hip).  Ties go up instead of to even: both are
// within half a unit (2^-25), which is all the representation promises.
__device__ __forceinline__ int fix_round(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// entry S*G+g of the chunk: row_share broadcast of the transposed chunk (L == 16) or a cross-lane read
template <int L, int S>
__device__ __forceinline__ int slot_take(int v, int g) {
  // row_share:S writes every lane: the form without an "old" operand needs no initialising move
  if constexpr (L == 16) return __builtin_amdgcn_mov_dpp(v, 0x150 + S, 0xF, 0xF, true);
  else return __shfl(v, S * (64 / L) + g, 64);
}

// The L steps of one 64-rating chunk, unrolled by template recursion (the DPP controls are immediates).
// FIX: the slot's rows are in the fixed-point representation (decided per slot, so per chunk it is a template
// argument, not a select per element).
// VAR: 0 the plain update; 1 ModelInvPopMF's weight on the error term (tw = float bits of wt, sgd_ifw.hip);
// 2 ModelDropoutSigmoid's truncated rank (tw = rank, sgd_tmf.hip); 3 trainSGDParSVD's per-dimension regulariser
// (regk = the ld values of mfx_sgd_set_dim_reg, svd.hip; no per-rating attribute).
template <int L, int C, int ARITH, bool OWN_U, bool FIX, int VAR, int S>
struct SlotSteps {
  template <class RowsT>
  static __device__ __forceinline__ void run(const RowsT& Um, int* q_lds, int tx, int ty, int tz, int tw, const float* regk,
                                             int g, int j, int nvalid, float lr, float uReg, float iReg,
                                             float4v (&pn)[C], uint32_t& pen, float4v (&pnn)[C], uint32_t& penn) {
    constexpr int G = 64 / L;
    constexpr int LD = 4 * L * C;
    const int e = S * G + g;
    const int li = slot_take<L, S>(ty, g);
    const float r = __builtin_bit_cast(float, slot_take<L, S>(tz, g));
    const int var = (VAR == 1 || VAR == 2) ? slot_take<L, S>(tw, g) : 0;   // (cross-lane reads stay outside the divergent part)
    // K <= 128: rows of steps S+1 and S+2 are already requested; take S, shift, request S+2.
    // Wider rows: one step ahead only (two buffers of 4C registers each do not fit next to the rows themselves
    // under the 128-VGPR cap of a 1024-thread workgroup: measured spills, -6 % at K = 256)
    // LEAN: the hogTrain path on wide rows re-reads the owned row chunk by chunk from LDS instead of holding it, which
    // leaves room for the two-steps-ahead pipeline (at 125 M ratings / K = 256 the shallow one cost 6 %)
    constexpr bool LEAN = C > 2 && FIX && !OWN_U && ARITH == MFX_ARITH_F32 && VAR == 0;
    constexpr bool DEEP = C <= 2 || LEAN;
    float4v p[C];
    const uint32_t pe = pen;      // byte offset of this step's lock-free row (+ this lane's 16 bytes)
#pragma unroll
    for (int c = 0; c < C; c++) p[c] = pn[c];
    if constexpr (DEEP) {
#pragma unroll
      for (int c = 0; c < C; c++) pn[c] = pnn[c];
      pen = penn;
      if constexpr (S + 2 < L) {
        const int un = slot_take<L, S + 2>(tx, g);
        if (e + 2 * G < nvalid) {
          penn = (uint32_t)un * (uint32_t)(LD * 4) + 16u * (uint32_t)j;
#pragma unroll
          for (int c = 0; c < C; c++) pnn[c] = MFX_EXP == 4 ? float4v{0.01f, 0.02f, 0.03f, 0.01f} : Um.ldb(penn, c * 16 * L);
        }
      }
    } else if constexpr (S + 1 < L) {
      const int un = slot_take<L, S + 1>(tx, g);
      if (e + G < nvalid) {
        pen = (uint32_t)un * (uint32_t)(LD * 4) + 16u * (uint32_t)j;
#pragma unroll
        for (int c = 0; c < C; c++) pn[c] = Um.ldb(pen, c * 16 * L);
      }
    }
    if constexpr (LEAN) {
      if (e < nvalid) {
        int* qrow = q_lds + li * LD + 4 * j;
        float a = 0.0f;
#pragma unroll
        for (int c = 0; c < C; c++) {          // pass 1: the dot, same per-lane chain as group_dot (on the unscaled float(Q))
          const int4 qi = *(const int4*)(qrow + c * 4 * L);
          a = __builtin_fmaf(p[c].x, (float)qi.x, a);
          a = __builtin_fmaf(p[c].y, (float)qi.y, a);
          a = __builtin_fmaf(p[c].z, (float)qi.z, a);
          a = __builtin_fmaf(p[c].w, (float)qi.w, a);
        }
        const float est = group_sum<L>(a) * FIX_INV;
        const float c1 = -2.0f * (r - est), cu = 2.0f * uReg;
        const float c1s = c1 * FIX_INV, cis = (2.0f * iReg) * FIX_INV;
        const float lrs = lr * -FIX_SCALE;     // (see the delta branch below)
#pragma unroll
        for (int c = 0; c < C; c++) {          // pass 2: both steps of this chunk
          const int4 qi = *(const int4*)(qrow + c * 4 * L);
          const float4v q = float4v{(float)qi.x, (float)qi.y, (float)qi.z, (float)qi.w};
#pragma unroll
          for (int x = 0; x < 4; x++) p[c][x] = upd_f32(p[c][x], q[x], c1s, cu, lr);
          Um.stb(pe, c * 16 * L, p[c]);
#pragma unroll
          for (int x = 0; x < 4; x++) {
            atomicAdd(qrow + c * 4 * L + x, fix_round(lrs * (c1 * p[c][x] + cis * q[x])));   // ds_add_u32
          }
        }
      }
    } else if (e < nvalid) {
      int* qrow = q_lds + li * LD + 4 * j;
      float4v q[C];
      constexpr bool UNSCALED = FIX && !OWN_U && ARITH == MFX_ARITH_F32 && VAR == 0;   // the hogTrain branch folds 2^-24 into its scalars
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int4 qi = *(const int4*)(qrow + c * 4 * L);
        if (UNSCALED) q[c] = float4v{(float)qi.x, (float)qi.y, (float)qi.z, (float)qi.w};
        else if (FIX) q[c] = float4v{(float)qi.x * FIX_INV, (float)qi.y * FIX_INV, (float)qi.z * FIX_INV, (float)qi.w * FIX_INV};
        else q[c] = __builtin_bit_cast(float4v, qi);
      }
      // p = the row from global memory, q = the owned row; the reference updates the USER row first
      // and the item row with the updated user row (modelMF.cpp:94-103) whichever side is owned
      float est;
      if constexpr (VAR == 2) {            // dimensions k < rank only (modelDropoutSigmoid.cpp:174)
        float a = 0.0f;
#pragma unroll
        for (int c = 0; c < C; c++)
#pragma unroll
          for (int x = 0; x < 4; x++)
            if (c * 4 * L + 4 * j + x < var) a = __builtin_fmaf(p[c][x], q[c][x], a);
        est = group_sum<L>(a);
      } else {
        est = group_dot<L, C>(p, q);
      }
      if constexpr (VAR != 0) {
        static_assert(VAR == 0 || !OWN_U, "the sibling models run with owned item rows");
        float4v q0[C];
#pragma unroll
        for (int c = 0; c < C; c++) q0[c] = q[c];
        double m2;
        if constexpr (VAR == 1) m2 = (-2.0 * (double)__builtin_bit_cast(float, var)) * ((double)r - (double)est);   // -2.0*wt*diff
        else { const float d = r - est; m2 = -2.0 * (double)d; }                                                  // float diff
        const double ru = 2.0 * (double)uReg, ri = 2.0 * (double)iReg, lrd = (double)lr;
        const int lim = VAR == 2 ? var : 4 * L * C;
#pragma unroll
        for (int c = 0; c < C; c++) {
          float4v rk = float4v{0.0f, 0.0f, 0.0f, 0.0f};
          if constexpr (VAR == 3) rk = *(const float4v*)(regk + c * 4 * L + 4 * j);   // 2.0*((sing_a+1)/(sing_b+sigma_k)), both sides
#pragma unroll
          for (int x = 0; x < 4; x++)
            if (c * 4 * L + 4 * j + x < lim) p[c][x] = upd_ref64(p[c][x], q[c][x], m2, VAR == 3 ? 2.0 * (double)rk[x] : ru, lrd);
#pragma unroll
          for (int x = 0; x < 4; x++)
            if (c * 4 * L + 4 * j + x < lim) q[c][x] = upd_ref64(q[c][x], p[c][x], m2, VAR == 3 ? 2.0 * (double)rk[x] : ri, lrd);
          if (c * 4 * L + 4 * j < lim) {
            Um.stb(pe, c * 16 * L, p[c]);
            if (FIX) {
#pragma unroll
              for (int x = 0; x < 4; x++)
                if (c * 4 * L + 4 * j + x < lim)
                  atomicAdd(qrow + c * 4 * L + x, __float2int_rn((q[c][x] - q0[c][x]) * FIX_SCALE));   // ds_add_u32
            } else {
              *(int4*)(qrow + c * 4 * L) = __builtin_bit_cast(int4, q[c]);
            }
          }
        }
      } else if constexpr (FIX && !OWN_U && ARITH == MFX_ARITH_F32) {
        // hogTrain's arithmetic (modelMF.cpp:1755-1762) with the item step taken as a delta: the new item
        // row would be q - t, t = lr*(c1*p' + ci*q); the owner copy receives round(-t * 2^24) directly.
        // q here is the UNSCALED float(Q) (see the load above): q_true = q * 2^-24 exactly, so est_true = est * 2^-24,
        // c1*q_true = (c1 * 2^-24)*q and ci*q_true = (ci * 2^-24)*q bit for bit (powers of two commute with every rounding)
        const float c1 = -2.0f * (r - est * FIX_INV), cu = 2.0f * uReg;
        const float c1s = c1 * FIX_INV, cis = (2.0f * iReg) * FIX_INV;
        // the delta round(-(lr * X) * 2^24) is taken as round((lr * -2^24) * X): scaling by a power of two commutes with the
        // rounding of the product, so the integer is the same (where lr * X is subnormal both forms round to 0)
        const float lrs = lr * -FIX_SCALE;
#pragma unroll
        for (int c = 0; c < C; c++) {
#pragma unroll
          for (int x = 0; x < 4; x++) p[c][x] = upd_f32(p[c][x], q[c][x], c1s, cu, lr);
#if MFX_EXP == 3 || MFX_EXP == 5
#pragma unroll
          for (int y = 0; y < (MFX_EXP == 3 ? 4 : 8); y++)
#pragma unroll
            for (int x = 0; x < 4; x++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(q[c][x]) : "v"(1.0f));
#endif
#if MFX_EXP != 1
          Um.stb(pe, c * 16 * L, p[c]);
#endif
#pragma unroll
          for (int x = 0; x < 4; x++) {
            const float t = lrs * (c1 * p[c][x] + cis * q[c][x]);
#if MFX_EXP != 2
            atomicAdd(qrow + c * 4 * L + x, fix_round(t));   // ds_add_u32
#else
            asm volatile("" ::"v"(fix_round(t)));
#endif
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
          Um.stb(pe, c * 16 * L, p[c]);
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
      SlotSteps<L, C, ARITH, OWN_U, FIX, VAR, S + 1>::run(Um, q_lds, tx, ty, tz, tw, regk, g, j, nvalid, lr, uReg, iReg, pn, pen, pnn, penn);
  }
};

// C = 1 (K <= 64): 2 workgroups per CU (<= 64 VGPRs); wider ranks keep 1 workgroup per CU and get 128 VGPRs
//
// SWEEP = false: round `round` of the epoch; the workgroups on XCD x (HW_REG_XCC_ID) take tile
// (x, (x + round) mod 8), user rows stay in that XCD's L2 (sc1 loads, plain stores).
// SWEEP = true: the placement-independent drain of whatever the rounds left (normally nothing), ONE launch of
// DRAIN_WGS resident workgroups.  Every workgroup reads the tile counters, all meet at a grid barrier (nobody has
// pulled a slot yet, so all of them see the same counters) and leave when nothing is left: the normal case costs one
// launch and one barrier.  Otherwise they walk the same diagonals as the rounds, x taken from the workgroup index and a
// grid barrier between diagonals, so that an item row still has ONE owner at a time whatever the XCC_ID values were
// (a partition mode that reports a constant id leaves 56 of the 64 tiles to this path); the user rows go through
// memory (sc1 loads and write-through stores: the workgroups of one tile may sit on several XCDs).
// It also carries the two test hooks of include/mfx.h: `tile_only` >= 0 restricts a launch to that tile and `one`
// makes ONE lane group visit a slot's ratings one at a time (MFX_SGD_F_ONE_GROUP).
constexpr int DRAIN_WGS = 128;        // at most; the launch takes what is resident on THIS device (launch_slots)
// one wave: did the rounds leave a slot anywhere?  -> ctr[NTILE + 3]
static __global__ void slots_left_kernel(const int32_t* __restrict__ tile_slot, unsigned* __restrict__ ctr) {
  bool left = false;
  for (int t = threadIdx.x; t < NTILE; t += 64) left = left || ctr[t] < (unsigned)(tile_slot[t + 1] - tile_slot[t]);
  const unsigned long long any = __builtin_amdgcn_ballot_w64(left);
  const int t = threadIdx.x;
  if (t == 0) ctr[NTILE + 3] = any != 0ull ? 1u : 0u;
}
// ALLW: every wave of the workgroup takes part (compile-time chunk stride; a run-time stride cost 3.6 % at C2 -- measured A/B on one
// box: 19.2 vs 19.9 G updates/s); otherwise `aw` waves do (small tiles, see mfx_launch_sgd_tiled).
//
// PERSIST (round 3, an experiment that did not pay: MFX_SGD_PERSIST=1): the eight rounds in ONE launch.  A round only needs its
// NEIGHBOUR: in round r XCD x owns item block (x + r) mod 8, which XCD x + 1 owned in round r - 1, and user block x, which is its own
// throughout.  So the workgroups on XCD x walk the rounds by themselves; a slot starts when each of its <= 64 item rows has received
// the visits that precede this slot's in round order (rowver[row] >= slot_need[entry]; a visit is finished when the row's
// write-through stores are acknowledged).  The item rows change XCD between rounds, so their staging loads and write-backs go
// through memory (sc1: L1-bypass loads, write-through stores -- 2 x 7 MB per round at C2); the user rows stay in the XCD's L2 as
// before.  Needs every workgroup resident and all eight XCC ids populated (launch_slots checks both, else the eight launches); a
// waiter that sees no progress for 2 s raises the abort flags.  Measured at C2: with no waits at all (wrong on
// purpose) 23.0 G updates/s; with a hand-off per tile 19.2, per row 21.0 - 21.2, against 21.6 - 21.9 for the eight launches.
template <int L, int C, int ARITH, bool SWEEP, bool OWN_U, int VAR, bool ALLW = false, bool PERSIST = false>
__global__ __launch_bounds__(WG, (4 * L * C <= 64 ? 8 : 4)) void sgd_slots_kernel(const int4* __restrict__ rec,
                                                           const int64_t* __restrict__ slot_beg,
                                                           const int32_t* __restrict__ slot_ibeg,
                                                           const int32_t* __restrict__ slot_items,
                                                           const int32_t* __restrict__ tile_slot, unsigned* ctr,
                                                           int round, float* Oth, float* Own, uint32_t obytes,
                                                           float lr, float uReg, float iReg, uint32_t k0,
                                                           uint32_t k1, const int32_t* __restrict__ attr,
                                                           unsigned* __restrict__ visit, int tile_only, int one, int aw,
                                                           uint32_t ownbytes, unsigned* rowver, const uint8_t* __restrict__ slot_need) {
  static_assert(!PERSIST || (!SWEEP && SUB == 1), "the one-launch epoch walks the eight XCD rounds");
  constexpr int G = 64 / L;
  constexpr int LD = 4 * L * C;
  constexpr int LD4 = LD / 4;
  constexpr int ROWS = SlotRows<4 * L * C>::value;
  __shared__ __attribute__((aligned(16))) int q_lds[ROWS * LD];
  __shared__ int s_slot, s_bad;
  // Oth: the lock-free side (user rows when item rows are owned, and vice versa); Own: staged in LDS
  const Rows<SWEEP ? 1 : 3> Um(Oth, obytes);
  const Rows<PERSIST ? 1 : 0> Om(Own, ownbytes);     // the owned rows' table: staging loads and write-backs of a slot
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane / L, j = lane % L;
  const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7);  // HW_REG_XCC_ID[3:0]
  // round r: XCD x takes user block x*SUB + r%SUB and item block (x + r/SUB) mod 8
  const int x = SWEEP ? (int)(blockIdx.x & 7) : xcc;
  const bool onegrp = SWEEP && one != 0;
  const bool drain = SWEEP && tile_only < 0;          // the whole-epoch drain: all diagonals, grid barriers in between
  int4* q4 = (int4*)q_lds;
#if MFX_EXP == 8
  const unsigned long long exp_t0 = wall_clock64();
  unsigned exp_slots = 0, exp_ratings = 0;
#endif
  if (drain) {
    // anything left anywhere?  slots_left_kernel has looked (ctr[NTILE + 3], written behind the rounds and in front of this launch):
    // normally nothing is, and the drain costs a launch without a grid barrier (with the barrier: 18 us of a 950 us epoch at C2)
    if (__hip_atomic_load(&ctr[NTILE + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    // (ctr[NTILE], ctr[NTILE + 1]: barrier counter and abort flag, zeroed with the counters)
    if (tid == 0) s_bad = 0;
    __syncthreads();
    if (tid < NTILE && __hip_atomic_load(&ctr[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(tile_slot[tid + 1] - tile_slot[tid]))
      s_bad = 1;
    __syncthreads();
    const bool left = s_bad != 0;
    if (grid_barrier<0>(ctr + NTILE, gridDim.x)) {
      // gave up: ctr[NTILE + 2] is the STICKY copy of the abort flag -- the per-epoch memset does not touch it, the host clears it
      // only when it has reported it (mfx_slots_check_abort)
      if (tid == 0) __hip_atomic_store(&ctr[NTILE + 2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (!left) return;
  }
  const int r_end = (drain || PERSIST) ? NUB : round + 1;
  for (int rr = (drain || PERSIST) ? 0 : round; rr < r_end; rr++) {
    const int tile = (SWEEP && tile_only >= 0) ? tile_only : (x * SUB + rr % SUB) * 8 + ((x + rr / SUB) & 7);
    const int s0 = tile_slot[tile], ns = tile_slot[tile + 1] - s0;
    bool skip = false;
    if (SWEEP) {  // nothing left in this tile: do not queue on its counter
      if (tid == 0) s_slot = (int)__hip_atomic_load(&ctr[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      skip = s_slot >= ns;
      __syncthreads();
    }
    // PERSIST: the next slot is pulled when the ratings of the one at work are done, and the finished-visit counts of its rows are
    // requested then, in front of the write-back -- a check at the moment of use is a dependent round trip to memory in front of
    // every staging (2-3 us of a 25 us slot: measured as the whole loss of the first cut); here it travels while the write-back's
    // stores are acknowledged.  A count that was not there yet is polled when the slot's turn comes.  (Pulling a whole slot ahead
    // was tried: with five slots per workgroup and tile, a slot held back is a fifth of the tile's balance -- 17.5 G updates/s.)
    int ahead = 0;                       // PERSIST: the slot pulled ahead (index in the tile)
    unsigned pre_ver = 0, pre_need = 0;  // ... and this thread's row of it: finished visits as of the request, visits it needs
    int pre_row = -1;
    int pub_row = -1;                    // PERSIST: the row of the slot just written back that this thread still has to publish
    if (PERSIST && !skip) {
      if (tid == 0) s_slot = (int)atomicAdd(&ctr[tile], 1u);
      __syncthreads();
      ahead = s_slot;
      __syncthreads();
      if (ahead < ns) {
        const int ib0 = slot_ibeg[s0 + ahead], ni0 = slot_ibeg[s0 + ahead + 1] - ib0;
        if (tid < ni0) { pre_row = slot_items[ib0 + tid]; pre_need = (unsigned)slot_need[ib0 + tid]; pre_ver = __hip_atomic_load(&rowver[pre_row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
    }
    while (!skip) {
      int sl;
      if (PERSIST) {
        sl = ahead;
        if (tid == 0) s_bad = 0;           // (the wait below holds a barrier in front of the staging)
      } else {
        if (tid == 0) { s_slot = (int)atomicAdd(&ctr[tile], 1u); s_bad = 0; }
        __syncthreads();
        sl = s_slot;
        __syncthreads();
      }
      if (sl >= ns) break;
      const int slot = s0 + sl;
      const int64_t rb = slot_beg[slot], R = slot_beg[slot + 1] - rb;
      const int ib = slot_ibeg[slot], ni = slot_ibeg[slot + 1] - ib;
      const int my_row = PERSIST ? pre_row : -1;
      if (PERSIST && tile_only != -2) {      // (tile_only == -2: MFX_PERSIST_NOWAIT=1, a timing diagnostic with wrong results)
        // every owned row of the slot must have received its visits of the EARLIER rounds (their owners sit on other XCDs and run
        // on their own): row i carries a count of finished visits, the slot list says how many precede this one (slot_need)
        const bool wait_mine = tid < ni;
        unsigned ver = pre_ver;
        const long long t0 = wall_clock64();
        for (;;) {
          const bool ok = !wait_mine || ver >= pre_need;
          if (__syncthreads_and(ok)) break;
          __builtin_amdgcn_s_sleep(2);
          if (wait_mine) ver = __hip_atomic_load(&rowver[my_row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bool ab = false;
          if (tid == 0) {
            ab = __hip_atomic_load(&ctr[NTILE + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (!ab && wall_clock64() - t0 > 200000000LL) {         // 100 MHz constant clock: 2 s without the rows
              __hip_atomic_store(&ctr[NTILE + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(&ctr[NTILE + 2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ab = true;
            }
          }
          if (__syncthreads_or(ab)) return;
        }
      }
#if MFX_EXP == 8
      exp_slots++; exp_ratings += (unsigned)R;
#endif
      // stage the slot's item rows (as floats first, to decide the representation)
      bool mybad = false;
      for (int x = tid; x < ni * LD4; x += WG) {
        const int row = x / LD4, c4 = x % LD4;
#if MFX_EXP == 7
        const float4v v = __builtin_nontemporal_load((const float4v*)(Own + (int64_t)slot_items[ib + row] * LD + 4 * c4));
#else
        const float4v v = Om.ld((int64_t)slot_items[ib + row] * LD + 4 * c4);
#endif
#pragma unroll
        for (int e = 0; e < 4; e++) mybad |= !(__builtin_fabsf(v[e]) <= FIX_MAX);   // also true for NaN
        q4[row * LD4 + c4] = __builtin_bit_cast(int4, v);
      }
      if (mybad) s_bad = 1;
      // PERSIST: the rows of the slot written back before this one are published HERE, one staging later: their write-through stores
      // have been acknowledged by now without anybody waiting for it (a wait at the end of every slot cost 6 %)
      if (PERSIST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (PERSIST && pub_row >= 0) __hip_atomic_fetch_add(&rowver[pub_row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
      // (A popular row's slot is a POLE: one owner, however many ratings -- C2: 6 200 of the most popular item per tile, ended about when
      //  its round does.  Raised wave priority for such slots, s_setprio 3 from 512 ... 4 096 ratings on, was measured on one box:
      //  21.12 - 21.29 G updates/s against 21.16 - 21.25 without: nothing.)
      // one group (test hook): wave 0 alone, L ratings per chunk, rating cb+s in entry s*G (group 0's entry of step s)
      // aw = waves of the workgroup that take part (16 unless the tile has few lock-free rows: mfx_launch_sgd_tiled)
      const int64_t cb0 = onegrp ? (wave == 0 ? 0 : R) : ALLW ? (int64_t)wave * 64 : (wave < aw ? (int64_t)wave * 64 : R);
      const int64_t cstep = onegrp ? L : ALLW ? (WG / 64) * 64 : (int64_t)aw * 64;
      for (int64_t cb = cb0; cb < R; cb += cstep) {
        const int64_t t = onegrp ? cb + lane / G : cb + lane;
        const bool ok = onegrp ? (lane % G == 0 && t < R) : t < R;
        int4 rc4 = make_int4(0, 0, 0, 0);
        int tw = 0;
        if (ok) {
          const int64_t src = rb + slot_perm(t, R, ks0, ks1);
#if MFX_EXP == 6 || MFX_EXP == 7
          {
            typedef int int4v __attribute__((ext_vector_type(4)));
            const int4v t4 = __builtin_nontemporal_load((const int4v*)&rec[src]);
            rc4 = make_int4(t4.x, t4.y, t4.z, t4.w);
          }
#else
          rc4 = rec[src];
#endif
          if (VAR == 1 || VAR == 2) tw = attr[src];
#if MFX_EXP != 8
          if (visit) atomicAdd(&visit[src], 1u);       // MFX_SGD_F_COUNT_VISITS
#endif
        }
        int nvalid = (int)(R - cb < 64 ? R - cb : 64);
        if (onegrp) nvalid = g == 0 ? (int)(R - cb < L ? R - cb : L) * G : 0;
        // L == 16: a group is one DPP row.  Transpose the chunk once (3 ds_bpermute) so that lane s of row g
        // holds entry s*G+g; every step then takes its entry with a row_share broadcast (no LDS traffic).
        int tx = rc4.x, ty = rc4.y, tz = rc4.z;
        if (L == 16) {
          const int src = (lane & 15) * G + (lane >> 4);
          tx = __shfl(rc4.x, src, 64); ty = __shfl(rc4.y, src, 64); tz = __shfl(rc4.z, src, 64);
          if (VAR == 1 || VAR == 2) tw = __shfl(tw, src, 64);
        }
        // software pipeline: the lock-free rows of steps s+1 and s+2 are requested before step s is computed
        float4v pn[C], pnn[C];
        uint32_t pen = (uint32_t)slot_take<L, 0>(tx, g) * (uint32_t)(LD * 4) + 16u * (uint32_t)j;
        uint32_t penn = (uint32_t)slot_take<L, 1>(tx, g) * (uint32_t)(LD * 4) + 16u * (uint32_t)j;
        if (g < nvalid) {
#pragma unroll
          for (int c = 0; c < C; c++) pn[c] = MFX_EXP == 4 ? float4v{0.01f, 0.02f, 0.03f, 0.01f} : Um.ldb(pen, c * 16 * L);
        }
        if ((C <= 2 || (!OWN_U && ARITH == MFX_ARITH_F32 && VAR == 0)) && G + g < nvalid) {   // two-steps-ahead pipelines only
#pragma unroll
          for (int c = 0; c < C; c++) pnn[c] = MFX_EXP == 4 ? float4v{0.01f, 0.02f, 0.03f, 0.01f} : Um.ldb(penn, c * 16 * L);
        }
        const float* regk = VAR == 3 ? (const float*)attr : nullptr;
        if (fix) SlotSteps<L, C, ARITH, OWN_U, true, VAR, 0>::run(Um, q_lds, tx, ty, tz, tw, regk, g, j, nvalid, lr, uReg, iReg, pn, pen, pnn, penn);
        else SlotSteps<L, C, ARITH, OWN_U, false, VAR, 0>::run(Um, q_lds, tx, ty, tz, tw, regk, g, j, nvalid, lr, uReg, iReg, pn, pen, pnn, penn);
      }
      if (PERSIST && tid == 0) s_slot = (int)atomicAdd(&ctr[tile], 1u);
      __syncthreads();
      if (PERSIST) {
        ahead = s_slot;
        pre_row = -1; pre_ver = 0; pre_need = 0;
        if (ahead < ns) {
          const int ibn = slot_ibeg[s0 + ahead], nin = slot_ibeg[s0 + ahead + 1] - ibn;
          if (tid < nin) { pre_row = slot_items[ibn + tid]; pre_need = (unsigned)slot_need[ibn + tid]; pre_ver = __hip_atomic_load(&rowver[pre_row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
      }
      // write the item rows back (this workgroup is their only owner during the round)
      for (int x = tid; x < ni * LD4; x += WG) {
        const int row = x / LD4, c4 = x % LD4;
        const int4 qi = q4[x];
        const int64_t vrow = (int64_t)slot_items[ib + row] * LD + 4 * c4;
        float4v v;
        if (fix) {
          // new row = staged fp32 row + the accumulated fixed-point delta: an untouched row is
          // written back bit for bit, a touched one carries only the rounding of its updates
          const float4v v0 = Om.ld(vrow);
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
        Om.st(vrow, v);
      }
      __syncthreads();
      if (PERSIST) pub_row = tid < ni ? my_row : -1;
    }
    if (PERSIST) {       // the last slot this workgroup worked off in the tile: its rows may change XCD when every wave's stores are acknowledged
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (pub_row >= 0) __hip_atomic_fetch_add(&rowver[pub_row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the next diagonal re-owns these item rows from other workgroups (other XCDs): L2 write-back, barrier, invalidate
    if (drain && rr + 1 < r_end && grid_barrier<0>(ctr + NTILE, (unsigned)(rr + 2) * gridDim.x)) {
      if (tid == 0) __hip_atomic_store(&ctr[NTILE + 2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
#if MFX_EXP == 8
  if (!SWEEP && visit && tid == 0) {
    unsigned* o = visit + ((size_t)round * gridDim.x + blockIdx.x) * 4;
    o[0] = (unsigned)exp_t0; o[1] = (unsigned)wall_clock64(); o[2] = (unsigned)xcc | exp_slots << 8; o[3] = exp_ratings;
  }
#endif
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------
template <int L, int C, int ARITH, bool OWN_U, int VAR>
static int launch_slots(mfx_ctx* ctx, SlotList* S, const mfx_sgd_opts* o, int blocks, uint32_t k0, uint32_t k1) {
  float* oth = OWN_U ? ctx->V : ctx->U;
  float* own = OWN_U ? ctx->U : ctx->V;
  const uint64_t ob = (uint64_t)(OWN_U ? ctx->nI : ctx->nU) * ctx->ld * 4;
  const int32_t* at = VAR == 3 ? (const int32_t*)ctx->dimreg : (VAR ? S->attr : (const int32_t*)nullptr);
  unsigned* visit = (o->flags & MFX_SGD_F_COUNT_VISITS) ? S->visit : nullptr;
  HIPCHK(hipMemsetAsync(S->ctr, 0, (NTILE + 2) * sizeof(unsigned), ctx->stream));
  if (o->flags & MFX_SGD_F_ONE_GROUP) {   // test hook: tiles in order, one workgroup, one lane group
    for (int tile = 0; tile < NTILE; tile++)
      hipLaunchKernelGGL((sgd_slots_kernel<L, C, ARITH, true, OWN_U, VAR>), dim3(1), dim3(WG), 0, ctx->stream,
                         (const int4*)S->rec, S->slot_beg, S->slot_ibeg, S->slot_items, S->tile_slot, S->ctr, 0,
                         oth, own, (uint32_t)ob, o->learnRate, o->uReg, o->iReg, k0, k1, at, visit, tile, 1, 1, 0u, nullptr, nullptr);
    HIPCHK(hipGetLastError());
    return MFX_OK;
  }
  // the eight rounds in one launch (PERSIST, see the kernel) -- an EXPERIMENT, MFX_SGD_PERSIST=1: measured 8 % slower than the eight
  // launches at C2 (19.2 vs 20.9 G updates/s on one box).  Needs the plain update with all waves at work, every workgroup resident,
  // all eight XCC ids populated by this grid, the owned table within a buffer descriptor's reach.
  bool persist = false;
  if constexpr (VAR == 0 && SUB == 1 && !OWN_U) {       // (user rows owned: it is the LOCK-FREE rows that would change XCD between rounds)
    const uint64_t ownb = (uint64_t)(OWN_U ? ctx->nU : ctx->nI) * ctx->ld * 4;
    const char* pe = getenv("MFX_SGD_PERSIST");
    if (S->active_waves == WG / 64 && ownb < (1ull << 32) && pe && pe[0] == '1' && !(o->flags & MFX_SGD_F_DRAIN_ONLY)) {
      static int resident_dev[64];                       // per instantiation and device (+ 1; 0 = not asked yet): workgroups of the persistent kernel it holds
      int& resident_p1 = resident_dev[ctx->device & 63];
      int resident = resident_p1 - 1;
      if (resident < 0) {
        int per_cu = 0, dev = 0, cus = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sgd_slots_kernel<L, C, ARITH, false, OWN_U, VAR, true, true>, WG, 0));
        HIPCHK(hipGetDevice(&dev));
        HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        resident = std::max(0, per_cu) * std::max(0, cus);
        resident_p1 = resident + 1;
      }
      int rc = 0;
      persist = blocks <= resident && mfx_xcc_ids_populated(ctx, blocks, &rc);
      if (rc) return rc;
    }
    if (persist) {
      const int64_t nown = OWN_U ? ctx->nU : ctx->nI;
      int rc = mfx_slots_build_needs(ctx, S, nown);
      if (rc) return rc;
      HIPCHK(hipMemsetAsync(S->rowver, 0, sizeof(unsigned) * (size_t)nown, ctx->stream));
      ProfScope ps(ctx, MFX_K_SGD);
      hipLaunchKernelGGL((sgd_slots_kernel<L, C, ARITH, false, OWN_U, VAR, true, true>), dim3(blocks), dim3(WG), 0, ctx->stream,
                         (const int4*)S->rec, S->slot_beg, S->slot_ibeg, S->slot_items, S->tile_slot, S->ctr, 0,
                         oth, own, (uint32_t)ob, o->learnRate, o->uReg, o->iReg, k0, k1, at, visit, getenv("MFX_PERSIST_NOWAIT") ? -2 : -1, 0,
                         S->active_waves, (uint32_t)ownb, S->rowver, (const uint8_t*)S->slot_need);
    }
  }
  // the rounds of an epoch in a fresh random order (sgd_slots.h, "Tilings": the same cyclic order of pairings epoch after epoch is
  // half of what a static tiling costs); any order of the diagonals visits every tile once.  MFX_SGD_ROUND_PERM=0: 0, 1, 2, ...
  int rorder[NUB];
  for (int r = 0; r < NUB; r++) rorder[r] = r;
  {
    static const bool perm_on = [] { const char* e = getenv("MFX_SGD_ROUND_PERM"); return !e || atoi(e) != 0; }();
    uint32_t h = mfx_mix32(k0 ^ 0x7f4a7c15U) ^ k1;
    for (int r = NUB - 1; r > 0 && perm_on; r--) {
      h = mfx_mix32(h + 0x9e3779b9U);
      std::swap(rorder[r], rorder[(int)(h % (uint32_t)(r + 1))]);
    }
  }
  for (int rix = 0; rix < NUB && !persist && !(o->flags & MFX_SGD_F_DRAIN_ONLY); rix++) {
    const int round = rorder[rix];
    ProfScope ps(ctx, MFX_K_SGD);
    if (S->active_waves == WG / 64)
      hipLaunchKernelGGL((sgd_slots_kernel<L, C, ARITH, false, OWN_U, VAR, true>), dim3(blocks), dim3(WG), 0, ctx->stream,
                         (const int4*)S->rec, S->slot_beg, S->slot_ibeg, S->slot_items, S->tile_slot, S->ctr, round,
                         oth, own, (uint32_t)ob, o->learnRate, o->uReg, o->iReg, k0, k1, at, visit, -1, 0, S->active_waves, 0u, nullptr, nullptr);
    else
      hipLaunchKernelGGL((sgd_slots_kernel<L, C, ARITH, false, OWN_U, VAR, false>), dim3(blocks), dim3(WG), 0, ctx->stream,
                         (const int4*)S->rec, S->slot_beg, S->slot_ibeg, S->slot_items, S->tile_slot, S->ctr, round,
                         oth, own, (uint32_t)ob, o->learnRate, o->uReg, o->iReg, k0, k1, at, visit, -1, 0, S->active_waves, 0u, nullptr, nullptr);
  }
  {
    // the drain: one launch; the same diagonals keyed on the workgroup index (an item row keeps a single owner).  Its grid
    // barriers need every workgroup RESIDENT: the grid is what this device (or partition: a CPX / QPX partition has 32 - 64
    // CUs) holds of this kernel, a multiple of 8 (the diagonals are keyed on blockIdx & 7), at most DRAIN_WGS.
    static int drain_wgs_dev[64];                      // per instantiation and device (a process may drive a partitioned and a whole device)
    int& drain_wgs = drain_wgs_dev[ctx->device & 63];
    if (drain_wgs == 0) {
      int per_cu = 0, dev = 0, cus = 0;
      HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sgd_slots_kernel<L, C, ARITH, true, OWN_U, VAR>, WG, 0));
      HIPCHK(hipGetDevice(&dev));
      HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      const int fit = std::min(DRAIN_WGS, std::max(0, per_cu) * std::max(0, cus)) / 8 * 8;
      NEED(fit >= 8, MFX_E_STATE, "MFX_SGD_TILED: fewer than 8 workgroups of the drain are resident on this device (%d per CU x %d CUs)", per_cu, cus);
      if (const char* e = getenv("MFX_SGD_DRAIN_WGS")) drain_wgs = std::max(8, std::min(fit, atoi(e) / 8 * 8));   // test knob
      else drain_wgs = fit;
    }
    ProfScope ps(ctx, MFX_K_SGD_SWEEP);
    hipLaunchKernelGGL(slots_left_kernel, dim3(1), dim3(64), 0, ctx->stream, (const int32_t*)S->tile_slot, S->ctr);
    hipLaunchKernelGGL((sgd_slots_kernel<L, C, ARITH, true, OWN_U, VAR>), dim3(drain_wgs), dim3(WG), 0, ctx->stream,
                       (const int4*)S->rec, S->slot_beg, S->slot_ibeg, S->slot_items, S->tile_slot, S->ctr, 0,
                       oth, own, (uint32_t)ob, o->learnRate, o->uReg, o->iReg, k0, k1, at, visit, -1, 0, S->active_waves, 0u, nullptr, nullptr);
  }
  HIPCHK(hipGetLastError());
  // a drain whose barrier gave up (2 s without progress: the device is shared with another resident kernel) leaves a STICKY
  // flag in ctr[NTILE + 2] (the per-epoch memset above clears ctr[0 .. NTILE + 2) only); its copy is looked at behind every
  // stream synchronisation of the C entry points and at the start of the next epoch (mfx_slots_check_abort)
  if (S->abort_host) HIPCHK(hipMemcpyAsync(S->abort_host, S->ctr + NTILE + 2, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  return MFX_OK;
}

template <int L, int C>
static int launch_arith(mfx_ctx* ctx, SlotList* S, int side, const mfx_sgd_opts* o, int blocks, uint32_t k0,
                        uint32_t k1) {
  // sibling models (S->var): owned item rows only, their own arithmetic
  if constexpr (C <= 4) {      // K <= 256, as the flat kernels of sgd_ifw.hip / sgd_tmf.hip
    if (S->var == 1) return launch_slots<L, C, MFX_ARITH_REF64, false, 1>(ctx, S, o, blocks, k0, k1);
    if (S->var == 2) return launch_slots<L, C, MFX_ARITH_REF64F, false, 2>(ctx, S, o, blocks, k0, k1);
    if (S->var == 3) return launch_slots<L, C, MFX_ARITH_REF64F, false, 3>(ctx, S, o, blocks, k0, k1);
  } else if (S->var != 0) {
    return mfx_fail(ctx, MFX_E_ARG, "MFX_SGD_TILED: rating weights / truncated ranks are built for K <= 256");
  }
#define MFX_SIDE(A) \
  (side ? launch_slots<L, C, A, true, 0>(ctx, S, o, blocks, k0, k1) : launch_slots<L, C, A, false, 0>(ctx, S, o, blocks, k0, k1))
  switch (o->arith) {
    case MFX_ARITH_REF64: return MFX_SIDE(MFX_ARITH_REF64);
    case MFX_ARITH_REF64F: return MFX_SIDE(MFX_ARITH_REF64F);
    default: return MFX_SIDE(MFX_ARITH_F32);
  }
#undef MFX_SIDE
}

// one entry point per rank shape, defined by MFX_SLOTS_INSTANCE in sgd_slots_inst_<L>x<C>.hip
#define MFX_SLOTS_INSTANCE(LL, CC)                                                                                      \
  int mfx_slots_launch_##LL##x##CC(mfx_ctx* ctx, SlotList* S, int side, const mfx_sgd_opts* o, int blocks, uint32_t k0, \
                                   uint32_t k1) {                                                                       \
    return launch_arith<LL, CC>(ctx, S, side, o, blocks, k0, k1);                                                       \
  }

#endif
