// als.hip -- ALS half-sweep for gfx950: per-row gathered Gramian on the f32 MFMA
// (v_mfma_f32_32x32x2_f32, exact fp32 fma chain) + in-register LDL^T solve.
//
// Replaces ModelMF::trainALS' two loops (modelMF.cpp:805-841 users, :844-880 items):
//   A = sum_{j in row, r>0} y_j y_j^T  (full K x K, both triangles),  b = sum r y_j,
//   A_jj += reg (not degree-scaled),  x = A^-1 b   (Eigen: YTY.ldlt().solve(b)).
//
// One wavefront per row segment.  Two ratings enter each MFMA step (k = 2): lanes 0-31
// carry y_j of the even rating, lanes 32-63 of the odd one; four 32x32 accumulator
// tiles hold the 64x64 Gramian (K <= 64, zero padded).  Rows longer than SEG ratings
// are split into segments whose partial (A,b) go to a slab and are summed in segment
// order by a second launch, so a 50k-rating item does not serialise the sweep.
// After accumulation lanes l and l^32 swap half-columns so that lane i owns row i of A
// (= column i, A is symmetric); the factorisation then runs entirely in registers with
// v_readlane broadcasts of the pivot row: no LDS, no pivoting (A + reg*I is SPD; Eigen's
// LDLT pivots on the diagonal, which changes round-off only -- see DESIGN.md).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "mfx_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef MFX_ALS_LDS_SPLIT
#define MFX_ALS_LDS_SPLIT 8      // eighths of a step's trailing pairs that stay on v_readlane before the rest comes through LDS (8 = 0 eighths:
                                 // all through LDS, the default; 0: none through LDS, the round-3 solve; + 16: no sched_barrier between steps)
#endif
constexpr int GT_STRIDE = 36;   // row stride of the transposition tile in LDS (floats): 16-byte aligned rows, banks staggered

namespace {
constexpr int SLAB = 66 * 64;     // floats per partial: 4 tiles x 16 regs + 2 rhs, per lane
}  // namespace

__device__ __forceinline__ float rdlane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

struct GramAcc {
  f32x16 t[2][2];
  float b0, b1;
};

__device__ __forceinline__ void gram_zero(GramAcc& g) {
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) g.t[i][j][r] = 0.0f;
  g.b0 = g.b1 = 0.0f;
}

// accumulate ratings [beg,end) of one row: ind[] = counterpart index, val[] = rating.
// Batches of GB steps (2 ratings each); the 2*GB gathered row loads of batch q+1 are issued before the 4*GB MFMAs of
// batch q, so the MFMA pipe works through one batch while the next one's rows are in flight.  LD (the row stride:
// 16, 32 or 64 floats) is a template parameter: the row address is a shift, no load sits behind a lane predicate
// (a predicated load becomes its own basic block and the waits around it can no longer be counted), and for
// LD <= 32 only the first 32x32 tile exists.
namespace {
constexpr int GB = 8;
struct GramBatch {
  f32x2 y[GB];     // (y_j[idx], y_j[32 + idx]) of the step's rating
  float r[GB];
};
}  // namespace

// BIG: the factor table is 4 GB or more and needs 64-bit row offsets; otherwise a row's byte offset is one 32-bit shift
// and the loads use the scalar-base + 32-bit-offset form.  (The f32 MFMA shares the SIMD's issue slots with ordinary
// vector instructions -- measured: their times add -- so every instruction saved per step is MFMA time gained.)
template <int LD, bool BIG>
__device__ __forceinline__ void gram_load_batch(GramBatch& b, int q, int mj, float mr, const float* __restrict__ Y, int lane) {
  const int half = lane >> 5, idx = lane & 31;
  int j[GB];
  // Step t takes lanes 2t (lower half) and 2t+1 (upper half) of the batch's quarter of the index block: one address,
  // constant offsets.
  const int addr = ((((q * (2 * GB)) & 63) + half) << 2);
#pragma unroll
  for (int t = 0; t < GB; t++) {
    j[t] = __builtin_amdgcn_ds_bpermute(addr + 8 * t, mj);
    b.r[t] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr + 8 * t, __builtin_bit_cast(int, mr)));
  }
#pragma unroll
  for (int t = 0; t < GB; t++) {
    const float* y;
    if (BIG) y = Y + (uint64_t)(uint32_t)j[t] * LD;
    else y = (const float*)((const char*)Y + (uint32_t)((uint32_t)j[t] * (uint32_t)(LD * sizeof(float))));
    if (LD == 64) {
      b.y[t][0] = y[idx];
      b.y[t][1] = y[32 + idx];
    } else if (LD == 32) {
      b.y[t][0] = y[idx];
      b.y[t][1] = 0.0f;
    } else {
      const float v = y[idx & (LD - 1)];
      b.y[t][0] = idx < LD ? v : 0.0f;
      b.y[t][1] = 0.0f;
    }
  }
}
// NS steps of a batch, no compare and no select: a rating that must not count was pointed at the zero row
template <int LD, int NS>
__device__ __forceinline__ void gram_mfma_batch(GramAcc& g, const GramBatch& b) {
  f32x2 bb = {g.b0, g.b1};
#pragma unroll
  for (int t = 0; t < NS; t++) {
    const f32x2 y = b.y[t];
    g.t[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y[0], y[0], g.t[0][0], 0, 0, 0);
    if (LD > 32) {
      // the lower-left tile would be the bit-identical mirror of the upper-right one (same products, same order):
      // it is not accumulated; gram_solve reads it out of t[0][1] transposed
      g.t[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y[0], y[1], g.t[0][1], 0, 0, 0);
      g.t[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y[1], y[1], g.t[1][1], 0, 0, 0);
    }
    bb = __builtin_elementwise_fma(f32x2{b.r[t], b.r[t]}, y, bb);
  }
  g.b0 = bb[0];
  g.b1 = bb[1];
}

// zrow: index of the all-zero row behind the table.  Ratings <= 0 are skipped (modelMF.cpp:819,857) and a row's last
// index block is padded: both read the zero row with rating 0, so they add exact zeros to A and b.
template <int LD, bool BIG>
__device__ __forceinline__ void gram_accumulate(GramAcc& g, const float* __restrict__ Y,
                                                const int32_t* __restrict__ ind,
                                                const float* __restrict__ val, int64_t beg, int64_t end,
                                                int lane, int zrow) {
  if (end <= beg) return;
  const int len = (int)(end - beg);                 // a segment is at most SEG ratings
  const int nbatch = (len + 2 * GB - 1) / (2 * GB);
  // index blocks of 64 ratings: the current one and the one after it
  int mj, mjn;
  float mr, mrn;
  auto load_block = [&](int64_t off, int& j, float& r) {
    const bool ok = off < len;
    const float v = ok ? val[beg + off] : 0.0f;
    const bool use = v > 0.0f;
    j = use ? ind[beg + (ok ? off : 0)] : zrow;
    r = use ? v : 0.0f;
  };
  load_block(lane, mj, mr);
  load_block(64 + lane, mjn, mrn);
  // entering batch p: when it starts a new index block, the prefetched block becomes current and the next is requested
  auto advance = [&](int p) {
    if ((p & 3) == 0) {
      mj = mjn;
      mr = mrn;
      load_block((int64_t)(p / 4 + 1) * 64 + lane, mjn, mrn);
    }
  };
  GramBatch ba, bb;
  gram_load_batch<LD, BIG>(ba, 0, mj, mr, Y, lane);
  int q = 0;
  for (; q + 2 < nbatch; q += 2) {      // two full batches per turn, the loads one batch ahead of the MFMAs
    advance(q + 1);
    gram_load_batch<LD, BIG>(bb, q + 1, mj, mr, Y, lane);
    gram_mfma_batch<LD, GB>(g, ba);
    advance(q + 2);
    gram_load_batch<LD, BIG>(ba, q + 2, mj, mr, Y, lane);
    gram_mfma_batch<LD, GB>(g, bb);
  }
  // the row's last batch runs 4 or 8 steps (its unused steps add zeros)
  const bool short_tail = len - (nbatch - 1) * (2 * GB) <= GB;   // wave-uniform
  if (q + 1 < nbatch) {
    advance(q + 1);
    gram_load_batch<LD, BIG>(bb, q + 1, mj, mr, Y, lane);
    gram_mfma_batch<LD, GB>(g, ba);
    if (short_tail) gram_mfma_batch<LD, GB / 2>(g, bb);
    else gram_mfma_batch<LD, GB>(g, bb);
  } else {
    if (short_tail) gram_mfma_batch<LD, GB / 2>(g, ba);
    else gram_mfma_batch<LD, GB>(g, ba);
  }
}


// v_permlane32_swap as inline assembly WITH its wait states: gfx950 wants two between a VALU write of a VGPR and a permlane swap
// that reads it, and two more before a VALU reads what the swap wrote; the compiler's hazard recognizer does not look inside an
// asm statement, and the operands here are "+v" copies the compiler makes right in front of it (`v_mov v145, v62` directly
// followed by the swap: what a round-3 rebuild produced in als_reduce_kernel -- rows with more than one segment came out 5 % off,
// rows solved by als_segment_kernel did not; the round-2 build happened to put the copy elsewhere).
#define MFX_SWAP32(A, B) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(A), "+v"(B))
// Turn the tile layout into "lane i owns row i", add reg, solve, return x_i in lane i.
__device__ __forceinline__ float gram_solve(const GramAcc& g, int K, float reg, int lane, float* tr) {
  // a[2p], a[2p+1] = A(lane, 2p), A(lane, 2p+1): pairs, so that the trailing update is one packed fma per two columns
  f32x2 a[32];
  // Of columns c and 32+c (c = l&31) lane l holds the rows whose bit 2 equals l>>5, its partner l^32 the others.
  // v_permlane32_swap trades the upper half of the first tile column with the lower half of the second: afterwards
  // the first register is row I0 and the second row I0+4 of the lane's own column 32*(l>>5)+c (= row, A is symmetric).
  // (inline asm: the compiler's own handling of the builtin merged most of the 33 swaps of this function into three.
  //  The accumulators were written by MFMAs and asm operands are invisible to the hazard recogniser, hence the nops:
  //  18 wait states cover a 16-pass MFMA result read by a VALU instruction.)
  // Rows 32..63 of the columns < 32 (the tile that was not accumulated) are the transpose of t[0][1]: it goes through
  // LDS (tr: 32 rows of GT_STRIDE floats, one wavefront per workgroup) -- element (row, col) of t[0][1] is written at
  // tr[row][col], lane (h, c) reads row c, columns 8q + 4h .. + 3: the rows 32 + 8q + 4h + e of its own column.
  {
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; r++) tr[((r & 3) + 8 * (r >> 2) + 4 * h) * GT_STRIDE + c] = g.t[0][1][r];
  }
  __syncthreads();
  float tx[16];
  {
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 v = *(const f32x4*)(tr + c * GT_STRIDE + 8 * q + 4 * h);
      tx[4 * q + 0] = v[0]; tx[4 * q + 1] = v[1]; tx[4 * q + 2] = v[2]; tx[4 * q + 3] = v[3];
    }
  }
  __syncthreads();   // tr is reused by the next row of this workgroup
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int I0 = (r & 3) + 8 * (r >> 2);   // row with bit 2 clear; I0+4 has it set
    float lo = g.t[0][0][r], hi = g.t[0][1][r];
    MFX_SWAP32(lo, hi);
    a[I0 >> 1][I0 & 1] = lo;
    a[(I0 + 4) >> 1][I0 & 1] = hi;
    // rows 32 + I0 and 32 + I0 + 4: the lanes of the first 32 columns bring the transposed values (register r of a lane
    // in the lower half is row 32 + I0 of its column, in the upper half row 32 + I0 + 4 of the partner's), the others t[1][1]
    float lo2 = tx[r], hi2 = g.t[1][1][r];
    MFX_SWAP32(lo2, hi2);
    a[(32 + I0) >> 1][I0 & 1] = lo2;
    a[(32 + I0 + 4) >> 1][I0 & 1] = hi2;
  }
  float z0 = g.b0, z1 = g.b1;
  MFX_SWAP32(z0, z1);
  float z = z0 + z1;
  // YTY(j,j) += reg for j < K (modelMF.cpp:831-833); padded dimensions (all zero) become identity rows
  const float regv = lane < K ? reg : 1.0f;
#pragma unroll
  for (int I = 0; I < 64; I++) a[I >> 1][I & 1] += lane == I ? regv : 0.0f;
  // Right-looking LDL^T, row i of A in lane i, no pivoting (A + reg*I is SPD).  Step k: lanes i > k form l_ik =
  // a_ik / d_k and subtract l_ik * (pivot row k, read from lane k) from their row; lanes <= k stand still, so that
  // lane k keeps d_k in a[k] and the pivot row d_k * l_jk (j > k) in a[j] -- column k of L, which the back
  // substitution below wants in one lane.  The right-hand side rides along as one more column (L y = b).
  // (Round 3 wrote the trailing update out with the scalars alternating between two register pairs, so that no fma needs an
  // `s_nop 1` behind its two readlanes -- 1 100 fewer s_nop per solve, bit-identical, 3.98 instead of 3.99 ms per C3 iteration: with
  // two waves per SIMD the partner wave issues in those slots.  What the solve costs is its 2 016 readlanes and 1 008 packed fmas.)
  float d = 1.0f;
  {
#pragma unroll
    for (int k = 0; k < 64; k++) {
      const float dk = rdlane(a[k >> 1][k & 1], k);
      float rk = __builtin_amdgcn_rcpf(dk);
      rk = __builtin_fmaf(__builtin_fmaf(-dk, rk, 1.0f), rk, rk);
      // the lane number is re-materialised per step: with a loop-invariant one the compiler forms all 128 lane masks up
      // front and spills them (two v_readlane per use instead of one v_cmp)
      int lv = lane;
      asm volatile("" : "+v"(lv));
      const float lik = lv > k ? a[k >> 1][k & 1] * rk : 0.0f;
      if ((k & 1) == 0) a[k >> 1][1] = __builtin_fmaf(-lik, rdlane(a[k >> 1][1], k), a[k >> 1][1]);
      const f32x2 l2 = {lik, lik};
#if MFX_ALS_LDS_SPLIT
      // Part of the pivot row through LDS instead of v_readlane: by symmetry element (k, j) is lane j's a[k] -- ONE ds_write_b32 puts
      // the whole row into tr, and the pairs from PL on read it back as broadcasts (two pairs per ds_read_b128) while the VALU does
      // the readlanes of the pairs below PL.  (a_jk and a_kj agree to rounding, not bitwise: the factorisation stays an LU of A to
      // eps.)
      const int P0 = (k >> 1) + 1;                                           // (constants once the k loop is unrolled)
      const int PL = (P0 + (32 - P0) * (MFX_ALS_LDS_SPLIT & 7) / 8 + 1) & ~1;   // even: a ds_read_b128 holds pairs PL, PL + 1 (8: all of them through LDS)
      f32x4 lrow[16];
      if (PL < 32) {
        tr[lane] = a[k >> 1][k & 1];
#pragma unroll
        for (int q = PL; q < 32; q += 2) lrow[(q - PL) / 2] = *(const f32x4*)(tr + 2 * q);
      }
#pragma unroll
      for (int p = P0; p < (PL < 32 ? PL : 32); p++) {
        const f32x2 row = {rdlane(a[p][0], k), rdlane(a[p][1], k)};
        a[p] = __builtin_elementwise_fma(-l2, row, a[p]);
      }
#pragma unroll
      for (int q = PL; q < 32; q += 2) {
        const f32x4 v = lrow[(q - PL) / 2];
        a[q] = __builtin_elementwise_fma(-l2, f32x2{v[0], v[1]}, a[q]);
        a[q + 1] = __builtin_elementwise_fma(-l2, f32x2{v[2], v[3]}, a[q + 1]);
      }
#else
#pragma unroll
      for (int p = (k >> 1) + 1; p < 32; p++) {
        const f32x2 row = {rdlane(a[p][0], k), rdlane(a[p][1], k)};
        a[p] = __builtin_elementwise_fma(-l2, row, a[p]);   // v_pk_fma_f32 with the scalar pair as one source
      }
#endif
      z = __builtin_fmaf(-lik, rdlane(z, k), z);
      a[k >> 1][k & 1] = lv > k ? lik : a[k >> 1][k & 1];
      d = lv == k ? dk : d;
      if (!(MFX_ALS_LDS_SPLIT & 16)) __builtin_amdgcn_sched_barrier(0);   // keep the steps apart: interleaved, each column becomes one dependent chain
    }
  }
  // L^T x = D^-1 y from the last row up: x_j is final in lane j; lanes k < j subtract (d_k l_jk) x_j before their division
  float rd = __builtin_amdgcn_rcpf(d);
  rd = __builtin_fmaf(__builtin_fmaf(-d, rd, 1.0f), rd, rd);
  float acc = z, x = 0.0f;
#pragma unroll
  for (int j = 63; j >= 0; j--) {
    const float xv = acc * rd;
    x = lane == j ? xv : x;
    acc = __builtin_fmaf(-a[j >> 1][j & 1], rdlane(xv, j), acc);
  }
  return x;
}

// store / load the accumulators of one row (SLAB floats, lane-interleaved)
__device__ __forceinline__ void gram_store(const GramAcc& g, float* o) {
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (i == 1 && j == 0) continue;   // not accumulated (mirror of tile (0,1)); its slab slot stays unused
#pragma unroll
      for (int r = 0; r < 16; r++) o[((i * 2 + j) * 16 + r) * 64] = g.t[i][j][r];
    }
  o[64 * 64] = g.b0;
  o[65 * 64] = g.b1;
}
__device__ __forceinline__ void gram_add(GramAcc& g, const float* o) {
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (i == 1 && j == 0) continue;
#pragma unroll
      for (int r = 0; r < 16; r++) g.t[i][j][r] += o[((i * 2 + j) * 16 + r) * 64];
    }
  g.b0 += o[64 * 64];
  g.b1 += o[65 * 64];
}

// SOLVE 1: solve single-segment rows in place; 0: timing probe; 2 (sharded item sweep): emit every row's
// accumulators to grow[row] for the all-reduce over the ranks
template <int SOLVE, int LD, bool BIG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void als_segment_kernel(const int32_t* __restrict__ seg_row,
                                                         const int64_t* __restrict__ seg_beg,
                                                         const int64_t* __restrict__ seg_end,
                                                         const int32_t* __restrict__ seg_slab, int64_t nseg,
                                                         const int32_t* __restrict__ ind,
                                                         const float* __restrict__ val,
                                                         const float* __restrict__ Y, float* __restrict__ X,
                                                         float* __restrict__ slabs, int K, int ld, float reg,
                                                         float* __restrict__ grow, int zrow) {
  __shared__ __attribute__((aligned(16))) float tr[32 * GT_STRIDE];   // gram_solve's transposition tile
  const int lane = threadIdx.x;
  for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
    GramAcc g;
    gram_zero(g);
    gram_accumulate<LD, BIG>(g, Y, ind, val, seg_beg[s], seg_end[s], lane, zrow);
    const int slab = seg_slab[s];
    if (slab < 0) {
      if (SOLVE == 2) {
        gram_store(g, grow + (int64_t)seg_row[s] * SLAB + lane);
      } else if (SOLVE == 1) {
        const float x = gram_solve(g, K, reg, lane, tr);
        if (lane < K) X[(int64_t)seg_row[s] * ld + lane] = x;
      } else if (lane < K) {   // timing probe (MFX_ALS_NOSOLVE): keep the accumulators alive, skip the solve
        X[(int64_t)seg_row[s] * ld + lane] = g.t[0][0][0] + g.t[0][1][1] + g.t[1][0][2] + g.t[1][1][3] + g.b0 + g.b1;
      }
    } else {
      gram_store(g, slabs + (int64_t)slab * SLAB + lane);
    }
  }
}

template <bool EMIT>
__global__ __launch_bounds__(64) void als_reduce_kernel(const int32_t* __restrict__ mrow,
                                                        const int32_t* __restrict__ mrow_first,
                                                        const int32_t* __restrict__ mrow_n, int64_t nmrow,
                                                        const float* __restrict__ slabs, float* __restrict__ X,
                                                        int K, int ld, float reg, float* __restrict__ grow) {
  __shared__ __attribute__((aligned(16))) float tr[32 * GT_STRIDE];   // gram_solve's transposition tile
  const int lane = threadIdx.x;
  for (int64_t m = blockIdx.x; m < nmrow; m += gridDim.x) {
    GramAcc g;
    gram_zero(g);
    const int first = mrow_first[m], n = mrow_n[m];
    for (int s = 0; s < n; s++)   // fixed segment order: reproducible
      gram_add(g, slabs + (int64_t)(first + s) * SLAB + lane);
    if (EMIT) { gram_store(g, grow + (int64_t)mrow[m] * SLAB + lane); continue; }
    const float x = gram_solve(g, K, reg, lane, tr);
    if (lane < K) X[(int64_t)mrow[m] * ld + lane] = x;
  }
}
// sharded item sweep: solve every item some rank has ratings for, from the accumulators summed over the ranks
__global__ __launch_bounds__(64) void als_global_solve_kernel(const float* __restrict__ grow, const double* __restrict__ gcol, int32_t row0,
                                                              int32_t nrows, float* __restrict__ X, int K, int ld, float reg) {
  __shared__ __attribute__((aligned(16))) float tr[32 * GT_STRIDE];   // gram_solve's transposition tile
  const int lane = threadIdx.x;
  for (int64_t r = (int64_t)row0 + blockIdx.x; r < nrows; r += gridDim.x) {
    if (gcol[r] == 0.0) continue;
    GramAcc g;
    gram_zero(g);
    gram_add(g, grow + r * SLAB + lane);
    const float x = gram_solve(g, K, reg, lane, tr);
    if (lane < K) X[r * ld + lane] = x;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T>
static int upload_vec(mfx_ctx* ctx, T** dst, const std::vector<T>& v) {
  int rc = dev_alloc(ctx, dst, v.size());
  if (rc) return rc;
  if (!v.empty()) HIPCHK(hipMemcpyAsync(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, ctx->stream));
  return MFX_OK;
}

static void free_segs(RowSegs& s) {
  dev_free(s.seg_row); dev_free(s.seg_beg); dev_free(s.seg_end); dev_free(s.seg_slab);
  dev_free(s.mrow); dev_free(s.mrow_first); dev_free(s.mrow_n);
  s = RowSegs();
}

void mfx_segs_free_internal(mfx_ctx* ctx) {
  free_segs(ctx->segs[0]);
  free_segs(ctx->segs[1]);
}
void mfx_als_free_internal(mfx_ctx* ctx) {
  mfx_als_wide_free_internal(ctx);
  dev_free(ctx->als_slabs);
  ctx->als_slab_cap = 0;
}

// Build the segment lists of one side from the device row pointers (read back once).
static int build_side(mfx_ctx* ctx, RowSegs& sd, const int64_t* dptr, int32_t n) {
  constexpr int SEG = MFX_SEG;
  std::vector<int64_t> ptr((size_t)n + 1);
  HIPCHK(hipMemcpy(ptr.data(), dptr, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyDeviceToHost));
  std::vector<int32_t> rows;
  rows.reserve(n);
  for (int32_t r = 0; r < n; r++)
    if (ptr[r + 1] > ptr[r]) rows.push_back(r);  // rows without ratings are invalid: left untouched
  std::stable_sort(rows.begin(), rows.end(),
                   [&](int32_t a, int32_t b) { return ptr[a + 1] - ptr[a] > ptr[b + 1] - ptr[b]; });
  std::vector<int32_t> seg_row, seg_slab, mrow, mfirst, mn;
  std::vector<int64_t> seg_beg, seg_end;
  int32_t nslab = 0;
  for (int32_t r : rows) {
    const int64_t b = ptr[r], e = ptr[r + 1], len = e - b;
    if (len <= SEG) {
      seg_row.push_back(r); seg_beg.push_back(b); seg_end.push_back(e); seg_slab.push_back(-1);
    } else {
      const int nsg = (int)((len + SEG - 1) / SEG);
      mrow.push_back(r); mfirst.push_back(nslab); mn.push_back(nsg);
      for (int s = 0; s < nsg; s++) {
        seg_row.push_back(r);
        seg_beg.push_back(b + (int64_t)s * SEG);
        seg_end.push_back(std::min<int64_t>(e, b + (int64_t)(s + 1) * SEG));
        seg_slab.push_back(nslab++);
      }
    }
  }
  int rc;
  if ((rc = upload_vec(ctx, &sd.seg_row, seg_row))) return rc;
  if ((rc = upload_vec(ctx, &sd.seg_beg, seg_beg))) return rc;
  if ((rc = upload_vec(ctx, &sd.seg_end, seg_end))) return rc;
  if ((rc = upload_vec(ctx, &sd.seg_slab, seg_slab))) return rc;
  if ((rc = upload_vec(ctx, &sd.mrow, mrow))) return rc;
  if ((rc = upload_vec(ctx, &sd.mrow_first, mfirst))) return rc;
  if ((rc = upload_vec(ctx, &sd.mrow_n, mn))) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  sd.nseg = (int64_t)seg_row.size();
  sd.nmrow = (int64_t)mrow.size();
  sd.nslab = nslab;
  sd.built = true;
  return MFX_OK;
}

int mfx_get_segments(mfx_ctx* ctx, int side, RowSegs** out) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  RowSegs& sd = ctx->segs[side];
  if (!sd.built) {
    int rc = build_side(ctx, sd, side == 0 ? m.rowptr : m.colptr, side == 0 ? m.nrows : m.ncols);
    if (rc) return rc;
  }
  *out = &sd;
  return MFX_OK;
}

// one launch of the accumulation kernel for the context's row stride
template <int SOLVE>
static int launch_segments(mfx_ctx* ctx, const RowSegs& sd, const int32_t* ind, const float* val, const float* Y, int64_t yrows,
                           float* X, float reg, float* grow) {
  const int blocks = (int)std::min<int64_t>(sd.nseg, 256 * 16);
  const bool big = (yrows + 1) * ctx->ld * (int64_t)sizeof(float) >= ((int64_t)1 << 32);   // gathered table of 4 GB or more
#define MFX_ALS_LAUNCH1(LDV, BIGV)                                                                                              \
  hipLaunchKernelGGL((als_segment_kernel<SOLVE, LDV, BIGV>), dim3(blocks), dim3(64), 0, ctx->stream, sd.seg_row, sd.seg_beg,     \
                     sd.seg_end, sd.seg_slab, sd.nseg, ind, val, Y, X, ctx->als_slabs, ctx->K, ctx->ld, reg, grow, (int)yrows)
#define MFX_ALS_LAUNCH(LDV)                                \
  do {                                                     \
    if (big) MFX_ALS_LAUNCH1(LDV, true);                   \
    else MFX_ALS_LAUNCH1(LDV, false);                      \
  } while (0)
  switch (ctx->ld) {
    case 16: MFX_ALS_LAUNCH(16); break;
    case 32: MFX_ALS_LAUNCH(32); break;
    case 64: MFX_ALS_LAUNCH(64); break;
    default: return mfx_fail(ctx, MFX_E_STATE, "ALS (K <= 64): unexpected row stride %d", ctx->ld);
  }
#undef MFX_ALS_LAUNCH
#undef MFX_ALS_LAUNCH1
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

extern "C" int mfx_als_half_sweep(mfx_ctx* ctx, int side, float reg) {
  if (!ctx) return MFX_E_ARG;
  NEED(side == MFX_SIDE_USERS || side == MFX_SIDE_ITEMS, MFX_E_ARG, "mfx_als_half_sweep: side=%d", side);
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  NEED(m.present && m.has_col, MFX_E_STATE, "mfx_als_half_sweep: train matrix with column view needed");
  NEED(ctx->U, MFX_E_STATE, "mfx_als_half_sweep: no model");
  NEED(ctx->K <= 256, MFX_E_ARG, "mfx_als_half_sweep: this build supports K <= 256 (got %d)", ctx->K);
  NEED(m.nrows <= ctx->nU && m.ncols <= ctx->nI, MFX_E_ARG, "mfx_als_half_sweep: matrix exceeds model");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->K > 64) return mfx_als_wide_half_sweep(ctx, side, reg);   // als_wide.hip
  RowSegs* sdp;
  int rc0 = mfx_get_segments(ctx, side, &sdp);
  if (rc0) return rc0;
  RowSegs& sd = *sdp;
  if (sd.nslab > ctx->als_slab_cap) {
    dev_free(ctx->als_slabs);
    int rc = dev_alloc(ctx, &ctx->als_slabs, (size_t)sd.nslab * SLAB);
    if (rc) return rc;
    ctx->als_slab_cap = sd.nslab;
  }
  const int32_t* ind = side == MFX_SIDE_USERS ? m.rowind : m.colind;
  const float* val = side == MFX_SIDE_USERS ? m.rowval : m.colval;
  const float* Y = side == MFX_SIDE_USERS ? ctx->V : ctx->U;
  float* X = side == MFX_SIDE_USERS ? ctx->U : ctx->V;
  const int64_t yrows = side == MFX_SIDE_USERS ? ctx->nI : ctx->nU;
  if (side == MFX_SIDE_ITEMS && mfx_sharded(ctx)) {
    // An item's users live on several ranks (user-block sharding): every rank accumulates (A, b) over ITS users.  With RCCL the
    // accumulators are REDUCE-SCATTERED -- rank r receives the sums of its slice of ceil(nItems / N) items, solves that slice, and
    // the solved rows are all-gathered ((N - 1) / N of the slab and of V per rank instead of twice the slab, and
    // every item solved once instead of N times).  On a caller-supplied all-reduce (gloo / MPI callers, the two-process tests)
    // the whole slab is summed on every rank and every rank solves every item.
    const double* gcol;
    int rc = mfx_comm_global_col_counts(ctx, &gcol);
    if (rc) return rc;
    const int N = std::max(1, ctx->nranks);
    // (MFX_ALS_ALLREDUCE=1: the whole slab all-reduced and every item solved on every rank, also over RCCL)
    static const bool force_allreduce = [] { const char* e = getenv("MFX_ALS_ALLREDUCE"); return e && atoi(e) != 0; }();
    const bool scatter = mfx_comm_has_rccl(ctx) && !force_allreduce;
    const int64_t per = ((int64_t)m.ncols + N - 1) / N;                  // items per slice (the last ones may be short or empty)
    const size_t gn = scatter ? (size_t)per * N * SLAB : (size_t)m.ncols * SLAB;
    if (!ctx->als_global || ctx->als_global_cap < gn) {
      dev_free(ctx->als_global);
      ctx->als_global_cap = 0;
      if ((rc = dev_alloc(ctx, &ctx->als_global, gn))) return rc;
      ctx->als_global_cap = gn;
    }
    HIPCHK(hipMemsetAsync(ctx->als_global, 0, gn * sizeof(float), ctx->stream));
    if (sd.nseg > 0) {
      ProfScope ps(ctx, MFX_K_ALS_GRAM);
      if ((rc = launch_segments<2>(ctx, sd, ind, val, Y, yrows, X, reg, ctx->als_global))) return rc;
    }
    if (sd.nmrow > 0) {
      hipLaunchKernelGGL(als_reduce_kernel<true>, dim3((int)std::min<int64_t>(sd.nmrow, 256 * 16)), dim3(64), 0, ctx->stream,
                         sd.mrow, sd.mrow_first, sd.mrow_n, sd.nmrow, ctx->als_slabs, X, ctx->K, ctx->ld, reg, ctx->als_global);
      HIPCHK(hipGetLastError());
    }
    if (!scatter) {
      if ((rc = mfx_comm_allreduce(ctx, ctx->als_global, gn, 0))) return rc;
      if (m.ncols > 0) {
        ProfScope ps(ctx, MFX_K_ALS_SOLVE);
        hipLaunchKernelGGL(als_global_solve_kernel, dim3(std::min(m.ncols, 256 * 16)), dim3(64), 0, ctx->stream, ctx->als_global, gcol,
                           0, m.ncols, X, ctx->K, ctx->ld, reg);
        HIPCHK(hipGetLastError());
      }
      return MFX_OK;
    }
    if ((rc = mfx_comm_reduce_scatter(ctx, ctx->als_global, (size_t)per * SLAB))) return rc;
    const int64_t lo = std::min<int64_t>((int64_t)ctx->rank * per, m.ncols), hi = std::min<int64_t>(lo + per, m.ncols);
    if (hi > lo) {
      ProfScope ps(ctx, MFX_K_ALS_SOLVE);
      hipLaunchKernelGGL(als_global_solve_kernel, dim3((unsigned)std::min<int64_t>(hi - lo, 256 * 16)), dim3(64), 0, ctx->stream, ctx->als_global,
                         gcol, (int32_t)lo, (int32_t)hi, X, ctx->K, ctx->ld, reg);
      HIPCHK(hipGetLastError());
    }
    // the solved slices (per x ld floats each, the last ones padded) gathered through a slab of their own, then the first ncols rows
    // go back into V: V itself has nItems + 1 rows, fewer than N slices when the last slice is short
    const size_t slice = (size_t)per * ctx->ld;
    if (ctx->comm_tmp_cap < slice * ((size_t)N + 1)) {
      dev_free(ctx->comm_tmp);
      ctx->comm_tmp_cap = 0;
      if ((rc = dev_alloc(ctx, &ctx->comm_tmp, slice * ((size_t)N + 1)))) return rc;
      ctx->comm_tmp_cap = slice * ((size_t)N + 1);
    }
    float* mine = ctx->comm_tmp;
    float* all = ctx->comm_tmp + slice;
    HIPCHK(hipMemsetAsync(mine, 0, slice * sizeof(float), ctx->stream));
    if (hi > lo) HIPCHK(hipMemcpyAsync(mine, X + (size_t)lo * ctx->ld, (size_t)(hi - lo) * ctx->ld * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = mfx_comm_allgather(ctx, mine, all, slice))) return rc;
    HIPCHK(hipMemcpyAsync(X, all, (size_t)m.ncols * ctx->ld * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    if (!ctx->comm_checked_als) {            // once per communicator: the slices of ceil(ncols / N) items and their offsets are rank arithmetic
      ctx->comm_checked_als = true;
      if ((rc = mfx_comm_check_replicas(ctx, X, (size_t)m.ncols * ctx->ld, "sharded ALS item sweep (reduce-scatter, slice solve, all-gather)"))) return rc;
    }
    return MFX_OK;
  }
  if (sd.nseg > 0) {
    ProfScope ps(ctx, MFX_K_ALS_GRAM);
    static const bool nosolve = getenv("MFX_ALS_NOSOLVE") != nullptr;   // timing probe only
    int rc = nosolve ? launch_segments<0>(ctx, sd, ind, val, Y, yrows, X, reg, nullptr) : launch_segments<1>(ctx, sd, ind, val, Y, yrows, X, reg, nullptr);
    if (rc) return rc;
  }
  if (sd.nmrow > 0) {
    ProfScope ps(ctx, MFX_K_ALS_SOLVE);
    const int blocks = (int)std::min<int64_t>(sd.nmrow, 256 * 16);
    hipLaunchKernelGGL(als_reduce_kernel<false>, dim3(blocks), dim3(64), 0, ctx->stream, sd.mrow, sd.mrow_first,
                       sd.mrow_n, sd.nmrow, ctx->als_slabs, X, ctx->K, ctx->ld, reg, (float*)nullptr);
    HIPCHK(hipGetLastError());
  }
  return MFX_OK;
}
