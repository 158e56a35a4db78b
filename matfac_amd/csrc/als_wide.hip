// als_wide.hip -- ALS half-sweep for 64 < K <= 256 (modelMF.cpp:805-841 users, :844-880 items).
//
// The 64-wide kernel of als.hip keeps the whole Gramian of a row in one wavefront's accumulators and factorises
// it in registers; a 128..256-wide Gramian does not fit.  Here the K x K matrix is cut into 64 x 64 blocks:
//   phase A  one wavefront per (row segment, block pair I >= J): the same gathered-row MFMA accumulation
//            (v_mfma_f32_32x32x2_f32, two ratings per step), a from block I of y, b from block J; the
//            right-hand side rides on the diagonal pairs.  Only the lower block triangle is formed -- the
//            reference computes both triangles (:821-826) but they are bit-identical mirrors.
//   phase B  one workgroup of (KP/8)^2 threads per row: sums the row's segment partials into a lower triangle
//            held 2D-cyclically in registers (36 per thread), adds reg to the diagonal (:831-833), unpivoted LDL^T
//            with the pivot column broadcast through a double-buffered LDS vector (one barrier per step), then L
//            and D go to LDS (K = 256: 132 KB) for the forward and the column-oriented backward substitution.
// Segments are processed in batches so that the partials never exceed ~8 GB whatever the matrix size.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "mfx_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

namespace {
constexpr int BLK = 64 * 64;
struct WideSide {
  int32_t *wrow = nullptr, *wfirst = nullptr, *wn = nullptr;   // rows in segment order: id, first segment, segments
  std::vector<int64_t> batch_seg, batch_row;                    // batch boundaries (segments, rows)
  int64_t nrows = 0, nseg = 0;
  bool built = false;
};
struct WideState {
  WideSide side[2];
  float* slabs = nullptr;
  size_t slab_floats = 0;
  float* bws = nullptr;          // blocked solver: per-wave workspace
  size_t bws_floats = 0;
};
WideState* wst(mfx_ctx* ctx) { return (WideState*)ctx->als_wide; }
}  // namespace

void mfx_als_wide_free_internal(mfx_ctx* ctx) {
  WideState* s = wst(ctx);
  if (!s) return;
  for (WideSide& w : s->side) { dev_free(w.wrow); dev_free(w.wfirst); dev_free(w.wn); }
  dev_free(s->slabs);
  dev_free(s->bws);
  delete s;
  ctx->als_wide = nullptr;
}

// ---------------------------------------------------------------------------
// phase A
// ---------------------------------------------------------------------------
// The accumulation is the software-pipelined loop of als.hip (gram_accumulate): batches of WB steps, the gathered
// loads of batch q+1 in flight while the MFMAs of batch q run, no load behind a lane predicate, 32-bit row offsets
// unless the table is 4 GB or more (BIG).  DIAG: a == b (pairs on the block diagonal), half the loads, and the
// right-hand side rides along.
namespace {
constexpr int WB = 8;
struct WideBatch {
  float a0[WB], a1[WB], b0[WB], b1[WB], r[WB];
};
struct WideAcc {
  f32x16 t00, t01, t10, t11;
  float b0, b1;
};
}  // namespace

template <bool DIAG, bool BIG>
__device__ __forceinline__ void alsw_load_batch(WideBatch& w, int q, int mj, float mr, const float* __restrict__ Y, int ldbytes, int oa,
                                                int ob, int lane) {
  const int half = lane >> 5, idx = lane & 31;
  int j[WB];
  const int addr = ((((q * (2 * WB)) & 63) + half) << 2);     // one address per batch, the steps are constant offsets
#pragma unroll
  for (int t = 0; t < WB; t++) {
    j[t] = __builtin_amdgcn_ds_bpermute(addr + 8 * t, mj);
    w.r[t] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr + 8 * t, __builtin_bit_cast(int, mr)));
  }
#pragma unroll
  for (int t = 0; t < WB; t++) {
    const float* y;
    if (BIG) y = (const float*)((const char*)Y + (uint64_t)(uint32_t)j[t] * (uint32_t)ldbytes);
    else y = (const float*)((const char*)Y + (uint32_t)((uint32_t)j[t] * (uint32_t)ldbytes));
    w.a0[t] = y[oa + idx];
    w.a1[t] = y[oa + 32 + idx];
    if (!DIAG) {
      w.b0[t] = y[ob + idx];
      w.b1[t] = y[ob + 32 + idx];
    }
  }
}
// NS steps, no compare and no select: a rating that must not count was pointed at the table's zero row (als.hip)
template <bool DIAG, int NS>
__device__ __forceinline__ void alsw_mfma_batch(WideAcc& g, const WideBatch& w) {
#pragma unroll
  for (int t = 0; t < NS; t++) {
    const float a0 = w.a0[t], a1 = w.a1[t];
    const float y0 = DIAG ? w.a0[t] : w.b0[t], y1 = DIAG ? w.a1[t] : w.b1[t];
    g.t00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, y0, g.t00, 0, 0, 0);
    if (!DIAG) g.t01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, y1, g.t01, 0, 0, 0);   // above the diagonal: never read
    g.t10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, y0, g.t10, 0, 0, 0);
    g.t11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, y1, g.t11, 0, 0, 0);
    if (DIAG) {
      g.b0 = __builtin_fmaf(w.r[t], a0, g.b0);
      g.b1 = __builtin_fmaf(w.r[t], a1, g.b1);
    }
  }
}
template <bool DIAG, bool BIG>
__device__ __forceinline__ void alsw_accumulate(WideAcc& g, const float* __restrict__ Y, const int32_t* __restrict__ ind,
                                                const float* __restrict__ val, int64_t beg, int64_t end, int ldbytes, int oa, int ob,
                                                int lane, int zrow) {
  if (end <= beg) return;
  const int len = (int)(end - beg);
  const int nbatch = (len + 2 * WB - 1) / (2 * WB);
  int mj, mjn;
  float mr, mrn;
  // ratings <= 0 are skipped (modelMF.cpp:819,857) and the last index block is padded: both gather the zero row
  auto load_block = [&](int64_t off, int& j, float& r) {
    const bool ok = off < len;
    const float v = ok ? val[beg + off] : 0.0f;
    const bool use = v > 0.0f;
    j = use ? ind[beg + (ok ? off : 0)] : zrow;
    r = use ? v : 0.0f;
  };
  load_block(lane, mj, mr);
  load_block(64 + lane, mjn, mrn);
  auto advance = [&](int p) {
    if ((p & 3) == 0) {
      mj = mjn;
      mr = mrn;
      load_block((int64_t)(p / 4 + 1) * 64 + lane, mjn, mrn);
    }
  };
  WideBatch wa, wb;
  alsw_load_batch<DIAG, BIG>(wa, 0, mj, mr, Y, ldbytes, oa, ob, lane);
  int q = 0;
  for (; q + 2 < nbatch; q += 2) {
    advance(q + 1);
    alsw_load_batch<DIAG, BIG>(wb, q + 1, mj, mr, Y, ldbytes, oa, ob, lane);
    alsw_mfma_batch<DIAG, WB>(g, wa);
    advance(q + 2);
    alsw_load_batch<DIAG, BIG>(wa, q + 2, mj, mr, Y, ldbytes, oa, ob, lane);
    alsw_mfma_batch<DIAG, WB>(g, wb);
  }
  const bool short_tail = len - (nbatch - 1) * (2 * WB) <= WB;   // the last batch runs 4 or 8 steps (unused ones add zeros)
  if (q + 1 < nbatch) {
    advance(q + 1);
    alsw_load_batch<DIAG, BIG>(wb, q + 1, mj, mr, Y, ldbytes, oa, ob, lane);
    alsw_mfma_batch<DIAG, WB>(g, wa);
    if (short_tail) alsw_mfma_batch<DIAG, WB / 2>(g, wb);
    else alsw_mfma_batch<DIAG, WB>(g, wb);
  } else {
    if (short_tail) alsw_mfma_batch<DIAG, WB / 2>(g, wa);
    else alsw_mfma_batch<DIAG, WB>(g, wa);
  }
}

template <bool BIG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void alsw_gram_kernel(
    const int64_t* __restrict__ seg_beg, const int64_t* __restrict__ seg_end, int64_t seg0, int64_t nseg, int npairs,
    const int32_t* __restrict__ ind, const float* __restrict__ val, const float* __restrict__ Y, int ld, float* __restrict__ slabs,
    int64_t stride, int zrow) {
  const int lane = threadIdx.x, half = lane >> 5, idx = lane & 31;
  for (int64_t u = blockIdx.x; u < nseg * npairs; u += gridDim.x) {
    const int64_t s = u / npairs;
    const int p = (int)(u - s * npairs);
    int I = 0;
    while ((I + 1) * (I + 2) / 2 <= p) I++;
    const int J = p - I * (I + 1) / 2;
    WideAcc g;
#pragma unroll
    for (int r = 0; r < 16; r++) { g.t00[r] = 0.0f; g.t01[r] = 0.0f; g.t10[r] = 0.0f; g.t11[r] = 0.0f; }
    g.b0 = g.b1 = 0.0f;
    const int64_t beg = seg_beg[seg0 + s], end = seg_end[seg0 + s];
    if (I == J) alsw_accumulate<true, BIG>(g, Y, ind, val, beg, end, ld * (int)sizeof(float), 64 * I, 64 * J, lane, zrow);
    else alsw_accumulate<false, BIG>(g, Y, ind, val, beg, end, ld * (int)sizeof(float), 64 * I, 64 * J, lane, zrow);
    float* o = slabs + s * stride + (int64_t)p * BLK + lane;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      o[(0 * 16 + r) * 64] = g.t00[r];
      if (I != J) o[(1 * 16 + r) * 64] = g.t01[r];
      o[(2 * 16 + r) * 64] = g.t10[r];
      o[(3 * 16 + r) * 64] = g.t11[r];
    }
    if (I == J) {   // right-hand side of block I: even + odd rating halves
      const float s0 = g.b0 + __shfl_xor(g.b0, 32, 64), s1 = g.b1 + __shfl_xor(g.b1, 32, 64);
      float* ob_ = slabs + s * stride + (int64_t)npairs * BLK + 64 * I;
      if (half == 0) { ob_[idx] = s0; ob_[32 + idx] = s1; }
    }
  }
}

// ---------------------------------------------------------------------------
// phase B
// ---------------------------------------------------------------------------
// position of element (gi, gj), gj <= gi, inside a segment's partials (the MFMA accumulator layout of phase A)
__device__ __forceinline__ int alsw_slab_offset(int gi, int gj) {
  const int I = gi >> 6, J = gj >> 6, rr = gi & 63, cc = gj & 63;
  const int p = I * (I + 1) / 2 + J, t = (rr >> 5) * 2 + (cc >> 5), r32 = rr & 31, c32 = cc & 31;
  const int r = (r32 & 3) + 4 * (r32 >> 3), lane = c32 + 32 * ((r32 >> 2) & 1);
  return p * BLK + (t * 16 + r) * 64 + lane;
}

// One workgroup of P*P threads per row, KP = 8P.  The lower triangle lives in REGISTERS, 2D-cyclic: thread
// (ti, tj) owns the elements (ti + P*a, tj + P*b), a >= b -- 36 registers.  Step k of the right-looking LDL^T:
// the owners of column k publish it (unscaled) through a double-buffered LDS vector, one barrier, then every
// thread updates its own elements with 8 + 8 LDS reads.  The column loop is blocked by P so that every register
// index is a compile-time constant.  L and D are then written to LDS for the two substitutions.
template <int P>
__global__ __launch_bounds__(P * P) void alsw_solve_kernel(const int32_t* __restrict__ wrow, const int32_t* __restrict__ wfirst,
                                                          const int32_t* __restrict__ wn, int64_t row0, int64_t nrows, int64_t seg0,
                                                          const float* __restrict__ slabs, int64_t stride, int npairs, int K,
                                                          int ld, float reg, float* __restrict__ X) {
  constexpr int KP = 8 * P;
  extern __shared__ float lds[];
  float* Lp = lds;                              // packed strictly-lower L (+ unused diagonal slots)
  float* colbuf = Lp + KP * (KP + 1) / 2;       // [2][KP]
  float* dvec = colbuf + 2 * KP;                // [KP]

  const int tid = threadIdx.x, ti = tid / P, tj = tid % P;
  for (int64_t m = blockIdx.x; m < nrows; m += gridDim.x) {
    const int row = wrow[row0 + m], n = wn[row0 + m];
    const int64_t first = (int64_t)wfirst[row0 + m] - seg0;
    f32x2s A2[8][4];   // A(a, c) = element (ti + P a, tj + P c), c <= a; pairs of columns for v_pk_fma_f32
#define A(a, c) A2[a][(c) >> 1][(c) & 1]
    float z = 0.0f;
#pragma unroll
    for (int a = 0; a < 8; a++)
#pragma unroll
      for (int b = 0; b <= a; b++) A(a, b) = 0.0f;
#pragma nounroll
    for (int sg = 0; sg < n; sg++) {   // segment order: reproducible
      const float* sl = slabs + (first + sg) * stride;
#pragma unroll
      for (int a = 0; a < 8; a++)
#pragma unroll
        for (int b = 0; b <= a; b++) {
          const int gi = ti + P * a, gj = tj + P * b;
          if (gj <= gi) A(a, b) += sl[alsw_slab_offset(gi, gj)];
        }
      if (tid < KP) z += sl[(int64_t)npairs * BLK + tid];
    }
    // YTY(j,j) += reg for j < K (modelMF.cpp:831-833); padded dimensions become identity rows
    if (ti == tj) {
#pragma unroll
      for (int a = 0; a < 8; a++) A(a, a) = (ti + P * a) < K ? A(a, a) + reg : 1.0f;
    }
#pragma unroll
    for (int b = 0; b < 8; b++) {
#pragma nounroll
      for (int kk = 0; kk < P; kk++) {
        const int k = b * P + kk;
        float* buf = colbuf + (k & 1) * KP;
        if (tj == kk) {
#pragma unroll
          for (int a = b; a < 8; a++) buf[ti + P * a] = A(a, b);
        }
        __syncthreads();
        const float dk = buf[k];
        float rdk = __builtin_amdgcn_rcpf(dk);                           // reciprocal + one Newton step (as als.hip)
        rdk = __builtin_fmaf(__builtin_fmaf(-dk, rdk, 1.0f), rdk, rdk);
        float li[8], cj[8];
#pragma unroll
        for (int a = b; a < 8; a++) li[a] = buf[ti + P * a] * rdk;
#pragma unroll
        for (int c = b; c < 8; c++) cj[c] = buf[tj + P * c];
        // element (ti + P a, tj + P c) is below column k and inside the lower triangle: decided at compile time
        // except in the block column of k (c == b: tj > kk) and on the diagonal blocks (a == c: tj <= ti)
        const bool right_of_k = tj > kk, on_or_below = tj <= ti;
        // in the block column of k only the threads right of k take part: they see the pivot column, the others a zero
        // (A - l * 0 is A), so that column needs one select instead of one per element
        const float cjb = right_of_k ? cj[b] : 0.0f;
#pragma unroll
        for (int a = b; a < 8; a++)
#pragma unroll
          for (int c = b; c <= a; c++) {
            // (all three conditions are compile-time constants once a, b, c are unrolled)
            if ((c & 1) == 1 && c - 1 >= b && c < a) continue;            // went with its even neighbour
            if ((c & 1) == 0 && c + 1 < a) {                              // two columns strictly below the diagonal blocks
              const f32x2s nl = {-li[a], -li[a]};
              const f32x2s v = {c == b ? cjb : cj[c], cj[c + 1 < 8 ? c + 1 : 0]};
              A2[a][c >> 1] = __builtin_elementwise_fma(nl, v, A2[a][c >> 1]);
              continue;
            }
            const float upd = __builtin_fmaf(-li[a], c == b ? cjb : cj[c], A(a, c));
            if (a > c) A(a, c) = upd;                         // strictly below the diagonal blocks
            else A(a, c) = on_or_below ? upd : A(a, c);       // diagonal block: lower triangle only
          }
        if (tj == kk) {
#pragma unroll
          for (int a = b; a < 8; a++)
            if (ti + P * a > k) A(a, b) = li[a];
        }
      }
    }
    // L (strictly lower) and D to LDS
#pragma unroll
    for (int a = 0; a < 8; a++)
#pragma unroll
      for (int b = 0; b <= a; b++) {
        const int gi = ti + P * a, gj = tj + P * b;
        if (gj < gi) Lp[gi * (gi + 1) / 2 + gj] = A(a, b);
        else if (gj == gi) dvec[gi] = A(a, b);
      }
    __syncthreads();
    // The substitutions run on ONE wavefront (lane l owns rows l, l+64, ...): the pivot value travels by a
    // cross-lane read instead of a workgroup barrier per step.
    if (tid < KP) colbuf[tid] = z;     // right-hand side, gathered from the threads that summed it
    __syncthreads();
    if (tid < 64) {
      constexpr int Q = KP / 64;
      float zz[Q];
#pragma unroll
      for (int q = 0; q < Q; q++) zz[q] = colbuf[tid + 64 * q];
      // L y = b
#pragma unroll
      for (int q = 0; q < Q; q++) {
#pragma unroll 8
        for (int kk = 0; kk < 64; kk++) {   // unrolled so that the L reads of the next steps are in flight; the pivot is a v_readlane
          const int k = 64 * q + kk;
          const float zk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zz[q]), kk));
#pragma unroll
          for (int q2 = q; q2 < Q; q2++) {
            const int i = tid + 64 * q2;
            if (i > k) zz[q2] = zz[q2] - Lp[i * (i + 1) / 2 + k] * zk;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < Q; q++) zz[q] = zz[q] / dvec[tid + 64 * q];
      // L^T x = y, one column of L^T (= row j of L) per step
#pragma unroll
      for (int q = Q - 1; q >= 0; q--) {
#pragma unroll 8
        for (int jj = 63; jj >= 0; jj--) {
          const int j = 64 * q + jj;
          const float xj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zz[q]), jj));
          const float* Lj = Lp + j * (j + 1) / 2;
#pragma unroll
          for (int q2 = 0; q2 <= q; q2++) {
            const int i = tid + 64 * q2;
            if (i < j) zz[q2] = zz[q2] - Lj[i] * xj;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < Q; q++) {
        const int i = tid + 64 * q;
        if (i < K) X[(int64_t)row * ld + i] = zz[q];
      }
    }
    __syncthreads();
  }
}
#undef A


// ---------------------------------------------------------------------------
// phase B, blocked: ONE WAVEFRONT PER ROW, 64 x 64 blocks, the block products on the f32 MFMA
// ---------------------------------------------------------------------------
// Left-looking block LDL^T of the K x K system (K = 64 C).  For block column J:
//   * every block (I, J), I >= J, is loaded in the accumulator layout phase A wrote it in and receives
//         A(I,J) -= sum_{k<J} (L(I,k) D_k) L(J,k)^T                      -- 32 steps of v_mfma_f32_32x32x2_f32 per tile and k;
//     both operands come from the wave's workspace in HBM/L2, where finished blocks are kept ROW-major; MFMA step t sums the
//     columns t (lower half of the wave) and 32 + t (upper half), so a lane reads 32 consecutive columns of its rows idx and
//     32 + idx, four per load;
//   * the block goes through a padded LDS image (64 x 65 floats per wave) to change layout: the diagonal block to
//     "lane i owns row i" (its missing upper-right tile is read transposed: the block is symmetric), an off-diagonal block
//     to "lane j owns COLUMN j";
//   * diagonal block: unpivoted right-looking LDL^T in registers, the block's right-hand side riding along (as als.hip);
//     afterwards lane k holds L(k, m<k), d_k and the pivot row d_k L(j>k, k) -- kept in registers for the panel below and
//     written to the workspace for the back substitution;
//   * off-diagonal block (the panel): lane i owns row i of B; X (D L^T) = B is solved row by row -- step j fixes x_j d_j = b_j and
//     the later columns lose x_j (d_j L(c,j)), the scalar being entry c of lane j's pivot row (one v_readlane per (step, column));
//     lane i then writes its row of L(I,J) and of -L(I,J) D_J (the copy the MFMA's first operand reads) as 16 wide stores each.
// Forward substitution happens on the way (a block's right-hand side is corrected with the lane's own rows of the finished
// blocks), the back substitution walks the block columns from the last: w = L(I,J)^T x_I reads a column of every row.
namespace {
constexpr int IMG_LD = 65;                 // row stride of the LDS image (floats): row- and column-wise reads are conflict-free
struct BlkAcc { f32x16 t00, t01, t10, t11; };

// acc (4 tiles; t01 not valid when sym) -> LDS image, then lane l takes row l (by_col = false) or column l (by_col = true)
__device__ __forceinline__ void alsb_relayout(const BlkAcc& g, float* img, int lane, bool sym, bool by_col, float (&v)[64]) {
  const int h = lane >> 5, c = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
    img[rr * IMG_LD + c] = g.t00[r];
    if (!sym) img[rr * IMG_LD + 32 + c] = g.t01[r];
    img[(32 + rr) * IMG_LD + c] = g.t10[r];
    img[(32 + rr) * IMG_LD + 32 + c] = g.t11[r];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (by_col) {
#pragma unroll
    for (int q = 0; q < 64; q++) v[q] = img[q * IMG_LD + lane];
  } else {
#pragma unroll
    for (int q = 0; q < 64; q++) {
      // symmetric block: A(l, q) for l < 32 <= q was not accumulated; it equals A(q, l)
      const bool mirror = sym && lane < 32 && q >= 32;
      v[q] = mirror ? img[q * IMG_LD + lane] : img[lane * IMG_LD + q];
    }
  }
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float alsb_bcast(float x, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
}  // namespace

namespace {
// One buffer descriptor per wave-uniform base pointer: the lane offset sits in ONE vector register and everything else in the
// scalar / immediate offset of the instruction.  (Plain pointer arithmetic made the compiler keep ~64 64-bit addresses per
// block alive and spill them: 1.2 KB of scratch per lane and a third of the kernel's time.)
// (The scalar-offset operand is used here with offsets of a few hundred KB at most: the descriptor's base is the wave's own workspace or
//  slab.  The tiled SGD kernel lost half-rows when it handed a chunk constant to the SGPR offset of accesses whose VECTOR offset was a
//  row's byte offset in a 1.28 GB table -- sgd_common.h: the rows that failed were those at byte offsets >= 2^30.  Nothing here comes
//  near that; the launcher checks the extents.)
struct BufF {
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ BufF(const float* base) { rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffffc, 0x00020000); }
  __device__ __forceinline__ float ld(int voff_bytes, int soff_bytes) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff_bytes, soff_bytes, 0));
  }
  __device__ __forceinline__ f32x4w ld4(int voff_bytes, int soff_bytes) const {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rs, voff_bytes, soff_bytes, 0));
  }
  __device__ __forceinline__ void st(float v, int voff_bytes, int soff_bytes) const {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff_bytes, soff_bytes, 0);
  }
  __device__ __forceinline__ void st4(f32x4w v, int voff_bytes, int soff_bytes) const {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rs, voff_bytes, soff_bytes, 0);
  }
};
}  // namespace

template <int C>
__global__ __launch_bounds__(256, 2) void alsw_bsolve_kernel(const int32_t* __restrict__ wrow, const int32_t* __restrict__ wfirst,
                                                             const int32_t* __restrict__ wn, int64_t row0, int64_t nrows, int64_t seg0,
                                                             const float* __restrict__ slabs, int64_t stride, int K, int ld, float reg,
                                                             float* __restrict__ X, float* __restrict__ wsp, int64_t ws_stride, int dbg) {
  constexpr int NPAIRS = C * (C + 1) / 2;
  __shared__ float img_all[4][64 * IMG_LD];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, idx = lane & 31;
  float* img = img_all[wv];
  const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
  // workspace of this wave: off-diagonal blocks row-major, L and the pre-scaled -L D; diagonal blocks as factored rows
  const BufF W(wsp + wave * ws_stride);
  // byte offsets inside the workspace
  auto LT = [&](int I, int J) { return (I * (I - 1) / 2 + J) * (2 * BLK) * 4; };          // I > J
  auto LTD = [&](int I, int J) { return ((I * (I - 1) / 2 + J) * (2 * BLK) + BLK) * 4; };
  constexpr int DG = (C * (C - 1) / 2) * (2 * BLK) * 4;                                   // [C][64 lanes][64]
  constexpr int VEC = DG + C * BLK * 4;                                                   // z, d, x per block column: [3][C][64]
  const int l4 = lane * 4, l256 = lane * 256;
  // The block loops are REAL loops (the 64-step factorisation and panel bodies are unrolled once each, not once per block: the
  // first version, unrolled over the blocks too, was 77 k instructions at C = 4 and ran out of the instruction cache).
  for (int64_t m = wave; m < nrows; m += nwaves) {
    const int row = wrow[row0 + m], nsg = wn[row0 + m];
    const float* sl0 = slabs + ((int64_t)wfirst[row0 + m] - seg0) * stride;
    const int sbytes = (int)(stride * 4);
#pragma unroll 1
    for (int J = 0; J < C; J++) {
      // ---- right-hand side of block J, corrected by the finished block columns: r = b_J - sum_k L(J,k) z_k
      float rJ = 0.0f;
      for (int sg = 0; sg < nsg; sg++) rJ += BufF(sl0 + sg * stride).ld(l4, (NPAIRS * BLK + 64 * J) * 4);
#pragma unroll 1
      for (int k = 0; k < J; k++) {
        const int lt = LT(J, k);                      // row `lane` of L(J,k)
        const float zk = W.ld(l4, VEC + (0 * C + k) * 256);
#pragma unroll 4
        for (int q = 0; q < 64; q += 4) {
          const f32x4w v = W.ld4(l256, lt + q * 4);
          rJ = __builtin_fmaf(-v[0], alsb_bcast(zk, q), rJ);
          rJ = __builtin_fmaf(-v[1], alsb_bcast(zk, q + 1), rJ);
          rJ = __builtin_fmaf(-v[2], alsb_bcast(zk, q + 2), rJ);
          rJ = __builtin_fmaf(-v[3], alsb_bcast(zk, q + 3), rJ);
        }
      }
      float a[64];     // the factored diagonal block of this column: lane l = row l (see above)
      float dJ = 1.0f;
#pragma unroll 1
      for (int I = J; I < C; I++) {
        // ---- A(I,J): sum of the segment partials (segment order: reproducible)
        BlkAcc g;
        {
          const int p = I * (I + 1) / 2 + J;
#pragma unroll
          for (int r = 0; r < 16; r++) { g.t00[r] = 0.0f; g.t01[r] = 0.0f; g.t10[r] = 0.0f; g.t11[r] = 0.0f; }
          for (int sg = 0; sg < nsg; sg++) {
            const BufF S(sl0 + sg * stride);
            const int po = p * BLK * 4;
#pragma unroll
            for (int r = 0; r < 16; r++) {
              g.t00[r] += S.ld(l4, po + (0 * 16 + r) * 256);
              g.t01[r] += I != J ? S.ld(l4, po + (1 * 16 + r) * 256) : 0.0f;
              g.t10[r] += S.ld(l4, po + (2 * 16 + r) * 256);
              g.t11[r] += S.ld(l4, po + (3 * 16 + r) * 256);
            }
          }
        }
        // ---- minus the contributions of the finished block columns (the diagonal block computes its unused tile too: one code path)
#pragma unroll 1
        for (int k = 0; k < ((dbg & 4) ? 0 : J); k++) {
          const int pa = LTD(I, k);         // -D_k L(I,k), row-major
          const int pb = LT(J, k);          // L(J,k)
          // MFMA step t sums the two columns t (lower half of the wave) and 32 + t (upper half): a lane reads 32 consecutive
          // columns of its rows idx and 32 + idx, FOUR per load -- 32 wide loads per block product instead of 128 single ones
          const int vo = (idx * 64 + half * 32) * 4;
#pragma unroll 2
          for (int tq = 0; tq < 8; tq++) {
            const f32x4w a0 = W.ld4(vo, pa + tq * 16), a1 = W.ld4(vo, pa + 8192 + tq * 16);
            const f32x4w y0 = W.ld4(vo, pb + tq * 16), y1 = W.ld4(vo, pb + 8192 + tq * 16);
#pragma unroll
            for (int e = 0; e < 4; e++) {
              g.t00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], y0[e], g.t00, 0, 0, 0);
              g.t01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], y1[e], g.t01, 0, 0, 0);
              g.t10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], y0[e], g.t10, 0, 0, 0);
              g.t11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], y1[e], g.t11, 0, 0, 0);
            }
          }
        }
        if (I == J) {
          alsb_relayout(g, img, lane, true, false, a);
          // YTY(j,j) += reg for j < K (modelMF.cpp:831-833); padded dimensions (all zero) become identity rows
          const float regv = (64 * J + lane) < K ? reg : 1.0f;
#pragma unroll
          for (int q = 0; q < 64; q++) a[q] += lane == q ? regv : 0.0f;
          float z = rJ, d = 1.0f;
          if (!(dbg & 1))
#pragma unroll
          for (int k = 0; k < 64; k++) {
            const float dk = alsb_bcast(a[k], k);
            float rk = __builtin_amdgcn_rcpf(dk);
            rk = __builtin_fmaf(__builtin_fmaf(-dk, rk, 1.0f), rk, rk);
            int lv = lane;
            asm volatile("" : "+v"(lv));
            const float lik = lv > k ? a[k] * rk : 0.0f;
#pragma unroll
            for (int q = k + 1; q < 64; q++) a[q] = __builtin_fmaf(-lik, alsb_bcast(a[q], k), a[q]);
            z = __builtin_fmaf(-lik, alsb_bcast(z, k), z);
            a[k] = lv > k ? lik : a[k];
            d = lv == k ? dk : d;
            __builtin_amdgcn_sched_barrier(0);
          }
          dJ = d;
          W.st(z, l4, VEC + (0 * C + J) * 256);
          W.st(d, l4, VEC + (1 * C + J) * 256);
#pragma unroll
          for (int q = 0; q < 64; q += 4) W.st4(f32x4w{a[q], a[q + 1], a[q + 2], a[q + 3]}, l256, DG + J * BLK * 4 + q * 4);
        } else {
          // ---- panel block, lane i owns row i of B: X (D L^T) = B row by row.  Step j: x_j = b_j / d_j; the later columns lose
          // x_j (d_j L(c,j)), and d_j L(c,j) is entry c of lane j's pivot row: one v_readlane per (step, column)
          float y[64];
          alsb_relayout(g, img, lane, false, false, y);
          float rdv = __builtin_amdgcn_rcpf(dJ);
          rdv = __builtin_fmaf(__builtin_fmaf(-dJ, rdv, 1.0f), rdv, rdv);
          const int lt = LT(I, J), ltd = LTD(I, J);
          if (!(dbg & 2))
#pragma unroll
          for (int j = 0; j < 64; j++) {
            const float xj = y[j] * alsb_bcast(rdv, j);             // y[j] = x_j d_j is final from here on
#pragma unroll
            for (int c = j + 1; c < 64; c++) y[c] = __builtin_fmaf(-xj, alsb_bcast(a[c], j), y[c]);
            __builtin_amdgcn_sched_barrier(0);
          }
          // row `lane` of L(I,J) D_J is y; the two copies go out as 16 wide stores each (row-major blocks)
#pragma unroll
          for (int q = 0; q < 64; q += 4) {
            W.st4(f32x4w{-y[q], -y[q + 1], -y[q + 2], -y[q + 3]}, l256, ltd + q * 4);
            W.st4(f32x4w{y[q] * alsb_bcast(rdv, q), y[q + 1] * alsb_bcast(rdv, q + 1), y[q + 2] * alsb_bcast(rdv, q + 2),
                         y[q + 3] * alsb_bcast(rdv, q + 3)},
                  l256, lt + q * 4);
          }
        }
      }
      // the blocks this wave wrote are read back by other lanes in the next block column
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    // ---- back substitution, block columns from the last: D L^T x = z - D w,  w = sum_{I>J} L(I,J)^T x_I
#pragma unroll 1
    for (int J = C - 1; J >= 0; J--) {
      float w = 0.0f;
#pragma unroll 1
      for (int I = J + 1; I < C; I++) {
        const int lt = LT(I, J);                      // column `lane` of L(I,J): element `lane` of every row
        const float xI = W.ld(l4, VEC + (2 * C + I) * 256);
#pragma unroll 8
        for (int i = 0; i < 64; i++) w = __builtin_fmaf(W.ld(l4, lt + i * 256), alsb_bcast(xI, i), w);
      }
      float a[64];
#pragma unroll
      for (int q = 0; q < 64; q += 4) {
        const f32x4w v = W.ld4(l256, DG + J * BLK * 4 + q * 4);
        a[q] = v[0]; a[q + 1] = v[1]; a[q + 2] = v[2]; a[q + 3] = v[3];
      }
      const float d = W.ld(l4, VEC + (1 * C + J) * 256);
      float rd = __builtin_amdgcn_rcpf(d);
      rd = __builtin_fmaf(__builtin_fmaf(-d, rd, 1.0f), rd, rd);
      float acc = __builtin_fmaf(-d, w, W.ld(l4, VEC + (0 * C + J) * 256)), x = 0.0f;
#pragma unroll
      for (int j = 63; j >= 0; j--) {
        const float xv = acc * rd;
        x = lane == j ? xv : x;
        acc = __builtin_fmaf(-a[j], alsb_bcast(xv, j), acc);      // lanes k < j hold d_k L(j,k) in a[j]
      }
      W.st(x, l4, VEC + (2 * C + J) * 256);
      if (64 * J + lane < K) X[(int64_t)row * ld + 64 * J + lane] = x;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------
static int build_wide_side(mfx_ctx* ctx, WideSide& w, const RowSegs& sd, int64_t seg_floats) {
  std::vector<int32_t> seg_row((size_t)sd.nseg);
  if (sd.nseg) HIPCHK(hipMemcpy(seg_row.data(), sd.seg_row, sizeof(int32_t) * seg_row.size(), hipMemcpyDeviceToHost));
  std::vector<int32_t> wrow, wfirst, wn;
  for (int64_t s = 0; s < sd.nseg;) {
    int64_t e = s;
    while (e < sd.nseg && seg_row[e] == seg_row[s]) e++;
    wrow.push_back(seg_row[s]); wfirst.push_back((int32_t)s); wn.push_back((int32_t)(e - s));
    s = e;
  }
  // batches of whole rows whose partials stay under the cap
  const char* cap_env = getenv("MFX_ALS_SLAB_GB");
  const int64_t cap_bytes = (int64_t)((cap_env ? atof(cap_env) : 8.0) * (1 << 30));
  const int64_t cap_seg = std::max<int64_t>(1, cap_bytes / (seg_floats * 4));
  w.batch_seg.assign(1, 0);
  w.batch_row.assign(1, 0);
  int64_t cur = 0;
  for (size_t r = 0; r < wrow.size(); r++) {
    if (cur > 0 && cur + wn[r] > cap_seg) { w.batch_seg.push_back(wfirst[r]); w.batch_row.push_back((int64_t)r); cur = 0; }
    cur += wn[r];
  }
  w.batch_seg.push_back(sd.nseg);
  w.batch_row.push_back((int64_t)wrow.size());
  int rc;
  auto up = [&](int32_t** dst, const std::vector<int32_t>& v) -> int {
    int r = dev_alloc(ctx, dst, v.size());
    if (r) return r;
    if (!v.empty()) HIPCHK(hipMemcpy(*dst, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice));
    return MFX_OK;
  };
  if ((rc = up(&w.wrow, wrow)) || (rc = up(&w.wfirst, wfirst)) || (rc = up(&w.wn, wn))) return rc;
  w.nrows = (int64_t)wrow.size();
  w.nseg = sd.nseg;
  w.built = true;
  return MFX_OK;
}

int mfx_als_wide_half_sweep(mfx_ctx* ctx, int side, float reg) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  const int K = ctx->K, C = (K + 63) / 64, KP = 64 * C, npairs = C * (C + 1) / 2;
  NEED(K > 64 && K <= 256, MFX_E_ARG, "mfx_als_half_sweep: K=%d outside the wide kernel's range (64, 256]", K);
  NEED(ctx->ld >= KP, MFX_E_STATE, "mfx_als_half_sweep: factor rows are not padded to %d", KP);
  NEED(!(side == MFX_SIDE_ITEMS && mfx_sharded(ctx)), MFX_E_ARG, "mfx_als_half_sweep: the sharded item sweep is built for K <= 64");
  if (!ctx->als_wide) ctx->als_wide = new WideState;
  WideState* st = wst(ctx);
  RowSegs* sdp;
  int rc = mfx_get_segments(ctx, side, &sdp);
  if (rc) return rc;
  const int64_t stride = (int64_t)npairs * BLK + KP;
  WideSide& w = st->side[side];
  if (!w.built && (rc = build_wide_side(ctx, w, *sdp, stride))) return rc;
  const int32_t* ind = side == MFX_SIDE_USERS ? m.rowind : m.colind;
  const float* val = side == MFX_SIDE_USERS ? m.rowval : m.colval;
  const float* Y = side == MFX_SIDE_USERS ? ctx->V : ctx->U;
  float* X = side == MFX_SIDE_USERS ? ctx->U : ctx->V;
  const size_t lds = ((size_t)KP * (KP + 1) / 2 + 3 * (size_t)KP + 2) * sizeof(float);
  const void* solve_fn = C == 2 ? (const void*)alsw_solve_kernel<16> : C == 3 ? (const void*)alsw_solve_kernel<24> : (const void*)alsw_solve_kernel<32>;
  HIPCHK(hipFuncSetAttribute(solve_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (size_t b = 0; b + 1 < w.batch_seg.size(); b++) {
    const int64_t s0 = w.batch_seg[b], ns = w.batch_seg[b + 1] - s0;
    const int64_t r0 = w.batch_row[b], nr = w.batch_row[b + 1] - r0;
    if (ns == 0) continue;
    const size_t need = (size_t)ns * (size_t)stride;
    if (need > st->slab_floats) {
      dev_free(st->slabs);
      st->slab_floats = 0;
      if ((rc = dev_alloc(ctx, &st->slabs, need))) return rc;
      st->slab_floats = need;
    }
    {
      ProfScope ps(ctx, MFX_K_ALS_GRAM);
      const int blocks = (int)std::min<int64_t>(ns * npairs, 256 * 32);
      const int64_t yrows = side == MFX_SIDE_USERS ? ctx->nI : ctx->nU;
      if ((yrows + 1) * ctx->ld * (int64_t)sizeof(float) >= ((int64_t)1 << 32))
        hipLaunchKernelGGL(alsw_gram_kernel<true>, dim3(blocks), dim3(64), 0, ctx->stream, sdp->seg_beg, sdp->seg_end, s0, ns, npairs, ind,
                           val, Y, ctx->ld, st->slabs, stride, (int)yrows);
      else
        hipLaunchKernelGGL(alsw_gram_kernel<false>, dim3(blocks), dim3(64), 0, ctx->stream, sdp->seg_beg, sdp->seg_end, s0, ns, npairs, ind,
                           val, Y, ctx->ld, st->slabs, stride, (int)yrows);
      HIPCHK(hipGetLastError());
    }
    // blocked LDL^T (one wavefront per row, block products on the MFMA) from three blocks on: K = 192: 35 vs 60 ms, K = 256: 67 vs
    // 94 ms per iteration at the C2 matrix; at two blocks (K <= 128) the register-resident workgroup kernel below is still ahead
    // (10.9 vs 14.5 ms).  MFX_ALS_SOLVER=blocked / unblocked overrides.
    const char* sv = getenv("MFX_ALS_SOLVER");
    const bool blocked = sv ? sv[0] == 'b' : C >= 3;
    if (blocked) {
      ProfScope ps(ctx, MFX_K_ALS_SOLVE);
      int dev = 0, cus = 0;
      HIPCHK(hipGetDevice(&dev));
      HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      const int blocksb = (int)std::min<int64_t>((nr + 3) / 4, (int64_t)std::max(cus, 1) * 2);
      const char* dbe = getenv("MFX_ALS_DBG");       // timing probes: 1 no factorisation, 2 no panel solve, 4 no block products (results wrong)
      const int dbgb = dbe ? atoi(dbe) : 0;
      const int64_t ws_stride = ((int64_t)C * (C - 1) / 2 * 2 + C) * BLK + 3 * C * 64;
      const size_t needw = (size_t)blocksb * 4 * (size_t)ws_stride;
      NEED((size_t)ws_stride * 4 < ((size_t)1 << 30) && (size_t)stride * 4 < ((size_t)1 << 30), MFX_E_ARG,
           "wide ALS solve: a wave's workspace / slab (%lld / %lld floats) must stay below 2^30 bytes (buffer offsets, see BufF)", (long long)ws_stride,
           (long long)stride);
      if (needw > st->bws_floats) {
        dev_free(st->bws);
        st->bws_floats = 0;
        if ((rc = dev_alloc(ctx, &st->bws, needw))) return rc;
        st->bws_floats = needw;
      }
#define MFX_BSOLVE(CC)                                                                                                         \
  hipLaunchKernelGGL(alsw_bsolve_kernel<CC>, dim3(blocksb), dim3(256), 0, ctx->stream, w.wrow, w.wfirst, w.wn, r0, nr, s0, st->slabs, \
                     stride, K, ctx->ld, reg, X, st->bws, ws_stride, dbgb)
      if (C == 2) MFX_BSOLVE(2); else if (C == 3) MFX_BSOLVE(3); else MFX_BSOLVE(4);
#undef MFX_BSOLVE
      HIPCHK(hipGetLastError());
    } else {
      ProfScope ps(ctx, MFX_K_ALS_SOLVE);
      const int blocks = (int)std::min<int64_t>(nr, 256 * 8);
#define MFX_SOLVE(PP)                                                                                                        \
  hipLaunchKernelGGL(alsw_solve_kernel<PP>, dim3(blocks), dim3(PP * PP), lds, ctx->stream, w.wrow, w.wfirst, w.wn, r0, nr, s0, \
                     st->slabs, stride, npairs, K, ctx->ld, reg, X)
      if (C == 2) MFX_SOLVE(16); else if (C == 3) MFX_SOLVE(24); else MFX_SOLVE(32);
#undef MFX_SOLVE
      HIPCHK(hipGetLastError());
    }
  }
  return MFX_OK;
}
