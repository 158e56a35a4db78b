// sgd.hip -- rank-K SGD update kernels for gfx950 (wave64).
//
// Replaces the inner loops of ModelMF::train (modelMF.cpp:83-105), hogTrain
// (:1747-1763), trainUShuffle (:637-659) and trainSGDPar (:279-302).
//
// Work decomposition: L lanes of a wavefront own one rating (G = 64/L ratings per
// wave step); lane j holds floats [4j,4j+4) of every 4L-wide chunk of the two
// factor rows, i.e. each row is read and written as coalesced 16-byte pieces.
// The fp32 dot is a per-lane fma chain + xor butterfly inside the L-lane group
// (DPP row operations, no LDS).  The bracket arithmetic follows the reference line
// by line (-ffp-contract=off: no fusing that the reference build does not do).
#include <algorithm>
#include <cstdlib>

#include "mfx_internal.h"

#include "sgd_common.h"

template <int L, int C, int ARITH, int POL>
__device__ __forceinline__ void sgd_visit(const Rows<POL>& Um, const Rows<POL>& Vm, int64_t pe, int64_t qe,
                                          float r, float lr, float uReg, float iReg) {
  float4v p[C], q[C];
#pragma unroll
  for (int c = 0; c < C; c++) {
    p[c] = Um.ld(pe + c * 4 * L);
    q[c] = Vm.ld(qe + c * 4 * L);
  }
  const float est = group_dot<L, C>(p, q);
  sgd_axpys<C, ARITH>(p, q, r, est, lr, uReg, iReg);
#pragma unroll
  for (int c = 0; c < C; c++) {
    Um.st(pe + c * 4 * L, p[c]);
    Vm.st(qe + c * 4 * L, q[c]);
  }
}

// One 64-rating chunk of the epoch list: coalesced SoA read by the whole wave, then the
// chunk is walked G ratings at a time (group g takes entries g, G+g, 2G+g, ...).
template <int L, int C, int ARITH, int POL>
__device__ __forceinline__ void sgd_chunk(const Rows<POL>& Um, const Rows<POL>& Vm,
                                          const int32_t* __restrict__ eu, const int32_t* __restrict__ ei,
                                          const float* __restrict__ er, int64_t idx0, int nvalid, int lane,
                                          float lr, float uReg, float iReg) {
  constexpr int G = 64 / L;
  constexpr int LD = 4 * L * C;
  const int g = lane / L, j = lane % L;
  const bool ok = lane < nvalid;
  const int mu = ok ? eu[idx0 + lane] : 0;
  const int mi = ok ? ei[idx0 + lane] : 0;
  const float mr = ok ? er[idx0 + lane] : 0.0f;
#pragma unroll 1
  for (int s = 0; s < L; s++) {
    const int e = s * G + g;
    const int u = __shfl(mu, e, 64);
    const int it = __shfl(mi, e, 64);
    const float r = __shfl(mr, e, 64);
    if (e < nvalid)
      sgd_visit<L, C, ARITH, POL>(Um, Vm, (int64_t)u * LD + 4 * j, (int64_t)it * LD + 4 * j, r, lr, uReg, iReg);
  }
}

// Hogwild: every wave streams 64-rating chunks of the epoch list, grid-stride.
template <int L, int C, int ARITH, int POL>
__global__ __launch_bounds__(256) void sgd_hogwild_kernel(const int32_t* __restrict__ eu,
                                                          const int32_t* __restrict__ ei,
                                                          const float* __restrict__ er, int64_t first,
                                                          int64_t count, float* U, float* V, uint32_t ubytes,
                                                          uint32_t vbytes, float lr, float uReg, float iReg) {
  const Rows<POL> Um(U, ubytes), Vm(V, vbytes);
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t base = wave * 64; base < count; base += nwaves * 64) {
    const int nvalid = (int)(count - base < 64 ? count - base : 64);
    sgd_chunk<L, C, ARITH, POL>(Um, Vm, eu, ei, er, first + base, nvalid, lane, lr, uReg, iReg);
  }
}

// Serial: ONE group of L lanes visits the list in order.  Element k of a row is
// always read and written by the same lane, so program order alone makes every
// visit see the previous visits' writes: this is the reference's sequential loop.
template <int L, int C, int ARITH>
__global__ __launch_bounds__(64) void sgd_serial_kernel(const int32_t* __restrict__ eu,
                                                        const int32_t* __restrict__ ei,
                                                        const float* __restrict__ er, int64_t first,
                                                        int64_t count, float* U, float* V, float lr,
                                                        float uReg, float iReg) {
  constexpr int LD = 4 * L * C;
  const int j = threadIdx.x;
  if (j >= L) return;
  const Rows<0> Um(U, 0), Vm(V, 0);
  for (int64_t t = 0; t < count; t++) {
    const int u = eu[first + t];
    const int it = ei[first + t];
    const float r = er[first + t];
    sgd_visit<L, C, ARITH, 0>(Um, Vm, (int64_t)u * LD + 4 * j, (int64_t)it * LD + 4 * j, r, lr, uReg, iReg);
  }
}

// User-ordered (trainUShuffle in parallel): one group per user row, the user row
// stays in registers for the whole row, items in CSR order; lock-free on V.
template <int L, int C, int ARITH>
__global__ __launch_bounds__(256) void sgd_users_kernel(const int32_t* __restrict__ ulist, int64_t nusers,
                                                        const int64_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ rowind,
                                                        const float* __restrict__ rowval, float* U, float* V,
                                                        uint32_t ubytes, uint32_t vbytes, float lr, float uReg,
                                                        float iReg) {
  constexpr int G = 64 / L;
  constexpr int LD = 4 * L * C;
  const int lane = threadIdx.x & 63;
  const int g = lane / L, j = lane % L;
  const int64_t grp = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * G + g;
  const int64_t ngrp = (((int64_t)gridDim.x * blockDim.x) >> 6) * G;
  const Rows<1> Um(U, ubytes), Vm(V, vbytes);   // item rows are shared by all XCDs: agent-coherent accesses
  for (int64_t t = grp; t < nusers; t += ngrp) {
    const int u = ulist[t];
    const int64_t b = rowptr[u], e = rowptr[u + 1];
    for (int64_t ii = b; ii < e; ii++)
      sgd_visit<L, C, ARITH, 1>(Um, Vm, (int64_t)u * LD + 4 * j, (int64_t)rowind[ii] * LD + 4 * j, rowval[ii], lr,
                                uReg, iReg);
  }
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
// experiment knob: number of 256-thread workgroups of the Hogwild kernels (in-flight ratings
// = blocks * 4 waves * 64/L); default fills the chip (256 CUs x 8 blocks).
static int sgd_blocks(const mfx_ctx* ctx, const mfx_sgd_opts* o) {
  static int b = -1;
  if (b < 0) {
    const char* e = getenv("MFX_SGD_BLOCKS");
    b = e ? atoi(e) : 0;
    if (b < 0) b = 0;
  }
  if (b > 0) return b;
  if (o->blocks > 0) return std::min(o->blocks, 8192);
  return std::max(8, std::min(2048, std::min(ctx->nU, ctx->nI) / 64));
}
static int sgd_policy() {
  static int pol = -1;
  if (pol < 0) {
    // default 1: the flat Hogwild kernel touches every row from all 8 XCDs, whose L2s are
    // not coherent with each other; plain write-back stores leave 8 diverging copies of
    // the hot rows (measured: much slower convergence), sc1 keeps one copy at the memory side
    const char* e = getenv("MFX_SGD_POLICY");
    pol = e ? atoi(e) : 1;
    if (pol < 0 || pol > 4) pol = 1;
  }
  return pol;
}
template <int L, int C, int ARITH>
static int launch_lca(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count, int64_t nusers) {
  ProfScope ps(ctx, MFX_K_SGD);
  if (o->mode == MFX_SGD_SERIAL) {
    hipLaunchKernelGGL((sgd_serial_kernel<L, C, ARITH>), dim3(1), dim3(64), 0, ctx->stream, ctx->eu, ctx->ei,
                       ctx->er, first, count, ctx->U, ctx->V, o->learnRate, o->uReg, o->iReg);
  } else if (o->mode == MFX_SGD_USERS) {
    constexpr int G = 64 / L;
    int64_t waves = (nusers + G - 1) / G;
    int blocks = (int)std::min<int64_t>((waves + 3) / 4, 2048);
    const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
    hipLaunchKernelGGL((sgd_users_kernel<L, C, ARITH>), dim3(blocks), dim3(256), 0, ctx->stream, ctx->ulist,
                       nusers, m.rowptr, m.rowind, m.rowval, ctx->U, ctx->V, (uint32_t)((uint64_t)ctx->nU * ctx->ld * 4),
                       (uint32_t)((uint64_t)ctx->nI * ctx->ld * 4), o->learnRate, o->uReg, o->iReg);
  } else {
    int64_t waves = (count + 63) / 64;
    int blocks = (int)std::min<int64_t>((waves + 3) / 4, sgd_blocks(ctx, o));
    const uint64_t ub = (uint64_t)ctx->nU * ctx->ld * 4, vb = (uint64_t)ctx->nI * ctx->ld * 4;
    int pol = sgd_policy();
    if (ub >= (1ull << 32) || vb >= (1ull << 32)) pol = 0;  // buffer descriptors address 4 GiB
#define MFX_LAUNCH_HOG(P)                                                                                     \
  hipLaunchKernelGGL((sgd_hogwild_kernel<L, C, ARITH, P>), dim3(blocks), dim3(256), 0, ctx->stream, ctx->eu, \
                     ctx->ei, ctx->er, first, count, ctx->U, ctx->V, (uint32_t)ub, (uint32_t)vb,             \
                     o->learnRate, o->uReg, o->iReg)
    switch (pol) {
      case 1: MFX_LAUNCH_HOG(1); break;
      case 2: MFX_LAUNCH_HOG(2); break;
      case 3: MFX_LAUNCH_HOG(3); break;
      case 4: MFX_LAUNCH_HOG(4); break;
      default: MFX_LAUNCH_HOG(0); break;
    }
#undef MFX_LAUNCH_HOG
  }
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

template <int L, int C>
static int launch_lc(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count, int64_t nusers) {
  switch (o->arith) {
    case MFX_ARITH_REF64: return launch_lca<L, C, MFX_ARITH_REF64>(ctx, o, first, count, nusers);
    case MFX_ARITH_REF64F: return launch_lca<L, C, MFX_ARITH_REF64F>(ctx, o, first, count, nusers);
    default: return launch_lca<L, C, MFX_ARITH_F32>(ctx, o, first, count, nusers);
  }
}

static int launch_any(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count, int64_t nusers) {
  const int L = ctx->L, C = ctx->C;
  if (L == 4) return launch_lc<4, 1>(ctx, o, first, count, nusers);
  if (L == 8) return launch_lc<8, 1>(ctx, o, first, count, nusers);
  switch (C) {
    case 1: return launch_lc<16, 1>(ctx, o, first, count, nusers);
    case 2: return launch_lc<16, 2>(ctx, o, first, count, nusers);
    case 3: return launch_lc<16, 3>(ctx, o, first, count, nusers);
    case 4: return launch_lc<16, 4>(ctx, o, first, count, nusers);
    case 5: return launch_lc<16, 5>(ctx, o, first, count, nusers);
    case 6: return launch_lc<16, 6>(ctx, o, first, count, nusers);
    case 7: return launch_lc<16, 7>(ctx, o, first, count, nusers);
    case 8: return launch_lc<16, 8>(ctx, o, first, count, nusers);
  }
  return mfx_fail(ctx, MFX_E_ARG, "sgd: unsupported rank shape L=%d C=%d", L, C);
}

int mfx_launch_sgd(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count) {
  if (ctx->bias_epoch) return mfx_launch_sgd_bias(ctx, o, first, count);                              // sgd_bias.hip
  if (o->mode == MFX_SGD_LEVELS) return mfx_launch_sgd_levels(ctx, o, first, count);                  // sgd_levels.hip
  if (ctx->dimreg) return mfx_launch_sgd_dimreg(ctx, o, first, count);   // trainSGDParSVD's regulariser (svd.hip)
  if (ctx->ifw) return mfx_launch_sgd_ifw(ctx, o, first, count);         // ModelInvPopMF's rating weights (sgd_ifw.hip)
  if (ctx->tmf_u) return mfx_launch_sgd_tmf(ctx, o, first, count);       // ModelDropoutSigmoid's truncated ranks (sgd_tmf.hip)
  return launch_any(ctx, o, first, count, 0);
}
int mfx_launch_sgd_users(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t nusers) {
  return launch_any(ctx, o, 0, 0, nusers);
}
