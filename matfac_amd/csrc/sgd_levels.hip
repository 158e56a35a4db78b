// sgd_levels.hip -- MFX_SGD_LEVELS: the reference's SEQUENTIAL SGD loops replayed bit for bit, in parallel.
//
// Replaces the loop of ModelMF::train (modelMF.cpp:83-105), trainUShuffle (:637-659) and -- with the visiting list
// the host class builds from the (user part, item part, matching) sequence -- trainSGDPar (:273-304), whenever the
// result has to BE the reference's: same list order, same arithmetic, np.array_equal with the oracle.
//
// Two visits that share neither the user nor the item touch disjoint rows and commute exactly.  The list order
// therefore only matters along the chains "ratings of one user" and "ratings of one item":
//     level(t) = 1 + max(level of the previous rating of user(t), level of the previous rating of item(t)).
// Ratings of one level are pairwise row-disjoint and every rating a visit depends on sits in a lower level, so
// running the levels in order, each level's ratings concurrently, gives exactly the sequential result.  The number
// of levels is about the length of the longest chain (the most popular item); a level holds nnz / levels ratings on
// average, thousands at the front and a handful in the tail.
//   phase A  levels [0, cut): one persistent launch, one workgroup per CU, a grid barrier (agent-scope release /
//            acquire around an atomic counter) between levels;
//   phase B  levels [cut, end) -- all of at most TAIL ratings -- on ONE workgroup with __syncthreads() between
//            levels: no grid barrier on the long thin tail, which is the dependent chain of the hottest rows.
// The level of every rating is computed on the host (one sequential pass over the list: the recurrence is the
// order itself); the lists are then gathered level-major on the device.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <vector>

#include "mfx_internal.h"

#include "sgd_common.h"

namespace {

constexpr int LV_WG = 256;        // threads per workgroup of both phases
constexpr int TAIL = 32;          // phase B takes over once every remaining level has at most this many ratings

struct LevelState {
  int32_t *lu = nullptr, *li = nullptr;   // (u, i, r) level-major
  float* lr = nullptr;
  int64_t* loff = nullptr;                // [nlevels + 1]
  uint32_t* lpos = nullptr;               // list position of every level-major entry (staging)
  unsigned* bar = nullptr;                // [2]: barrier counter, abort flag
  int64_t cap = 0, loff_cap = 0;
  int64_t nlevels = 0, cut = 0;
  double prep_ms = 0;                     // host time of the last level construction (download, levels, upload)
  std::vector<int32_t> hu, hi, lev, last_u, last_i;
  std::vector<uint32_t> hpos;
  std::vector<int64_t> hoff;
};
LevelState* lv(mfx_ctx* ctx) { return (LevelState*)ctx->levels; }

// POL 1: agent-scope (sc1) loads and write-through stores -- the rows never sit in a non-coherent cache, so the
// grid barrier needs no cache maintenance; POL 0: plain accesses (tables of 4 GiB and more, which a buffer
// descriptor cannot address), the barrier then writes back / invalidates the L2 (a full-cache walk per level).
template <int L, int C, int ARITH, int POL>
__device__ __forceinline__ void level_visit(const Rows<POL>& Um, const Rows<POL>& Vm, int u, int it, float r, int j, float lr,
                                            float uReg, float iReg) {
  constexpr int LD = 4 * L * C;
  const int64_t pe = (int64_t)u * LD + 4 * j, qe = (int64_t)it * LD + 4 * j;
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

template <int L, int C, int ARITH, int POL>
__global__ __launch_bounds__(LV_WG) void sgd_levels_grid_kernel(const int32_t* __restrict__ lu, const int32_t* __restrict__ li,
                                                                const float* __restrict__ lr_, const int64_t* __restrict__ loff,
                                                                int64_t lev0, int64_t lev1, float* U, float* V, uint32_t ubytes,
                                                                uint32_t vbytes, float lr, float uReg, float iReg, unsigned* bar) {
  constexpr int G = 64 / L;
  const Rows<POL> Um(U, ubytes), Vm(V, vbytes);
  const int lane = threadIdx.x & 63, g = lane / L, j = lane % L;
  const int64_t grp = ((int64_t)blockIdx.x * (LV_WG / 64) + (threadIdx.x >> 6)) * G + g;
  const int64_t ngrp = (int64_t)gridDim.x * (LV_WG / 64) * G;
  unsigned done = 0;
  for (int64_t l = lev0; l < lev1; l++) {
    const int64_t b = loff[l], e = loff[l + 1];
    for (int64_t x = b + grp; x < e; x += ngrp) level_visit<L, C, ARITH, POL>(Um, Vm, lu[x], li[x], lr_[x], j, lr, uReg, iReg);
    done += gridDim.x;
    if (l + 1 < lev1 && grid_barrier<POL>(bar, done)) return;
  }
}

template <int L, int C, int ARITH>
__global__ __launch_bounds__(LV_WG) void sgd_levels_tail_kernel(const int32_t* __restrict__ lu, const int32_t* __restrict__ li,
                                                                const float* __restrict__ lr_, const int64_t* __restrict__ loff,
                                                                int64_t lev0, int64_t lev1, float* U, float* V, float lr, float uReg,
                                                                float iReg) {
  constexpr int G = 64 / L;
  const Rows<0> Um(U, 0), Vm(V, 0);
  const int lane = threadIdx.x & 63, g = lane / L, j = lane % L;
  const int grp = (threadIdx.x >> 6) * G + g;
  constexpr int ngrp = (LV_WG / 64) * G;
  int64_t b = loff[lev0];
  for (int64_t l = lev0; l < lev1; l++) {
    const int64_t e = loff[l + 1];
    for (int64_t x = b + grp; x < e; x += ngrp) level_visit<L, C, ARITH, 0>(Um, Vm, lu[x], li[x], lr_[x], j, lr, uReg, iReg);
    b = e;
    // one CU, one L1, one L2: workgroup scope is enough between the levels of the tail
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's row stores have landed before anybody moves on
    __syncthreads();
  }
}

__global__ void levels_gather_kernel(const uint32_t* __restrict__ lpos, int64_t n, const int32_t* __restrict__ eu,
                                     const int32_t* __restrict__ ei, const float* __restrict__ er, int32_t* __restrict__ lu,
                                     int32_t* __restrict__ li, float* __restrict__ lr) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    const uint32_t s = lpos[t];
    lu[t] = eu[s]; li[t] = ei[s]; lr[t] = er[s];
  }
}

// levels of list[first, first+count): host pass, then level-major gather on the device
int build_levels(mfx_ctx* ctx, int64_t first, int64_t count) {
  LevelState* S = lv(ctx);
  if (!S) { S = new LevelState; ctx->levels = S; }
  const auto t0 = std::chrono::steady_clock::now();
  NEED(count < ((int64_t)1 << 32), MFX_E_ARG, "MFX_SGD_LEVELS: lists of 2^32 ratings or more are not supported");
  int rc;
  if (S->cap < count) {
    dev_free(S->lu); dev_free(S->li); dev_free(S->lr); dev_free(S->lpos);
    S->cap = 0;
    if ((rc = dev_alloc(ctx, &S->lu, (size_t)count)) || (rc = dev_alloc(ctx, &S->li, (size_t)count)) ||
        (rc = dev_alloc(ctx, &S->lr, (size_t)count)) || (rc = dev_alloc(ctx, &S->lpos, (size_t)count)))
      return rc;
    S->cap = count;
  }
  if (!S->bar && (rc = dev_alloc(ctx, &S->bar, (size_t)2))) return rc;
  S->hu.resize((size_t)count); S->hi.resize((size_t)count); S->lev.resize((size_t)count); S->hpos.resize((size_t)count);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(S->hu.data(), ctx->eu + first, sizeof(int32_t) * (size_t)count, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(S->hi.data(), ctx->ei + first, sizeof(int32_t) * (size_t)count, hipMemcpyDeviceToHost));
  S->last_u.assign((size_t)ctx->nU, 0);
  S->last_i.assign((size_t)ctx->nI, 0);
  int32_t* lu_ = S->last_u.data();
  int32_t* li_ = S->last_i.data();
  const int32_t *hu = S->hu.data(), *hi = S->hi.data();
  int32_t* lev = S->lev.data();
  int32_t nlev = 0;
  for (int64_t t = 0; t < count; t++) {
    const int32_t a = lu_[hu[t]], b = li_[hi[t]];
    const int32_t l = (a > b ? a : b) + 1;
    lev[t] = l; lu_[hu[t]] = l; li_[hi[t]] = l;
    nlev = l > nlev ? l : nlev;
  }
  S->hoff.assign((size_t)nlev + 2, 0);
  int64_t* off = S->hoff.data();
  for (int64_t t = 0; t < count; t++) off[lev[t] + 1]++;       // level l (1-based) -> slot l
  for (int32_t l = 1; l <= nlev + 1; l++) off[l] += off[l - 1];
  // off[l] = start of level l (1-based) ... shift to 0-based: level k (0-based) = [off[k+1], off[k+2])
  {
    std::vector<int64_t> cur(off + 1, off + nlev + 1);
    uint32_t* pos = S->hpos.data();
    for (int64_t t = 0; t < count; t++) pos[cur[(size_t)lev[t] - 1]++] = (uint32_t)t;   // stable: list order inside a level
  }
  if (S->loff_cap < (int64_t)nlev + 1) {
    dev_free(S->loff);
    S->loff_cap = 0;
    if ((rc = dev_alloc(ctx, &S->loff, (size_t)nlev + 1))) return rc;
    S->loff_cap = (int64_t)nlev + 1;
  }
  HIPCHK(hipMemcpyAsync(S->loff, off + 1, sizeof(int64_t) * ((size_t)nlev + 1), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(S->lpos, S->hpos.data(), sizeof(uint32_t) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
  const int blocks = (int)std::min<int64_t>((count + 255) / 256, 8192);
  hipLaunchKernelGGL(levels_gather_kernel, dim3(blocks), dim3(256), 0, ctx->stream, S->lpos, count, ctx->eu + first, ctx->ei + first,
                     ctx->er + first, S->lu, S->li, S->lr);
  HIPCHK(hipGetLastError());
  S->nlevels = nlev;
  // phase B starts behind the last level that is larger than TAIL
  int64_t cut = 0;
  for (int64_t k = 0; k < nlev; k++)
    if (off[k + 2] - off[k + 1] > TAIL) cut = k + 1;
  S->cut = cut;
  HIPCHK(hipStreamSynchronize(ctx->stream));    // the staging vectors are reused by the next call
  S->prep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (getenv("MFX_DEBUG"))
    fprintf(stderr, "[mfx] levels: %lld ratings in %lld levels (grid phase %lld levels, tail %lld), host preparation %.1f ms\n",
            (long long)count, (long long)nlev, (long long)cut, (long long)(nlev - cut), S->prep_ms);
  return MFX_OK;
}

template <int L, int C, int ARITH>
int launch_levels_lca(mfx_ctx* ctx, const mfx_sgd_opts* o) {
  LevelState* S = lv(ctx);
  ProfScope ps(ctx, MFX_K_SGD);
  if (S->cut > 0) {
    int dev = 0, cus = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int blocks = std::max(1, std::min(cus, 1024));      // one workgroup per CU: all resident, the barrier cannot starve
    HIPCHK(hipMemsetAsync(S->bar, 0, 2 * sizeof(unsigned), ctx->stream));
    const uint64_t ub = (uint64_t)ctx->nU * ctx->ld * 4, vb = (uint64_t)ctx->nI * ctx->ld * 4;
    if (ub < (1ull << 32) && vb < (1ull << 32) && !getenv("MFX_LEVELS_FENCE"))
      hipLaunchKernelGGL((sgd_levels_grid_kernel<L, C, ARITH, 1>), dim3(blocks), dim3(LV_WG), 0, ctx->stream, S->lu, S->li, S->lr,
                         S->loff, (int64_t)0, S->cut, ctx->U, ctx->V, (uint32_t)ub, (uint32_t)vb, o->learnRate, o->uReg, o->iReg, S->bar);
    else
      hipLaunchKernelGGL((sgd_levels_grid_kernel<L, C, ARITH, 0>), dim3(blocks), dim3(LV_WG), 0, ctx->stream, S->lu, S->li, S->lr,
                         S->loff, (int64_t)0, S->cut, ctx->U, ctx->V, 0u, 0u, o->learnRate, o->uReg, o->iReg, S->bar);
  }
  if (S->cut < S->nlevels)
    hipLaunchKernelGGL((sgd_levels_tail_kernel<L, C, ARITH>), dim3(1), dim3(LV_WG), 0, ctx->stream, S->lu, S->li, S->lr, S->loff, S->cut,
                       S->nlevels, ctx->U, ctx->V, o->learnRate, o->uReg, o->iReg);
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

template <int L, int C>
int launch_levels_lc(mfx_ctx* ctx, const mfx_sgd_opts* o) {
  switch (o->arith) {
    case MFX_ARITH_REF64: return launch_levels_lca<L, C, MFX_ARITH_REF64>(ctx, o);
    case MFX_ARITH_REF64F: return launch_levels_lca<L, C, MFX_ARITH_REF64F>(ctx, o);
    default: return launch_levels_lca<L, C, MFX_ARITH_F32>(ctx, o);
  }
}

}  // namespace

void mfx_levels_free_internal(mfx_ctx* ctx) {
  LevelState* S = lv(ctx);
  if (!S) return;
  dev_free(S->lu); dev_free(S->li); dev_free(S->lr); dev_free(S->loff); dev_free(S->lpos); dev_free(S->bar);
  delete S;
  ctx->levels = nullptr;
}

int mfx_launch_sgd_levels(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count) {
  // default: the barrier-free dataflow schedule (sgd_flow.hip); the level schedule below serves factor tables a buffer
  // descriptor cannot address (4 GiB and more) and MFX_EXACT_SCHED=levels -- the plain update only
  const char* sched = getenv("MFX_EXACT_SCHED");
  const bool variant = ctx->dimreg || ctx->ifw || ctx->tmf_u;
  ctx->last_exact_flow = mfx_flow_usable(ctx, count) && (variant || !(sched && sched[0] == 'l'));
  if (ctx->last_exact_flow) return mfx_launch_sgd_flow(ctx, o, first, count);
  NEED(!variant, MFX_E_ARG, "MFX_SGD_LEVELS: the SGD variants run on the dataflow schedule only (factor tables < 4 GiB)");
  int rc = build_levels(ctx, first, count);
  if (rc) return rc;
  const int L = ctx->L, C = ctx->C;
  if (L == 4) rc = launch_levels_lc<4, 1>(ctx, o);
  else if (L == 8) rc = launch_levels_lc<8, 1>(ctx, o);
  else switch (C) {
    case 1: rc = launch_levels_lc<16, 1>(ctx, o); break;
    case 2: rc = launch_levels_lc<16, 2>(ctx, o); break;
    case 3: rc = launch_levels_lc<16, 3>(ctx, o); break;
    case 4: rc = launch_levels_lc<16, 4>(ctx, o); break;
    case 5: rc = launch_levels_lc<16, 5>(ctx, o); break;
    case 6: rc = launch_levels_lc<16, 6>(ctx, o); break;
    case 7: rc = launch_levels_lc<16, 7>(ctx, o); break;
    case 8: rc = launch_levels_lc<16, 8>(ctx, o); break;
    default: return mfx_fail(ctx, MFX_E_ARG, "sgd levels: unsupported rank shape L=%d C=%d", L, C);
  }
  if (rc) return rc;
  // a barrier that gave up leaves its flag behind: report it instead of returning a half-run epoch
  LevelState* S = lv(ctx);
  if (S->cut > 0) {
    unsigned flag[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(flag, S->bar, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    NEED(flag[1] == 0, MFX_E_HIP, "MFX_SGD_LEVELS: the grid barrier timed out (device shared with another resident kernel?)");
  }
  return MFX_OK;
}

extern "C" int mfx_debug_levels_info(mfx_ctx* ctx, int64_t info[4], double* prep_ms) {
  if (!ctx) return MFX_E_ARG;
  NEED(info, MFX_E_ARG, "mfx_debug_levels_info: info NULL");
  if (ctx->last_exact_flow && mfx_flow_info(ctx, info, prep_ms)) return MFX_OK;
  LevelState* S = lv(ctx);
  NEED(S, MFX_E_STATE, "mfx_debug_levels_info: no MFX_SGD_LEVELS epoch has run");
  info[0] = 0; info[1] = S->nlevels; info[2] = S->cut; info[3] = TAIL;
  if (prep_ms) *prep_ms = S->prep_ms;
  return MFX_OK;
}
