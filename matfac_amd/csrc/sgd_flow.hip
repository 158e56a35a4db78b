// sgd_flow.hip -- the dataflow schedule of MFX_SGD_LEVELS: the reference's sequential SGD loops
// (modelMF.cpp:83-105, 637-659, 273-304) replayed bit for bit WITHOUT a global barrier.
//
// Same fact as sgd_levels.hip: two visits that share no row commute exactly, so the list order only matters along
// the chain of one user's ratings and the chain of one item's ratings.  Here the chains are enforced directly:
//   * the side with the longest chain (items, on rating data) is OWNED: every row of that side belongs to one lane
//     group, and a group visits the ratings of its rows in list order -- program order is the item chain;
//   * the other side carries a version counter per row: rating t, the k-th rating of its user in the list, waits
//     until ver[user] == k, and publishes ver[user] = k + 1 after its row stores have been acknowledged.
// The earliest unvisited rating of the list is always at the head of its group's queue with its version satisfied,
// so the schedule cannot stall as long as every group is resident (one launch of 4 workgroups per CU); the run time
// is the longest dependent chain times the latency of one visit, not levels x barrier.  A group polls the version of
// the rating at its queue head; all row and version accesses are agent-scope (sc1) loads and write-through stores, so
// no cache is ever stale and no cache maintenance is needed.  A watchdog (no progress anywhere in the wave for 2 s)
// raises an abort flag that every wave checks: the launch always drains.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "mfx_internal.h"

#include "sgd_common.h"
#include "sgd_variants.h"

namespace {

constexpr int FL_WG = 256;            // threads per workgroup
constexpr int FL_WG_PER_CU = 4;       // 16 waves per CU: resident for any rank shape (<= 128 VGPRs)

struct FlowState {
  int4* q = nullptr;                  // queue records {other-side row, owned row, rating bits, expected version}, group-major
  int64_t* qoff = nullptr;            // [groups + 1]
  uint32_t *lpos = nullptr, *vexp = nullptr;   // staging: list position and expected version of every queue slot
  unsigned* ver = nullptr;            // version per row of the other side
  unsigned* flag = nullptr;           // abort flag
  int64_t cap = 0, goff_cap = 0, ver_cap = 0;
  int64_t groups = 0, longest = 0;
  int own_user = 0;
  double prep_ms = 0;
  std::vector<int32_t> hu, hi, cnt, owner;
  std::vector<uint32_t> hpos, hver;
  std::vector<int64_t> hoff;
  // device-side construction (build_flow_device): sort buffers, per-row tables
  uint32_t *k0 = nullptr, *k1 = nullptr, *v0 = nullptr, *v1 = nullptr, *vert = nullptr;
  char* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  int64_t dcap = 0;
  int32_t *degU = nullptr, *degI = nullptr, *downer = nullptr;
  int64_t* dstart = nullptr;
  int64_t rows_cap = 0;
  std::vector<int32_t> hdeg;
  std::vector<int64_t> hstart;
};
FlowState* fl(mfx_ctx* ctx) { return (FlowState*)ctx->flow; }

__global__ void flow_gather_kernel(const uint32_t* __restrict__ lpos, const uint32_t* __restrict__ vexp, int64_t n,
                                   const int32_t* __restrict__ eu, const int32_t* __restrict__ ei, const float* __restrict__ er,
                                   int own_user, int4* __restrict__ q) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    const uint32_t s = lpos[t];
    const int u = eu[s], i = ei[s];
    q[t] = make_int4(own_user ? i : u, own_user ? u : i, __float_as_int(er[s]), (int)vexp[t]);
  }
}

template <int L, int C, int ARITH>
__global__ __launch_bounds__(FL_WG, FL_WG_PER_CU) void sgd_flow_kernel(const int4* __restrict__ q, const int64_t* __restrict__ qoff,
                                                         unsigned* ver, float* U, float* V, uint32_t ubytes, uint32_t vbytes,
                                                         int own_user, float lr, float uReg, float iReg, unsigned* flag) {
  constexpr int G = 64 / L;
  constexpr int LD = 4 * L * C;
  const Rows<1> Um(U, ubytes), Vm(V, vbytes);
  const int lane = threadIdx.x & 63, g = lane / L, j = lane % L;
  const int64_t grp = ((int64_t)blockIdx.x * (FL_WG / 64) + (threadIdx.x >> 6)) * G + g;
  int64_t pos = qoff[grp];
  const int64_t end = qoff[grp + 1];
  long long t_last = wall_clock64();
  int idle = 0;
  while (__builtin_amdgcn_ballot_w64(pos < end) != 0) {
    bool ready = false;
    int4 rec = make_int4(0, 0, 0, 0);
    if (pos < end) {
      rec = q[pos];
      ready = __hip_atomic_load(ver + rec.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)rec.w;
    }
    asm volatile("" ::: "memory");      // the row loads below stay behind the version check
    if (ready) {
      const int u = own_user ? rec.y : rec.x, it = own_user ? rec.x : rec.y;
      const int64_t pe = (int64_t)u * LD + 4 * j, qe = (int64_t)it * LD + 4 * j;
      float4v p[C], qv[C];
#pragma unroll
      for (int c = 0; c < C; c++) {
        p[c] = Um.ld(pe + c * 4 * L);
        qv[c] = Vm.ld(qe + c * 4 * L);
      }
      const float est = group_dot<L, C>(p, qv);
      sgd_axpys<C, ARITH>(p, qv, __int_as_float(rec.z), est, lr, uReg, iReg);
#pragma unroll
      for (int c = 0; c < C; c++) {
        Um.st(pe + c * 4 * L, p[c]);
        Vm.st(qe + c * 4 * L, qv[c]);
      }
      // the write-through row stores must be ACKNOWLEDGED before the version moves.  (A workgroup-scope release fence emits
      // nothing here -- the version store overtook the row stores on another channel and a C2 epoch came out wrong.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (j == 0) __hip_atomic_store(ver + rec.x, (unsigned)rec.w + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... then publish
      pos++;
    }
    if (__builtin_amdgcn_ballot_w64(ready) != 0) {
      t_last = wall_clock64();
      idle = 0;
    } else {
      __builtin_amdgcn_s_sleep(2);
      if ((++idle & 63) == 0) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        if (wall_clock64() - t_last > 200000000LL) {              // 100 MHz constant clock: 2 s without progress
          __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
      }
    }
  }
}

// The three SGD variants on the same schedule (VAR 1: ModelInvPopMF's rating weight, 2: ModelDropoutSigmoid's / ModelPoissonDropout's
// truncated rank, 3: trainSGDParSVD's per-dimension regulariser): the visit is the variant's own (sgd_variants.h), the weight / rank of a
// rating is derived from the user's and the item's attribute pair as in the variants' serial kernels.
template <int L, int C, int VAR>
__global__ __launch_bounds__(FL_WG, FL_WG_PER_CU) void sgd_flow_var_kernel(const int4* __restrict__ q, const int64_t* __restrict__ qoff,
                                                                           unsigned* ver, float* U, float* V, uint32_t ubytes, uint32_t vbytes,
                                                                           int own_user, float lr, float uReg, float iReg, unsigned* flag,
                                                                           const float2* __restrict__ ua, const float2* __restrict__ ia, float rho,
                                                                           const int2* __restrict__ tu, const int2* __restrict__ ti,
                                                                           const int32_t* __restrict__ du, const int32_t* __restrict__ di,
                                                                           const double* __restrict__ dexp, uint32_t seed, uint32_t epoch, int K,
                                                                           const float* __restrict__ regk) {
  constexpr int G = 64 / L;
  constexpr int LD = 4 * L * C;
  const Rows<1> Um(U, ubytes), Vm(V, vbytes);
  const int lane = threadIdx.x & 63, g = lane / L, j = lane % L;
  const int64_t grp = ((int64_t)blockIdx.x * (FL_WG / 64) + (threadIdx.x >> 6)) * G + g;
  int64_t pos = qoff[grp];
  const int64_t end = qoff[grp + 1];
  float4v rk[C];
  if (VAR == 3) {
#pragma unroll
    for (int c = 0; c < C; c++) rk[c] = *(const float4v*)(regk + c * 4 * L + 4 * j);
  }
  long long t_last = wall_clock64();
  int idle = 0;
  while (__builtin_amdgcn_ballot_w64(pos < end) != 0) {
    bool ready = false;
    int4 rec = make_int4(0, 0, 0, 0);
    if (pos < end) {
      rec = q[pos];
      ready = __hip_atomic_load(ver + rec.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)rec.w;
    }
    asm volatile("" ::: "memory");
    if (ready) {
      const int u = own_user ? rec.y : rec.x, it = own_user ? rec.x : rec.y;
      const int64_t pe = (int64_t)u * LD + 4 * j, qe = (int64_t)it * LD + 4 * j;
      const float r = __int_as_float(rec.z);
      if (VAR == 1) {
        visit_ifw<L, C, 1>(Um, Vm, pe, qe, r, mfx_ifw_weight(ua[u], ia[it], rho), lr, uReg, iReg);
      } else if (VAR == 2) {
        const int2 a = tu[u], b = ti[it];
        int rank = mfx_tmf_rank(a, b);
        if (du) {
          const int lam = __int_as_float(a.x) < __int_as_float(b.x) ? du[u] : di[it];
          rank = mfx_poisson_rank(lam, dexp[lam], mfx_draw_hash(seed, epoch, (uint32_t)u, (uint32_t)it), K);
        }
        visit_tmf<L, C, 1>(Um, Vm, pe, qe, r, rank, j, lr, uReg, iReg);
      } else {
        visit_dimreg<L, C, 1>(Um, Vm, pe, qe, r, lr, rk);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // row stores acknowledged, then the version moves
      if (j == 0) __hip_atomic_store(ver + rec.x, (unsigned)rec.w + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pos++;
    }
    if (__builtin_amdgcn_ballot_w64(ready) != 0) {
      t_last = wall_clock64();
      idle = 0;
    } else {
      __builtin_amdgcn_s_sleep(2);
      if ((++idle & 63) == 0) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        if (wall_clock64() - t_last > 200000000LL) {
          __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
      }
    }
  }
}

// ModelMFBias (modelMFBias.cpp:178-197) on the same schedule with ONE lane per rating: the owned side's bias stays with
// its lane's queue, the other side's bias carries the version counter.  The visit is sgd_bias.hip's, statement by statement.
__global__ __launch_bounds__(FL_WG, FL_WG_PER_CU) void bias_flow_kernel(const int4* __restrict__ q, const int64_t* __restrict__ qoff,
                                                                        unsigned* ver, float* ub, float* ib, int own_user, float lr,
                                                                        float uReg, float iReg, unsigned* flag) {
  const int64_t grp = (int64_t)blockIdx.x * FL_WG + threadIdx.x;
  int64_t pos = qoff[grp];
  const int64_t end = qoff[grp + 1];
  long long t_last = wall_clock64();
  int idle = 0;
  while (__builtin_amdgcn_ballot_w64(pos < end) != 0) {
    bool ready = false;
    int4 rec = make_int4(0, 0, 0, 0);
    if (pos < end) {
      rec = q[pos];
      ready = __hip_atomic_load(ver + rec.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)rec.w;
    }
    asm volatile("" ::: "memory");
    if (ready) {
      const int u = own_user ? rec.y : rec.x, it = own_user ? rec.x : rec.y;
      float bu = __hip_atomic_load(ub + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float bi = __hip_atomic_load(ib + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      {
        const float est = bu + bi;
        const double diff = (double)__int_as_float(rec.z) - (double)est;
        bu = (float)((double)bu - (double)lr * (-2.0 * diff + (2.0 * (double)uReg) * (double)bu));
        bi = (float)((double)bi - (double)lr * (-2.0 * diff + (2.0 * (double)iReg) * (double)bi));
      }
      __hip_atomic_store(ub + u, bu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ib + it, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the two stores are acknowledged, then the version moves
      __hip_atomic_store(ver + rec.x, (unsigned)rec.w + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pos++;
    }
    if (__builtin_amdgcn_ballot_w64(ready) != 0) {
      t_last = wall_clock64();
      idle = 0;
    } else {
      __builtin_amdgcn_s_sleep(2);
      if ((++idle & 63) == 0) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        if (wall_clock64() - t_last > 200000000LL) {
          __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
      }
    }
  }
}

int build_flow(mfx_ctx* ctx, int64_t first, int64_t count, int64_t groups) {
  FlowState* S = fl(ctx);
  const auto t0 = std::chrono::steady_clock::now();
  int rc;
  if (S->cap < count) {
    dev_free(S->q); dev_free(S->lpos); dev_free(S->vexp);
    S->cap = 0;
    if ((rc = dev_alloc(ctx, &S->q, (size_t)count)) || (rc = dev_alloc(ctx, &S->lpos, (size_t)count)) ||
        (rc = dev_alloc(ctx, &S->vexp, (size_t)count)))
      return rc;
    S->cap = count;
  }
  if (S->goff_cap < groups + 1) {
    dev_free(S->qoff);
    S->goff_cap = 0;
    if ((rc = dev_alloc(ctx, &S->qoff, (size_t)groups + 1))) return rc;
    S->goff_cap = groups + 1;
  }
  if (!S->flag && (rc = dev_alloc(ctx, &S->flag, (size_t)1))) return rc;
  S->hu.resize((size_t)count); S->hi.resize((size_t)count); S->hpos.resize((size_t)count); S->hver.resize((size_t)count);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(S->hu.data(), ctx->eu + first, sizeof(int32_t) * (size_t)count, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(S->hi.data(), ctx->ei + first, sizeof(int32_t) * (size_t)count, hipMemcpyDeviceToHost));
  // chain lengths of both sides in this list; the side with the longest chain is owned
  std::vector<int32_t> degU((size_t)ctx->nU, 0), degI((size_t)ctx->nI, 0);
  for (int64_t t = 0; t < count; t++) { degU[(size_t)S->hu[(size_t)t]]++; degI[(size_t)S->hi[(size_t)t]]++; }
  const int32_t maxU = *std::max_element(degU.begin(), degU.end()), maxI = *std::max_element(degI.begin(), degI.end());
  const int own_user = maxU > maxI ? 1 : 0;
  const std::vector<int32_t>& degOwn = own_user ? degU : degI;
  const int32_t* hown = own_user ? S->hu.data() : S->hi.data();
  const int32_t* hoth = own_user ? S->hi.data() : S->hu.data();
  const int64_t nOwn = own_user ? ctx->nU : ctx->nI, nOth = own_user ? ctx->nI : ctx->nU;
  // owned rows -> groups: longest chain first onto the least loaded group
  S->owner.assign((size_t)nOwn, 0);
  {
    std::vector<int32_t> rows;
    for (int64_t r = 0; r < nOwn; r++)
      if (degOwn[(size_t)r] > 0) rows.push_back((int32_t)r);
    std::sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) { return degOwn[a] != degOwn[b] ? degOwn[a] > degOwn[b] : a < b; });
    typedef std::pair<int64_t, int32_t> Load;   // (ratings so far, group)
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int64_t gI = 0; gI < groups; gI++) heap.push(Load(0, (int32_t)gI));
    for (int32_t r : rows) {
      Load l = heap.top();
      heap.pop();
      S->owner[(size_t)r] = l.second;
      l.first += degOwn[(size_t)r];
      heap.push(l);
    }
  }
  S->hoff.assign((size_t)groups + 1, 0);
  int64_t* off = S->hoff.data();
  for (int64_t t = 0; t < count; t++) off[S->owner[(size_t)hown[t]] + 1]++;
  int64_t longest = 0;
  for (int64_t gI = 0; gI < groups; gI++) { longest = std::max(longest, off[gI + 1]); off[gI + 1] += off[gI]; }
  {
    std::vector<int64_t> cur(off, off + groups);
    S->cnt.assign((size_t)nOth, 0);
    int32_t* cnt = S->cnt.data();
    uint32_t *hp = S->hpos.data(), *hv = S->hver.data();
    for (int64_t t = 0; t < count; t++) {       // list order inside a queue; version = rank of t in its other-side chain
      const int64_t slot = cur[(size_t)S->owner[(size_t)hown[t]]]++;
      hp[slot] = (uint32_t)t;
      hv[slot] = (uint32_t)cnt[hoth[t]]++;
    }
  }
  if (S->ver_cap < nOth) {
    dev_free(S->ver);
    S->ver_cap = 0;
    if ((rc = dev_alloc(ctx, &S->ver, (size_t)nOth))) return rc;
    S->ver_cap = nOth;
  }
  HIPCHK(hipMemsetAsync(S->ver, 0, sizeof(unsigned) * (size_t)nOth, ctx->stream));
  HIPCHK(hipMemsetAsync(S->flag, 0, sizeof(unsigned), ctx->stream));
  HIPCHK(hipMemcpyAsync(S->qoff, off, sizeof(int64_t) * ((size_t)groups + 1), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(S->lpos, S->hpos.data(), sizeof(uint32_t) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(S->vexp, S->hver.data(), sizeof(uint32_t) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
  const int blocks = (int)std::min<int64_t>((count + 255) / 256, 8192);
  hipLaunchKernelGGL(flow_gather_kernel, dim3(blocks), dim3(256), 0, ctx->stream, S->lpos, S->vexp, count, ctx->eu + first,
                     ctx->ei + first, ctx->er + first, own_user, S->q);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));    // the staging vectors are reused by the next call
  S->groups = groups; S->longest = longest; S->own_user = own_user;
  S->prep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (getenv("MFX_DEBUG"))
    fprintf(stderr, "[mfx] dataflow replay: %lld ratings on %lld groups (%s rows owned), longest queue %lld, longest chains %d users / %d items, "
            "host preparation %.1f ms\n", (long long)count, (long long)groups, own_user ? "user" : "item", (long long)longest, maxU, maxI,
            S->prep_ms);
  return MFX_OK;
}

// ---- the same construction on the device (default): the host keeps only the assignment of owned rows to queues -----
__global__ void flow_degrees_kernel(const int32_t* __restrict__ eu, const int32_t* __restrict__ ei, int64_t n, int32_t* __restrict__ degU,
                                    int32_t* __restrict__ degI) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    atomicAdd(degU + eu[t], 1);
    atomicAdd(degI + ei[t], 1);
  }
}
// key = row of `side` for every list position (or the queue of its owned row), value = the position
__global__ void flow_keys_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ owner, int64_t n, uint32_t* __restrict__ key,
                                 uint32_t* __restrict__ val) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    key[t] = owner ? (uint32_t)owner[rows[t]] : (uint32_t)rows[t];
    val[t] = (uint32_t)t;
  }
}
// sorted by other-side row (stable: list order inside a row): rank of position v1[k] in its row's chain = k - start[row]
__global__ void flow_rank_kernel(const uint32_t* __restrict__ k1, const uint32_t* __restrict__ v1, const int64_t* __restrict__ start, int64_t n,
                                 uint32_t* __restrict__ vert) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) vert[v1[k]] = (uint32_t)(k - start[k1[k]]);
}
__global__ void flow_gather2_kernel(const uint32_t* __restrict__ lpos, const uint32_t* __restrict__ vert, int64_t n, const int32_t* __restrict__ eu,
                                    const int32_t* __restrict__ ei, const float* __restrict__ er, int own_user, int4* __restrict__ q) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    const uint32_t s = lpos[t];
    const int u = eu[s], i = ei[s];
    q[t] = make_int4(own_user ? i : u, own_user ? u : i, __float_as_int(er[s]), (int)vert[s]);
  }
}
// qoff[g] = number of sorted queue keys < g
__global__ void flow_bounds_kernel(const uint32_t* __restrict__ keys, int64_t n, int64_t ng1, int64_t* __restrict__ qoff) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ng1; g += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)keys[mid] < g) lo = mid + 1; else hi = mid;
    }
    qoff[g] = lo;
  }
}
static int bits_for(uint64_t n) {
  int b = 1;
  while (b < 32 && ((uint64_t)1 << b) < n) b++;
  return b;
}

int build_flow_device(mfx_ctx* ctx, int64_t first, int64_t count, int64_t groups) {
  FlowState* S = fl(ctx);
  const auto t0 = std::chrono::steady_clock::now();
  int rc;
  const int64_t nU = ctx->nU, nI = ctx->nI;
  if (S->cap < count) {
    dev_free(S->q); dev_free(S->lpos); dev_free(S->vexp);
    S->cap = 0;
    if ((rc = dev_alloc(ctx, &S->q, (size_t)count)) || (rc = dev_alloc(ctx, &S->lpos, (size_t)count)) ||
        (rc = dev_alloc(ctx, &S->vexp, (size_t)count)))
      return rc;
    S->cap = count;
  }
  if (S->dcap < count) {
    dev_free(S->k0); dev_free(S->k1); dev_free(S->v0); dev_free(S->v1); dev_free(S->vert); dev_free(S->sort_tmp);
    S->dcap = 0;
    if ((rc = dev_alloc(ctx, &S->k0, (size_t)count)) || (rc = dev_alloc(ctx, &S->k1, (size_t)count)) || (rc = dev_alloc(ctx, &S->v0, (size_t)count)) ||
        (rc = dev_alloc(ctx, &S->v1, (size_t)count)) || (rc = dev_alloc(ctx, &S->vert, (size_t)count)))
      return rc;
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, S->k0, S->k1, S->v0, S->v1, (size_t)count, 0, 32, ctx->stream));
    if ((rc = dev_alloc(ctx, &S->sort_tmp, bytes))) return rc;
    S->sort_tmp_bytes = bytes;
    S->dcap = count;
  }
  if (S->rows_cap < nU + nI) {
    dev_free(S->degU); dev_free(S->downer); dev_free(S->dstart);
    S->rows_cap = 0;
    if ((rc = dev_alloc(ctx, &S->degU, (size_t)(nU + nI))) || (rc = dev_alloc(ctx, &S->downer, (size_t)std::max(nU, nI))) ||
        (rc = dev_alloc(ctx, &S->dstart, (size_t)std::max(nU, nI) + 1)))
      return rc;
    S->rows_cap = nU + nI;
  }
  S->degI = S->degU + nU;
  if (S->goff_cap < groups + 1) {
    dev_free(S->qoff);
    S->goff_cap = 0;
    if ((rc = dev_alloc(ctx, &S->qoff, (size_t)groups + 1))) return rc;
    S->goff_cap = groups + 1;
  }
  if (!S->flag && (rc = dev_alloc(ctx, &S->flag, (size_t)1))) return rc;
  const int32_t *eu = ctx->eu + first, *ei = ctx->ei + first;
  const float* er = ctx->er + first;
  const int grid = (int)std::min<int64_t>((count + 255) / 256, 8192);
  // chain lengths of both sides; the side with the longest chain is owned
  HIPCHK(hipMemsetAsync(S->degU, 0, sizeof(int32_t) * (size_t)(nU + nI), ctx->stream));
  hipLaunchKernelGGL(flow_degrees_kernel, dim3(grid), dim3(256), 0, ctx->stream, eu, ei, count, S->degU, S->degI);
  S->hdeg.resize((size_t)(nU + nI));
  HIPCHK(hipMemcpyAsync(S->hdeg.data(), S->degU, sizeof(int32_t) * (size_t)(nU + nI), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const int32_t *hdU = S->hdeg.data(), *hdI = S->hdeg.data() + nU;
  const int32_t maxU = *std::max_element(hdU, hdU + nU), maxI = *std::max_element(hdI, hdI + nI);
  const int own_user = maxU > maxI ? 1 : 0;
  const int32_t* degOwn = own_user ? hdU : hdI;
  const int32_t* degOth = own_user ? hdI : hdU;
  const int64_t nOwn = own_user ? nU : nI, nOth = own_user ? nI : nU;
  // owned rows -> queues: longest chain first onto the least loaded queue (host: a few 10^4..10^6 rows)
  S->owner.assign((size_t)nOwn, 0);
  {
    std::vector<int32_t> rows;
    for (int64_t r = 0; r < nOwn; r++)
      if (degOwn[r] > 0) rows.push_back((int32_t)r);
    std::sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) { return degOwn[a] != degOwn[b] ? degOwn[a] > degOwn[b] : a < b; });
    typedef std::pair<int64_t, int32_t> Load;
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int64_t gI = 0; gI < groups; gI++) heap.push(Load(0, (int32_t)gI));
    for (int32_t r : rows) {
      Load l = heap.top();
      heap.pop();
      S->owner[(size_t)r] = l.second;
      l.first += degOwn[r];
      heap.push(l);
    }
  }
  S->hstart.resize((size_t)nOth + 1);
  S->hstart[0] = 0;
  for (int64_t r = 0; r < nOth; r++) S->hstart[(size_t)r + 1] = S->hstart[(size_t)r] + degOth[r];
  HIPCHK(hipMemcpyAsync(S->downer, S->owner.data(), sizeof(int32_t) * (size_t)nOwn, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(S->dstart, S->hstart.data(), sizeof(int64_t) * ((size_t)nOth + 1), hipMemcpyHostToDevice, ctx->stream));
  if (S->ver_cap < nOth) {
    dev_free(S->ver);
    S->ver_cap = 0;
    if ((rc = dev_alloc(ctx, &S->ver, (size_t)nOth))) return rc;
    S->ver_cap = nOth;
  }
  HIPCHK(hipMemsetAsync(S->ver, 0, sizeof(unsigned) * (size_t)nOth, ctx->stream));
  HIPCHK(hipMemsetAsync(S->flag, 0, sizeof(unsigned), ctx->stream));
  const int32_t* own_rows = own_user ? eu : ei;
  const int32_t* oth_rows = own_user ? ei : eu;
  // expected version of every list position: its rank in the other side's chain (stable sort by that row)
  hipLaunchKernelGGL(flow_keys_kernel, dim3(grid), dim3(256), 0, ctx->stream, oth_rows, (const int32_t*)nullptr, count, S->k0, S->v0);
  size_t bytes = S->sort_tmp_bytes;
  HIPCHK(rocprim::radix_sort_pairs(S->sort_tmp, bytes, S->k0, S->k1, S->v0, S->v1, (size_t)count, 0, bits_for((uint64_t)nOth), ctx->stream));
  hipLaunchKernelGGL(flow_rank_kernel, dim3(grid), dim3(256), 0, ctx->stream, S->k1, S->v1, S->dstart, count, S->vert);
  // queue order: stable sort of the list positions by the queue of their owned row
  hipLaunchKernelGGL(flow_keys_kernel, dim3(grid), dim3(256), 0, ctx->stream, own_rows, (const int32_t*)S->downer, count, S->k0, S->v0);
  bytes = S->sort_tmp_bytes;
  HIPCHK(rocprim::radix_sort_pairs(S->sort_tmp, bytes, S->k0, S->k1, S->v0, S->lpos, (size_t)count, 0, bits_for((uint64_t)groups), ctx->stream));
  hipLaunchKernelGGL(flow_bounds_kernel, dim3((unsigned)std::min<int64_t>((groups + 256) / 256, 4096)), dim3(256), 0, ctx->stream, S->k1, count,
                     groups + 1, S->qoff);
  hipLaunchKernelGGL(flow_gather2_kernel, dim3(grid), dim3(256), 0, ctx->stream, S->lpos, S->vert, count, eu, ei, er, own_user, S->q);
  HIPCHK(hipGetLastError());
  S->hoff.resize((size_t)groups + 1);
  HIPCHK(hipMemcpyAsync(S->hoff.data(), S->qoff, sizeof(int64_t) * ((size_t)groups + 1), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int64_t longest = 0;
  for (int64_t gI = 0; gI < groups; gI++) longest = std::max(longest, S->hoff[(size_t)gI + 1] - S->hoff[(size_t)gI]);
  S->groups = groups; S->longest = longest; S->own_user = own_user;
  S->prep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (getenv("MFX_DEBUG"))
    fprintf(stderr, "[mfx] dataflow replay (device construction): %lld ratings on %lld queues (%s rows owned), longest queue %lld, longest chains "
            "%d users / %d items, preparation %.1f ms\n", (long long)count, (long long)groups, own_user ? "user" : "item", (long long)longest, maxU,
            maxI, S->prep_ms);
  return MFX_OK;
}

template <int L, int C>
int launch_flow_lc(mfx_ctx* ctx, const mfx_sgd_opts* o, int blocks) {
  FlowState* S = fl(ctx);
  const uint32_t ub = (uint32_t)((uint64_t)ctx->nU * ctx->ld * 4), vb = (uint32_t)((uint64_t)ctx->nI * ctx->ld * 4);
  ProfScope ps(ctx, MFX_K_SGD);
  const int var = ctx->ifw ? 1 : (ctx->tmf_u ? 2 : (ctx->dimreg ? 3 : 0));
  if (var != 0) {
    if constexpr (C <= 4) {      // K <= 256, as the variants' own kernels
      const float2 *ua = nullptr, *ia = nullptr;
      float rho = 0.0f;
      mfx_ifw_tables(ctx, &ua, &ia, &rho);
#define MFX_FLOWV(VV)                                                                                                          \
  hipLaunchKernelGGL((sgd_flow_var_kernel<L, C, VV>), dim3(blocks), dim3(FL_WG), 0, ctx->stream, (const int4*)S->q, S->qoff, S->ver, \
                     ctx->U, ctx->V, ub, vb, S->own_user, o->learnRate, o->uReg, o->iReg, S->flag, ua, ia, rho, ctx->tmf_u, ctx->tmf_i,  \
                     ctx->tmfd_u, ctx->tmfd_i, ctx->tmfd_exp, ctx->tmfd_seed, (uint32_t)o->epoch, ctx->K, ctx->dimreg)
      if (var == 1) MFX_FLOWV(1); else if (var == 2) MFX_FLOWV(2); else MFX_FLOWV(3);
#undef MFX_FLOWV
      HIPCHK(hipGetLastError());
      return MFX_OK;
    } else {
      return mfx_fail(ctx, MFX_E_ARG, "MFX_SGD_LEVELS: the SGD variants are built for K <= 256");
    }
  }
#define MFX_FLOW(A)                                                                                                        \
  hipLaunchKernelGGL((sgd_flow_kernel<L, C, A>), dim3(blocks), dim3(FL_WG), 0, ctx->stream, (const int4*)S->q, S->qoff, S->ver, \
                     ctx->U, ctx->V, ub, vb, S->own_user, o->learnRate, o->uReg, o->iReg, S->flag)
  switch (o->arith) {
    case MFX_ARITH_REF64: MFX_FLOW(MFX_ARITH_REF64); break;
    case MFX_ARITH_REF64F: MFX_FLOW(MFX_ARITH_REF64F); break;
    default: MFX_FLOW(MFX_ARITH_F32); break;
  }
#undef MFX_FLOW
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

}  // namespace

void mfx_flow_free_internal(mfx_ctx* ctx) {
  FlowState* S = fl(ctx);
  if (!S) return;
  dev_free(S->q); dev_free(S->qoff); dev_free(S->lpos); dev_free(S->vexp); dev_free(S->ver); dev_free(S->flag);
  dev_free(S->k0); dev_free(S->k1); dev_free(S->v0); dev_free(S->v1); dev_free(S->vert); dev_free(S->sort_tmp);
  dev_free(S->degU); dev_free(S->downer); dev_free(S->dstart);
  delete S;
  ctx->flow = nullptr;
}

bool mfx_flow_usable(const mfx_ctx* ctx, int64_t count) {
  return (uint64_t)ctx->nU * ctx->ld * 4 < (1ull << 32) && (uint64_t)ctx->nI * ctx->ld * 4 < (1ull << 32) &&
         count < ((int64_t)1 << 32);
}

int mfx_launch_sgd_flow(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count) {
  if (!fl(ctx)) ctx->flow = new FlowState;
  int dev = 0, cus = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int L = ctx->L, C = ctx->C;
  // every workgroup resident: the queues cannot starve each other; no more queues than there are rows to own
  const int64_t per_block = (int64_t)(FL_WG / 64) * (64 / L);
  const int blocks = (int)std::min<int64_t>((int64_t)std::max(1, cus) * FL_WG_PER_CU, (std::max(ctx->nU, ctx->nI) + per_block - 1) / per_block);
  const int64_t groups = (int64_t)blocks * per_block;
  // queues and versions are built on the device; MFX_FLOW_HOST=1 keeps the host statement of the same lists (the cross-check)
  int rc = getenv("MFX_FLOW_HOST") ? build_flow(ctx, first, count, groups) : build_flow_device(ctx, first, count, groups);
  if (rc) return rc;
  if (L == 4) rc = launch_flow_lc<4, 1>(ctx, o, blocks);
  else if (L == 8) rc = launch_flow_lc<8, 1>(ctx, o, blocks);
  else switch (C) {
    case 1: rc = launch_flow_lc<16, 1>(ctx, o, blocks); break;
    case 2: rc = launch_flow_lc<16, 2>(ctx, o, blocks); break;
    case 3: rc = launch_flow_lc<16, 3>(ctx, o, blocks); break;
    case 4: rc = launch_flow_lc<16, 4>(ctx, o, blocks); break;
    case 5: rc = launch_flow_lc<16, 5>(ctx, o, blocks); break;
    case 6: rc = launch_flow_lc<16, 6>(ctx, o, blocks); break;
    case 7: rc = launch_flow_lc<16, 7>(ctx, o, blocks); break;
    case 8: rc = launch_flow_lc<16, 8>(ctx, o, blocks); break;
    default: return mfx_fail(ctx, MFX_E_ARG, "sgd dataflow: unsupported rank shape L=%d C=%d", L, C);
  }
  if (rc) return rc;
  unsigned flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, fl(ctx)->flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  NEED(flag == 0, MFX_E_HIP, "MFX_SGD_LEVELS (dataflow): no progress for 2 s, launch abandoned (device shared with another resident kernel?)");
  return MFX_OK;
}

int mfx_launch_bias_flow(mfx_ctx* ctx, const mfx_sgd_opts* o, int64_t first, int64_t count) {
  if (!fl(ctx)) ctx->flow = new FlowState;
  int dev = 0, cus = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int blocks = (int)std::min<int64_t>((int64_t)std::max(1, cus) * FL_WG_PER_CU, (std::max(ctx->nU, ctx->nI) + FL_WG - 1) / FL_WG);
  int rc = getenv("MFX_FLOW_HOST") ? build_flow(ctx, first, count, (int64_t)blocks * FL_WG)     // one lane = one queue
                                   : build_flow_device(ctx, first, count, (int64_t)blocks * FL_WG);
  if (rc) return rc;
  FlowState* S = fl(ctx);
  {
    ProfScope ps(ctx, MFX_K_SGD);
    hipLaunchKernelGGL(bias_flow_kernel, dim3(blocks), dim3(FL_WG), 0, ctx->stream, (const int4*)S->q, S->qoff, S->ver, ctx->ub, ctx->ib,
                       S->own_user, o->learnRate, o->uReg, o->iReg, S->flag);
    HIPCHK(hipGetLastError());
  }
  unsigned flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, S->flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  NEED(flag == 0, MFX_E_HIP, "mfx_bias_epoch (dataflow): no progress for 2 s, launch abandoned");
  ctx->last_exact_flow = true;
  return MFX_OK;
}

// info = {1, longest queue, groups, owned side (0 item rows, 1 user rows)}
bool mfx_flow_info(mfx_ctx* ctx, int64_t info[4], double* prep_ms) {
  FlowState* S = fl(ctx);
  if (!S) return false;
  info[0] = 1; info[1] = S->longest; info[2] = S->groups; info[3] = S->own_user;
  if (prep_ms) *prep_ms = S->prep_ms;
  return true;
}

// test hook: the queues of the last dataflow epoch (records {other-side row, owned row, rating bits, expected version})
extern "C" int mfx_debug_flow_queues(mfx_ctx* ctx, int32_t* records, int64_t cap, int64_t* qoff, int64_t* n_records, int64_t* n_groups) {
  if (!ctx) return MFX_E_ARG;
  FlowState* S = fl(ctx);
  NEED(S && S->q && n_records && n_groups, MFX_E_STATE, "mfx_debug_flow_queues: no dataflow epoch has run");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::vector<int64_t> off((size_t)S->groups + 1);
  HIPCHK(hipMemcpy(off.data(), S->qoff, sizeof(int64_t) * off.size(), hipMemcpyDeviceToHost));
  *n_records = off.back();
  *n_groups = S->groups;
  if (!records) return MFX_OK;
  NEED(cap >= off.back(), MFX_E_ARG, "mfx_debug_flow_queues: cap too small");
  HIPCHK(hipMemcpy(records, S->q, sizeof(int4) * (size_t)off.back(), hipMemcpyDeviceToHost));
  if (qoff) memcpy(qoff, off.data(), sizeof(int64_t) * off.size());
  return MFX_OK;
}

