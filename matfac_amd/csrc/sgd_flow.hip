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
constexpr int FL_SLOT_SHIFT = 26;     // tagged schedule: queue record .y = owned row | LDS slot << 26
constexpr int FL_ROW_MASK = (1 << FL_SLOT_SHIFT) - 1;

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
  uint32_t *k0 = nullptr, *k1 = nullptr, *v0 = nullptr, *v1 = nullptr;
  int4* pack = nullptr;               // device builder: the queue record of every list position, in list order
  char* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  int64_t dcap = 0;
  int32_t *degU = nullptr, *degI = nullptr, *downer = nullptr;
  int64_t* dstart = nullptr;
  int64_t deg_cap = 0, own_cap = 0;   // capacities of degU (nU + nI) and of downer / dstart (max(nU, nI) [+ 1]), tracked separately
  std::vector<int32_t> hdeg, hpacked;
  std::vector<int64_t> hstart;
  // tagged schedule (sgd_flow_tag_kernel): granule copy {value, tag} of the other side's table; LDS slot of every owned row
  float* tagbuf = nullptr;
  int64_t tag_cap = 0;                // floats
  // hybrid ownership (sgd_flow_wide_kernel<.., HYB>): granule copy of the OWNED side's table too, rank | pub of every queue record,
  // the number of item queues (heavy users' queues follow them)
  float* tagbuf2 = nullptr;
  int64_t tag2_cap = 0;
  uint32_t* qa = nullptr;
  int64_t qa_cap = 0;
  int g_item = 0, n_heavy = 0;
  bool hybrid = false;
  // ... built on the device (build_flow_hybrid_device): queue of every user (-1: not heavy), rank << 1 | pub of every list position in
  // its item's chain and of its queue record; the heavy set and the item queues are cached per train matrix like the assignment below
  int32_t* duq = nullptr;
  uint32_t *hy_ri = nullptr, *hy_qa = nullptr;
  int64_t hy_cap = 0, duq_cap = 0;
  uint64_t hy_gen = ~0ull;
  int64_t hy_groups = -1, hy_nU = -1, hy_nI = -1, hy_count = -1;
  int hy_heavy_max = -1, hy_heavy_min = -1, hy_H = 0;
  std::vector<uint8_t> oslot;
  bool tagged = false;                // the queues of the last build carry the LDS slot of the owned row in .y
  // owner assignment (owned rows -> queues, LDS slots), cached per train matrix: it depends on the rows' chain lengths only for
  // BALANCE, never for correctness, and the lists replayed over one matrix are permutations of its ratings
  uint64_t assign_gen = ~0ull;
  int64_t assign_groups = -1, assign_nU = -1, assign_nI = -1, assign_count = -1;
  int assign_own_user = 0, assign_want = -2;
  int32_t assign_maxU = 0, assign_maxI = 0;
};
FlowState* fl(mfx_ctx* ctx) { return (FlowState*)ctx->flow; }

__global__ void flow_gather_kernel(const uint32_t* __restrict__ lpos, const uint32_t* __restrict__ vexp, int64_t n,
                                   const int32_t* __restrict__ eu, const int32_t* __restrict__ ei, const float* __restrict__ er,
                                   int own_user, const int32_t* __restrict__ owner, int4* __restrict__ q) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    const uint32_t s = lpos[t];
    const int u = eu[s], i = ei[s];
    const uint32_t slot = owner ? (uint32_t)owner[own_user ? u : i] >> FL_SLOT_SHIFT : 0u;   // tagged schedule: LDS slot of the owned row
    q[t] = make_int4(own_user ? i : u, (int)((uint32_t)(own_user ? u : i) | slot << FL_SLOT_SHIFT), __float_as_int(er[s]), (int)vexp[t]);
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

// ---------------------------------------------------------------------------------------------------------------------
// The tagged schedule (default for the plain update, K <= 256): the same queues, a faster hand-off.
//
// The version-counter kernel above pays four dependent memory round trips per visit (record, version, rows, store
// acknowledgement before the version may move): 1.9 us per step of the hottest item's chain, 93 ms per C2 epoch.  Here
//   * the other side's table is used through a GRANULE COPY: every float travels as an aligned 8-byte {value, tag} pair,
//     tag = number of visits the row has received so far in this epoch.  A visit that expects version k re-reads the row
//     until all its tags equal k -- the data is the flag (cdna_hip_programming.md Guideline 16, form R2): no separate
//     version word, no acknowledgement wait, one write-through store instruction per 16 bytes and lane;
//   * ONE QUEUE PER WAVEFRONT (lanes 0 .. L-1 hold the rows, the other lanes ride along with offsets behind the buffers):
//     everything about a record is wave-uniform, so records are read out of a 64-record register block with v_readlane into
//     scalar registers, row offsets are scalar arithmetic and every branch is a scalar branch.  (A first version kept four
//     16-lane queues per wave with their windows in LDS: its 780 instructions and 15 dependent LDS round trips per iteration
//     made the hottest queue's wave ISSUE-bound at 0.9 us per visit although 95 % of its iterations retired two visits and
//     the counted wait cost nothing; float instead of double arithmetic changed 5 %.);
//   * the rows of the next LA queue positions are requested while the head is visited (a static pipeline: LA landing
//     register sets, the loop unrolled LA times), so along an item's chain the user rows are on chip when their turn comes;
//   * the owned row of consecutive visits stays in REGISTERS; other owned rows of the queue live in the wave's LDS (QR
//     slots, placed by the queue builder) and go back to their table when they are displaced and at the end;
//   * all polls, record loads and tagged stores are inline assembly the compiler's wait-count pass does not see (it would
//     turn every wait into vmcnt(0), i.e. wait for the stores' acknowledgements): a step waits with a COUNTED
//     s_waitcnt vmcnt(N), N = the vector-memory instructions that are guaranteed to have been issued behind the poll it
//     needs -- every step issues the same number, an unused one with an offset behind its buffer (no memory access);
//   * a position whose row does not carry the expected tags yet is probed with ONE 16-byte load (lane 0, first granule)
//     until the tag shows up, then read whole: a stalled queue costs a 64-byte request per round trip, not a row.
// Progress: as above -- the earliest unvisited rating of the list is at the head of its queue, and a wave that waits, waits
// for its queue head only.
typedef int desc4 __attribute__((ext_vector_type(4)));
// Diagnostic build -DMFX_FLOW_STATS (scripts/flow_stats.sh): per queue {visits, visits that had to wait for their row,
// probes, owned-row switches, owned rows loaded from the table, shader cycles in the loop, cycles spent waiting for rows, blocks}
#ifdef MFX_FLOW_STATS
__device__ unsigned long long fl_stats[65536 * 8];
#define FL_STAT(i, v) st_[i] += (v)
#else
#define FL_STAT(i, v)
#endif
constexpr uint32_t FL_OOB = 0xfffffff0u;       // behind every buffer: loads return zeros, stores are dropped

__device__ __forceinline__ desc4 fl_desc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  return desc4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
template <bool SC1>
__device__ __forceinline__ uint4v fl_load(desc4 rs, uint32_t off) {
  uint4v v;
  if (SC1) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen sc1" : "=v"(v) : "v"(off), "s"(rs) : "memory");
  else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(rs) : "memory");
  return v;
}
// the load into a LANDING register of the request pipeline: in-out operand (see fw_load2 below)
__device__ __forceinline__ void fl_land(uint4v& v, desc4 rs, uint32_t off) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen sc1" : "+v"(v) : "v"(off), "s"(rs) : "memory");
}
template <bool SC1>
__device__ __forceinline__ void fl_store(desc4 rs, uint32_t off, uint4v v) {
  // (s_nop 1: a store of more than 64 bits needs a wait state before its data registers may be overwritten)
  if (SC1) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_nop 1" ::"v"(v), "v"(off), "s"(rs) : "memory");
  else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(off), "s"(rs) : "memory");
}

// The launch ends with its longest queue (C2: the most popular item's 49 777 visits, ten times the average queue), and that queue's wave
// shares its SIMD with three others: its ~ 70 instructions per visit then issue at ~ 10 cycles each.  Waves whose queue is more than twice
// the average raise their priority (own_flags bit 1; MFX_FLOW_PRIO=0 is the A/B): they are few, and whenever one is ready it issues.
__device__ __forceinline__ void fl_prio(int own_flags, uint32_t mine, const int64_t* __restrict__ qoff) {
  if (own_flags & 2) {
    const uint32_t ng = gridDim.x * (uint32_t)(FL_WG / 64);
    const uint64_t total = (uint64_t)qoff[ng];
    if ((uint64_t)mine * ng > 2ull * total) __builtin_amdgcn_s_setprio(3);
  }
}

template <int L, int C>
struct F2 {
  static constexpr int LD = 4 * L * C;
  static constexpr int LA = C <= 2 ? 4 : 2;        // rows of the other side in flight (queue positions ahead of the head)
  static constexpr int QR = C <= 2 ? 32 : 16;      // owned rows kept in LDS per queue
  static constexpr int NB = 64;                    // records per register block
  static constexpr int WS = QR * LD * 4 + QR * 8;  // LDS bytes per wave: QR rows | their row ids | their versions (hybrid ownership)
  static constexpr int LDS = WS * (FL_WG / 64);
  static constexpr int WGS = C <= 1 ? 4 : 2;       // workgroups per CU (registers: launch bounds; LDS: LDS * WGS <= 160 KiB)
  static constexpr int NSTEP = 4 * C;              // vector-memory instructions of a step: 2 C tagged stores, 2 C row requests
  static_assert(LDS * WGS <= 163840 && (LA - 1) * NSTEP <= 63 && NB % LA == 0, "tagged dataflow configuration");
};

template <int L, int C, int ARITH>
__global__ __launch_bounds__(FL_WG, (F2<L, C>::WGS)) void sgd_flow_tag_kernel(const int4* __restrict__ q, const int64_t* __restrict__ qoff,
                                                                              uint32_t qbytes, float* T, uint32_t tbytes, float* O, uint32_t obytes,
                                                                              int own_flags, float lr, float uReg, float iReg, unsigned* flag,
                                                                              const uint32_t* /*qa: hybrid ownership, wide kernel only*/, int /*g_item*/) {
  typedef F2<L, C> P;
  const int own_user = own_flags & 1;
  constexpr int LD = P::LD, LA = P::LA, QR = P::QR, NB = P::NB;
  extern __shared__ __attribute__((aligned(16))) char fl_smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane % L;
  const bool act = lane < L;                     // the lanes that hold the rows; the others ask for nothing and store nothing
  float4v* qv = (float4v*)(fl_smem + (size_t)wv * P::WS);          // [QR][C][L]
  int* qt = (int*)(fl_smem + (size_t)wv * P::WS + QR * LD * 4);    // [QR]
  const int64_t grp = (int64_t)blockIdx.x * (FL_WG / 64) + __builtin_amdgcn_readfirstlane(wv);
  uint32_t pos = (uint32_t)qoff[grp];
  const uint32_t end = (uint32_t)qoff[grp + 1];
  fl_prio(own_flags, end - pos, qoff);
  if (lane < QR) qt[lane] = -1;
  const desc4 dq = fl_desc(q, qbytes), dt = fl_desc(T, tbytes), dob = fl_desc(O, obytes);
  const uint32_t lane_off = act ? (uint32_t)(j * 16) : FL_OOB;       // this lane's 16 bytes inside a 16 L-byte piece
  uint4v Tr[LA][2 * C];                          // landing registers: slot = queue position % LA
#pragma unroll
  for (int k = 0; k < LA; k++)
#pragma unroll
    for (int h = 0; h < 2 * C; h++) Tr[k][h] = uint4v{0u, 0u, 0u, 0u};
  float4v ov[C];                                 // the owned row of the last visit (registers), its id and its LDS slot
#pragma unroll
  for (int c = 0; c < C; c++) ov[c] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
  int cur_row = -1, cur_slot = 0;
  bool aborted = false;
#ifdef MFX_FLOW_STATS
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#endif

  // request the row of record `rc` (row id in .x) into landing slot K, or nothing (offsets behind the buffer) when !live
#define FL_REQUEST(K, ROWX, LIVE)                                                                                     \
  {                                                                                                                   \
    const uint32_t rb_ = (LIVE) ? (uint32_t)(ROWX) * (uint32_t)(8 * LD) : FL_OOB;                                     \
    _Pragma("unroll") for (int c = 0; c < C; c++) _Pragma("unroll") for (int h = 0; h < 2; h++)                       \
        fl_land(Tr[K][2 * c + h], dt, ((LIVE) && act) ? rb_ + (uint32_t)(c * 32 * L + h * 16 * L) + lane_off : FL_OOB);   \
  }
  // one queue position: S = index in the block (uniform), K = S % LA (static), NWAIT = requests and stores issued behind its poll
#define FL_STEP(K, S, NWAIT)                                                                                          \
  {                                                                                                                   \
    const int s_ = (S);                                                                                               \
    const bool live_ = s_ < nb;                                                                                       \
    uint32_t stbase_ = FL_OOB, ntag_ = 0;                                                                             \
    float4v tv[C];                                                                                                    \
    _Pragma("unroll") for (int c = 0; c < C; c++) tv[c] = float4v{0.0f, 0.0f, 0.0f, 0.0f};                            \
    if (live_ && !aborted) {                                                                                          \
      const int sl_ = s_ & 63;                                                                                        \
      const int rx_ = __builtin_amdgcn_readlane(rec.x, sl_), ry_ = __builtin_amdgcn_readlane(rec.y, sl_);             \
      const int rz_ = __builtin_amdgcn_readlane(rec.z, sl_);                                                          \
      const uint32_t exp_ = (uint32_t)__builtin_amdgcn_readlane(rec.w, sl_);                                          \
      const int orow_ = ry_ & FL_ROW_MASK, slot_ = (int)((uint32_t)ry_ >> FL_SLOT_SHIFT) & (QR - 1);                  \
      if (orow_ != cur_row) {               /* another owned row: the held one goes to its LDS slot, this one comes in */ \
        FL_STAT(3, 1);                                                                                                \
        if (cur_row >= 0 && act) {                                                                                    \
          _Pragma("unroll") for (int c = 0; c < C; c++) qv[(cur_slot * C + c) * L + j] = ov[c];                       \
        }                                                                                                             \
        const int have_ = __builtin_amdgcn_readfirstlane(qt[slot_]);                                                  \
        if (have_ == orow_) {                                                                                         \
          _Pragma("unroll") for (int c = 0; c < C; c++) ov[c] = qv[(slot_ * C + c) * L + j];                          \
        } else {                            /* not in LDS: the slot's row goes back to the table, this one is loaded */ \
          FL_STAT(4, 1);                                                                                              \
          if (have_ >= 0) {                                                                                           \
            _Pragma("unroll") for (int c = 0; c < C; c++)                                                             \
                fl_store<false>(dob, act ? (uint32_t)have_ * (uint32_t)(4 * LD) + (uint32_t)(c * 16 * L) + lane_off : FL_OOB, \
                                __builtin_bit_cast(uint4v, qv[(slot_ * C + c) * L + j]));                             \
          }                                                                                                           \
          uint4v in_[C];                                                                                              \
          _Pragma("unroll") for (int c = 0; c < C; c++)                                                               \
              in_[c] = fl_load<false>(dob, act ? (uint32_t)orow_ * (uint32_t)(4 * LD) + (uint32_t)(c * 16 * L) + lane_off : FL_OOB); \
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                            \
          _Pragma("unroll") for (int c = 0; c < C; c++) {                                                             \
            asm volatile("" : "+v"(in_[c]));                                                                          \
            ov[c] = __builtin_bit_cast(float4v, in_[c]);                                                              \
          }                                                                                                           \
          if (lane == 0) qt[slot_] = orow_;                                                                           \
        }                                                                                                             \
        cur_row = orow_;                                                                                              \
        cur_slot = slot_;                                                                                             \
      }                                                                                                               \
      /* the row of the other side: requested LA positions ago */                                                     \
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NWAIT) : "memory");                                                    \
      _Pragma("unroll") for (int h = 0; h < 2 * C; h++) asm volatile("" : "+v"(Tr[K][h]));                            \
      bool ok_ = !act;                                                                                                \
      {                                                                                                               \
        bool t_ = true;                                                                                               \
        _Pragma("unroll") for (int h = 0; h < 2 * C; h++) t_ = t_ && Tr[K][h].y == exp_ && Tr[K][h].w == exp_;        \
        ok_ = ok_ || t_;                                                                                              \
      }                                                                                                               \
      if (__builtin_amdgcn_ballot_w64(ok_) != ~0ull) {       /* not there yet: probe the first granule, then read the row again */ \
        FL_STAT(1, 1);                                                                                                \
        FL_STATS_T0                                                                                                   \
        const uint32_t rb_ = (uint32_t)rx_ * (uint32_t)(8 * LD);                                                      \
        long long t_last_ = wall_clock64();                                                                           \
        int spins_ = 0;                                                                                               \
        for (;;) {                                                                                                    \
          uint4v pr_ = fl_load<true>(dt, lane == 0 ? rb_ : FL_OOB);                                                   \
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(pr_)::"memory");                                                   \
          FL_STAT(2, 1);                                                                                              \
          if ((uint32_t)__builtin_amdgcn_readfirstlane(pr_.y) == exp_) {                                              \
            _Pragma("unroll") for (int c = 0; c < C; c++) _Pragma("unroll") for (int h = 0; h < 2; h++)               \
                fl_land(Tr[K][2 * c + h], dt, act ? rb_ + (uint32_t)(c * 32 * L + h * 16 * L) + lane_off : FL_OOB);       \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
            _Pragma("unroll") for (int h = 0; h < 2 * C; h++) asm volatile("" : "+v"(Tr[K][h]));                      \
            bool t_ = true;                                                                                           \
            _Pragma("unroll") for (int h = 0; h < 2 * C; h++) t_ = t_ && Tr[K][h].y == exp_ && Tr[K][h].w == exp_;    \
            if (__builtin_amdgcn_ballot_w64(t_ || !act) == ~0ull) break;                                              \
          }                                                                                                           \
          if ((++spins_ & 63) == 0) {                                                                                 \
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { aborted = true; break; }       \
            if (wall_clock64() - t_last_ > 200000000LL) {      /* 100 MHz constant clock: 2 s on one queue head */    \
              __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                               \
              aborted = true;                                                                                         \
              break;                                                                                                  \
            }                                                                                                         \
          }                                                                                                           \
        }                                                                                                             \
        FL_STATS_T1                                                                                                   \
      }                                                                                                               \
      if (!aborted) {                                                                                                 \
        _Pragma("unroll") for (int c = 0; c < C; c++)                                                                 \
            tv[c] = float4v{__uint_as_float(Tr[K][2 * c].x), __uint_as_float(Tr[K][2 * c].z), __uint_as_float(Tr[K][2 * c + 1].x), \
                            __uint_as_float(Tr[K][2 * c + 1].z)};                                                     \
        if (own_user) {                                                                                               \
          const float est_ = group_dot<L, C>(ov, tv);                                                                 \
          sgd_axpys<C, ARITH>(ov, tv, __int_as_float(rz_), est_, lr, uReg, iReg);                                     \
        } else {                                                                                                      \
          const float est_ = group_dot<L, C>(tv, ov);                                                                 \
          sgd_axpys<C, ARITH>(tv, ov, __int_as_float(rz_), est_, lr, uReg, iReg);                                     \
        }                                                                                                             \
        stbase_ = act ? (uint32_t)rx_ * (uint32_t)(8 * LD) + lane_off : FL_OOB;                                       \
        ntag_ = exp_ + 1u;                                                                                            \
        FL_STAT(0, 1);                                                                                                \
      }                                                                                                               \
    }                                                                                                                 \
    /* the other side's row with its new tag, write-through (always 2 C instructions), then the request for position S + LA */ \
    _Pragma("unroll") for (int c = 0; c < C; c++) _Pragma("unroll") for (int h = 0; h < 2; h++)                       \
        fl_store<true>(dt, stbase_ == FL_OOB ? FL_OOB : stbase_ + (uint32_t)(c * 32 * L + h * 16 * L),               \
                       uint4v{__float_as_uint(tv[c][2 * h]), ntag_, __float_as_uint(tv[c][2 * h + 1]), ntag_});      \
    {                                                                                                                 \
      const int sn_ = s_ + LA;                                                                                        \
      const bool ln_ = sn_ < nb && !aborted;                                                                          \
      const int rxn_ = __builtin_amdgcn_readlane(rec.x, sn_ & 63);                                                    \
      FL_REQUEST(K, rxn_, ln_)                                                                                        \
    }                                                                                                                 \
  }
#ifdef MFX_FLOW_STATS
#define FL_STATS_T0 const unsigned long long st_w = __builtin_amdgcn_s_memtime();
#define FL_STATS_T1 st_[6] += __builtin_amdgcn_s_memtime() - st_w;
#else
#define FL_STATS_T0
#define FL_STATS_T1
#endif

  while (pos < end && !aborted) {
    // ---- a block of up to 64 records: lane l holds record pos + l ----------------------------------------------------
    const int nb = (int)min((uint32_t)NB, end - pos);
    uint4v rb = fl_load<false>(dq, lane < nb ? (pos + (uint32_t)lane) * 16u : FL_OOB);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb)::"memory");
    const int4 rec = make_int4((int)rb.x, (int)rb.y, (int)rb.z, (int)rb.w);
    FL_STAT(7, 1);
    // the first LA rows
#pragma unroll
    for (int k = 0; k < LA; k++) {
      const int rxn = __builtin_amdgcn_readlane(rec.x, k);
      FL_REQUEST(k, rxn, k < nb)
    }
    // the first LA positions: behind the request of position k lie the later requests of the prologue and k full steps
    if constexpr (LA == 4) {
      FL_STEP(0, 0, 3 * 2 * C + 0 * P::NSTEP)
      FL_STEP(1, 1, 2 * 2 * C + 1 * P::NSTEP)
      FL_STEP(2, 2, 1 * 2 * C + 2 * P::NSTEP)
      FL_STEP(3, 3, 0 * 2 * C + 3 * P::NSTEP)
    } else {
      FL_STEP(0, 0, 1 * 2 * C + 0 * P::NSTEP)
      FL_STEP(1, 1, 0 * 2 * C + 1 * P::NSTEP)
    }
    for (int s0 = LA; s0 < nb; s0 += LA) {
      if constexpr (LA == 4) {
        FL_STEP(0, s0, 3 * P::NSTEP)
        FL_STEP(1, s0 + 1, 3 * P::NSTEP)
        FL_STEP(2, s0 + 2, 3 * P::NSTEP)
        FL_STEP(3, s0 + 3, 3 * P::NSTEP)
      } else {
        FL_STEP(0, s0, P::NSTEP)
        FL_STEP(1, s0 + 1, P::NSTEP)
      }
    }
    // Nothing may be in flight when the landing registers leave the pipeline: the requests of the block's last LA steps (offsets
    // behind the buffer: zeros) still WRITE their registers when they return, and behind this point the compiler is free to use
    // those registers for something else -- a flush address, as the first C2 run of this kernel showed with a store to address 0.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < LA; k++)
#pragma unroll
      for (int h = 0; h < 2 * C; h++) asm volatile("" : "+v"(Tr[k][h]));
    pos += (uint32_t)nb;
  }
#undef FL_STEP
#undef FL_REQUEST
#ifdef MFX_FLOW_STATS
  st_[5] = __builtin_amdgcn_s_memtime() - st_t0;
  if (lane == 0 && grp < 65536)
    for (int i = 0; i < 8; i++) fl_stats[grp * 8 + i] = st_[i];
#endif
  // the owned rows go back to their table: the held one through its LDS slot, then every slot in use
  if (cur_row >= 0 && act) {
#pragma unroll
    for (int c = 0; c < C; c++) qv[(cur_slot * C + c) * L + j] = ov[c];
  }
  for (int sI = 0; sI < QR; sI++) {
    const int have = __builtin_amdgcn_readfirstlane(qt[sI]);
    if (have >= 0 && act) {
#pragma unroll
      for (int c = 0; c < C; c++) *(float4v*)(O + (int64_t)have * LD + c * 4 * L + 4 * j) = qv[(sI * C + c) * L + j];
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The tagged schedule, ONE ELEMENT PER LANE AND CHUNK (rows of 64 C floats, K > 32; round 3).  In the kernel above lanes 0 .. 15
// hold a row (4 elements per chunk each) and 48 lanes ride along: the wave that works off the longest queue issues ~ 220
// instructions per visit, 140 of them the reference's double bracket on its 4 C elements -- and a wave issues one vector
// instruction per ~ 4.5 cycles, so the hottest chain ran at 1 000 cycles per visit.  Here lane l = 4 j + x holds element 4 j + x
// of every 64-element chunk (the element lane j of the 16-lane layout holds in position x), all 64 lanes work, and the bracket is
// ONE element per lane and chunk.  The dot product keeps the device order of include/mfx.h bit for bit: the 4 C-long fma chain
// of "lane j" now hops through the four lanes of quad j (a quad_perm rotate between links), and the xor butterfly over j = 1, 2, 4, 8
// becomes row_half_mirror, row_mirror (all four lanes of a quad hold the quad's value), v_permlane16_swap and v_permlane32_swap.
// The granule copy is the one flow_tag_kernel writes for L = 16; a lane reads and writes ITS 8-byte granule {value, tag}.
// Everything else (queues, records by v_readlane, LA-deep request pipeline with counted waits, owned rows in registers / LDS
// slots, probe-then-poll) is the kernel above.
template <int C>
struct FW {
  static constexpr int LD = 64 * C;
  static constexpr int LA = 4;                     // rows of the other side in flight (2 C landing registers each)
  static constexpr int QR = C <= 2 ? 32 : 16;      // owned rows kept in LDS per queue
  static constexpr int NB = 64;
  static constexpr int WS = QR * LD * 4 + QR * 8;
  static constexpr int LDS = WS * (FL_WG / 64);
  static constexpr int WGS = C <= 1 ? 4 : 2;
  static constexpr int NSTEP = 2 * C;              // vector-memory instructions of a step: C tagged stores, C row requests
  static_assert(LDS * WGS <= 163840 && (LA - 1) * NSTEP <= 63 && NB % LA == 0, "wide tagged dataflow configuration");
};
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
// A landing register is an IN-OUT operand of every load into it and of every wait behind it: one live range from the first
// request to the last use, so the register allocator has no second definition to connect with a copy.  (With plain outputs the
// rotation of the unrolled pipeline gave the slot-0 register of the prologue and of the steady state different registers and a
// v_mov between them IN FRONT of the counted wait -- a copy of a register whose load had not landed: wrong rows, first seen in
// this kernel.)  tests/test_trips_cpu.py checks the compiled kernels for it.
__device__ __forceinline__ void fw_load2(unsigned long long& v, desc4 rs, uint32_t off) {          // sc1: past the CU's L1
  asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen sc1" : "+v"(v) : "v"(off), "s"(rs) : "memory");
}
__device__ __forceinline__ void fw_store2(desc4 rs, uint32_t off, uint2v v) {  // sc1: write-through
  asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen sc1" ::"v"(v), "v"(off), "s"(rs) : "memory");
}
__device__ __forceinline__ uint32_t fw_load1(desc4 rs, uint32_t off) {
  uint32_t v;
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(rs) : "memory");
  return v;
}
__device__ __forceinline__ void fw_store1(desc4 rs, uint32_t off, uint32_t v) {
  asm volatile("buffer_store_dword %0, %1, %2, 0 offen" ::"v"(v), "v"(off), "s"(rs) : "memory");
}
// a quad permutation writes every lane: the form without an "old" operand needs no initialising move
template <int CTRL>
__device__ __forceinline__ float fw_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// p.q in device order (include/mfx.h) with lane 4 j + x holding element x of "lane j" in every chunk
template <int C>
__device__ __forceinline__ float wide_dot(const float (&p)[C], const float (&q)[C]) {
  float a = 0.0f;
#pragma unroll
  for (int c = 0; c < C; c++)
#pragma unroll
    for (int x = 0; x < 4; x++) {
      if (c || x) a = fw_dpp<0x93>(a);             // quad_perm [3,0,1,2]: the link of lane x takes the chain from lane x - 1 (x = 0: from 3)
      a = __builtin_fmaf(p[c], q[c], a);           // (every lane computes; lane x holds the chain)
    }
  float s = fw_dpp<0xFF>(a);                       // quad_perm [3,3,3,3]: the finished chain of quad j, in its four lanes
  s = s + dpp_f<0x141>(s);                         // j xor 1: the partner quad of the 8-lane half   (row_half_mirror)
  s = s + dpp_f<0x140>(s);                         // j xor 2: the partner half of the 16-lane row   (row_mirror)
  {                                                // j xor 4: the partner row of the 32-lane half   (odd rows of a <-> even rows of b)
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {                                                // j xor 8: the other half of the wave
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return s;
}
// sgd_axpys on one element per chunk: user row p first, then the item row q with the updated p (modelMF.cpp:94-103)
template <int C, int ARITH>
__device__ __forceinline__ void wide_axpys(float (&p)[C], float (&q)[C], float r, float est, float lr, float uReg, float iReg) {
  if (ARITH == MFX_ARITH_F32) {
    const float c1 = -2.0f * (r - est);
    const float cu = 2.0f * uReg, ci = 2.0f * iReg;
#pragma unroll
    for (int c = 0; c < C; c++) {
      p[c] = upd_f32(p[c], q[c], c1, cu, lr);
      q[c] = upd_f32(q[c], p[c], c1, ci, lr);
    }
  } else {
    double diff;
    if (ARITH == MFX_ARITH_REF64F) { const float d = r - est; diff = (double)d; }
    else diff = (double)r - (double)est;
    const double m2 = -2.0 * diff;
    const double ru = 2.0 * (double)uReg, ri = 2.0 * (double)iReg, lrd = (double)lr;
#pragma unroll
    for (int c = 0; c < C; c++) {
      p[c] = upd_ref64(p[c], q[c], m2, ru, lrd);
      q[c] = upd_ref64(q[c], p[c], m2, ri, lrd);
    }
  }
}

// ---- POLE BLOCKS (round 4).  The launch lasts as long as its longest queue, and that queue is one popular row's visits back to back
// (C2: 49 777 of them): a dependent chain.  scripts/chain_probe.hip measured what one wave pays per LINK of such a chain on this chip:
// 8.5 - 9 cycles for any dependent VALU instruction (f32 or f64 alike), 16 with a DPP read of a register just written, 36 for
// mov + v_permlane32_swap + add, and 26 cycles for a scalar compare + branch even when it is NOT taken.  The generic step above spends
// ~ 690 cycles per visit: nine branches (owned-row switch, live / abort / tag checks, own_user), ~ 48 vector instructions, a dot chain of
// 4 fma + 4 DPP moves and two swap levels.  A block of 64 records that all visit ONE owned row runs the step below instead:
//   * one branch per visit -- "the NEXT position's row had not arrived with its tags": checked early, taken (to the probe loop) only
//     AFTER this visit's store, so nothing of ours that another queue may wait for is held back;
//   * the dot chain of "lane j" (include/mfx.h) without moves: every lane of quad j runs all four links, reading the owned element of
//     link x through the DPP source of v_fmac (quad_perm:[x,x,x,x]) and the other side's from four registers gathered off the chain:
//     the same products in the same order, so the same bits, and the quad broadcast of the finished chain falls away;
//   * the levels j ^ 4 and j ^ 8 as v_add_f32 row_bcast:15 (rows 1, 3) and row_bcast:31 (row 3): lane 63 ends with
//     (R3 + R2) + (R1 + R0) -- the butterfly's (R0 + R1) + (R2 + R3) up to commutations, i.e. the same float -- read with one v_readlane;
//   * the reference's  m2*y + reg2*x  (modelMF.cpp:96 / :102 in double) as fma(reg2, x, m2*y): reg2 and x are floats widened to double,
//     their product has <= 48 significant bits and is EXACT, so the fused form rounds once where the unfused one rounds once -- same bits,
//     one multiplication fewer per side.
// Everything else (queues, landing registers, counted waits, tags, probe-then-poll) is the kernel's.  MFX_FLOW_POLE=0: generic steps only.
template <int X>
__device__ __forceinline__ void fwp_link(float& a, float own, float oth, bool first) {
  // (s_nop 1: `own` may have been written by the previous visit's last conversion; a DPP read wants two wait states and the hazard
  //  recognizer does not look inside an asm statement)
  (void)first;
  if (X == 0) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(own), "v"(oth));
  if (X == 1) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(own), "v"(oth));
  if (X == 2) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(own), "v"(oth));
  if (X == 3) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(own), "v"(oth));
}
// LANDING REGISTERS OF THE POLE PATH ARE ACCUMULATION REGISTERS (a0 .. a31), named in the asm text.  The generic steps keep theirs as
// C++ values that are in-out operands of every request and wait, and rely on the register allocator never copying one (checked on
// the compiled code, tests/test_trips_cpu.py).  The first build of the pole path did it the same way and that property was gone for
// the whole kernel: SROA turned the array into one <4 x i64> value (an 8-register tuple, copied as a whole behind every re-read, the
// slot in flight included), and as sixteen separate variables the allocator still moved slots between registers on the edges of the
// step loop -- with their loads in flight.  A register the compiler never sees cannot be copied: slot K, chunk c lands in
// a[8 K + 2 c : 8 K + 2 c + 1] = {value, tag}; a request is `buffer_load_dwordx2 a[..]`, a step reads the pair out with
// v_accvgpr_read_b32 BEHIND its counted wait, inside the same asm statement.  The kernel uses no other accumulation registers (no
// MFMA, no spills: checked in tests/test_trips_cpu.py).
#define FWA_CLOBBER "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", \
                    "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
template <int IDX>
__device__ __forceinline__ void fwa_load(desc4 rs, uint32_t off) {
  static_assert(IDX >= 0 && IDX < 16, "landing pairs a[2 IDX : 2 IDX + 1] of a0 .. a31");
  asm volatile("buffer_load_dwordx2 a[%2:%3], %0, %1, 0 offen sc1" ::"v"(off), "s"(rs), "i"(2 * IDX), "i"(2 * IDX + 1) : "memory", FWA_CLOBBER);
}
template <int IDX>
__device__ __forceinline__ void fwa_read(uint32_t& val, uint32_t& tag) {
  static_assert(IDX >= 0 && IDX < 16, "landing pairs a[2 IDX : 2 IDX + 1] of a0 .. a31");
  asm volatile("v_accvgpr_read_b32 %0, a%2\n\tv_accvgpr_read_b32 %1, a%3" : "=v"(val), "=v"(tag) : "i"(2 * IDX), "i"(2 * IDX + 1) : "memory");
}
// the C requests of one row into slot K (row byte offset rb, or behind the buffer), and the C {value, tag} pairs of slot K;
// slot K, chunk c = pair K C + c
template <int K, int C>
__device__ __forceinline__ void fwa_request(desc4 rs, uint32_t rb) {
  fwa_load<K * C + 0>(rs, rb);
  if constexpr (C > 1) fwa_load<K * C + 1>(rs, rb == FL_OOB ? FL_OOB : rb + 512u);
  if constexpr (C > 2) fwa_load<K * C + 2>(rs, rb == FL_OOB ? FL_OOB : rb + 1024u);
  if constexpr (C > 3) fwa_load<K * C + 3>(rs, rb == FL_OOB ? FL_OOB : rb + 1536u);
}
template <int K, int C>
__device__ __forceinline__ void fwa_take(uint32_t (&val)[C], uint32_t (&tag)[C]) {
  fwa_read<K * C + 0>(val[0], tag[0]);
  if constexpr (C > 1) fwa_read<K * C + 1>(val[1], tag[1]);
  if constexpr (C > 2) fwa_read<K * C + 2>(val[2], tag[2]);
  if constexpr (C > 3) fwa_read<K * C + 3>(val[3], tag[3]);
}
// the requests of queue positions K .. N - 1 of a block (row ids in lane k of recx)
template <int K, int N, int C>
__device__ __forceinline__ void fwp_prologue(desc4 rs, int recx, uint32_t rowbytes, uint32_t g_off) {
  if constexpr (K < N) {
    fwa_request<K, C>(rs, (uint32_t)__builtin_amdgcn_readlane(recx, K) * rowbytes + g_off);
    fwp_prologue<K + 1, N, C>(rs, recx, rowbytes, g_off);
  }
}
// p.q in device order, uniform result (see above); own = the owned row (one element per lane and chunk), g = the other side's
// elements of this lane's quad, g[c][x] = element x
template <int C>
__device__ __forceinline__ float pole_dot(const float (&own)[C], const float (&g)[C][4]) {
  float a = 0.0f;
#pragma unroll
  for (int c = 0; c < C; c++) {
    fwp_link<0>(a, own[c], g[c][0], c == 0);
    fwp_link<1>(a, own[c], g[c][1], false);
    fwp_link<2>(a, own[c], g[c][2], false);
    fwp_link<3>(a, own[c], g[c][3], false);
  }
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(a));
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a));
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a));
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(a));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
}
// sgd_axpys on one element per chunk with the exact-product fma (see above); bit for bit wide_axpys
template <int C, int ARITH>
__device__ __forceinline__ void pole_axpys(float (&p)[C], float (&q)[C], float r, float est, float lr, float uReg, float iReg) {
  if (ARITH == MFX_ARITH_F32) {
    wide_axpys<C, ARITH>(p, q, r, est, lr, uReg, iReg);
  } else {
    double diff;
    if (ARITH == MFX_ARITH_REF64F) { const float d = r - est; diff = (double)d; }
    else diff = (double)r - (double)est;
    const double m2 = -2.0 * diff;
    const double ru = 2.0 * (double)uReg, ri = 2.0 * (double)iReg, lrd = (double)lr;
#pragma unroll
    for (int c = 0; c < C; c++) {
      const double p64 = (double)p[c], q64 = (double)q[c];
      p[c] = (float)(p64 - lrd * __builtin_fma(ru, p64, m2 * q64));
      q[c] = (float)(q64 - lrd * __builtin_fma(ri, q64, m2 * (double)p[c]));
    }
  }
}

// HYB (round 4, MFX_FLOW_HYBRID): HYBRID OWNERSHIP.  Queues 0 .. g_item - 1 own item rows as before; queue g_item + h owns ONE heavy
// user's row and holds ALL of that user's ratings -- the rows whose hand-off chains ended the launch (the busiest user's 10 717
// ratings were 10 717 hand-offs between item queues; in a queue of its own they are 10 717 steps of a pole).  An item row is then
// visited from two kinds of queues, so BOTH factor tables are in granule form for the launch (T = users', O = items'), every row
// carries its version (visits received) as its tag, and an owner treats its rows as a CACHE of the table: a row that is not in
// registers / LDS is read from the table by polling for the version this visit expects (qa = rank of the visit in the owned row's
// chain << 1 | pub), a displaced row is written back with its version, and a row whose NEXT visit lies in another queue (pub) is
// written back behind its visit and dropped from the cache -- the queue builder's pub bit also ends the block, so publishing happens
// at block ends only.  The other side of a visit is handled exactly as before (tagged load LA positions ahead, tagged store).
template <int C, int ARITH, bool HYB = false>
__global__ __launch_bounds__(FL_WG, (FW<C>::WGS)) void sgd_flow_wide_kernel(const int4* __restrict__ q, const int64_t* __restrict__ qoff,
                                                                           uint32_t qbytes, float* T, uint32_t tbytes, float* O, uint32_t obytes,
                                                                           int own_flags, float lr, float uReg, float iReg, unsigned* flag,
                                                                           const uint32_t* __restrict__ qa, int g_item) {
  typedef FW<C> P;
  constexpr int LD = P::LD, LA = P::LA, QR = P::QR, NB = P::NB;
  extern __shared__ __attribute__((aligned(16))) char fl_smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* qv = (float*)(fl_smem + (size_t)wv * P::WS);              // [QR][C][64]
  int* qt = (int*)(fl_smem + (size_t)wv * P::WS + QR * LD * 4);    // [QR]
  uint32_t* qver = (uint32_t*)(qt + QR);                           // [QR] hybrid: version of the cached row
  const int64_t grp = (int64_t)blockIdx.x * (FL_WG / 64) + __builtin_amdgcn_readfirstlane(wv);
  uint32_t pos = (uint32_t)qoff[grp];
  const uint32_t end = (uint32_t)qoff[grp + 1];
  const int own_user = HYB ? (grp >= (int64_t)g_item ? 1 : 0) : (own_flags & 1);
  const int pollfull = (own_flags >> 8) & 0xffff;      // pole path: tries of a stalled head that read the whole row (MFX_FLOW_POLLFULL)
  fl_prio(own_flags, end - pos, qoff);
  if (lane < QR) qt[lane] = -1;
  // dt: the table of the OTHER side of this queue's visits (granule form); dob: the owned side's (plain floats, or granule form when HYB)
  const desc4 dq = fl_desc(q, qbytes), dt = (HYB && own_user) ? fl_desc(O, obytes) : fl_desc(T, tbytes),
              dob = (HYB && own_user) ? fl_desc(T, tbytes) : fl_desc(O, obytes), dqa = fl_desc(qa, HYB ? qbytes / 4u : 0u);
  uint32_t cur_ver = 0;                                 // hybrid: version of the row held in registers (visits it has received)
  // this lane's granule inside a 512-byte chunk of the granule copy, and its float inside a 256-byte chunk of the owned table
  const uint32_t g_off = (uint32_t)(((lane >> 1) & 1) * 256 + (lane >> 2) * 16 + (lane & 1) * 8), o_off = (uint32_t)lane * 4u;
  // the landing registers: one 64-bit INTEGER {tag : value} per pipeline slot and chunk.  (As an array of two-element vectors
  // the compiler promoted the whole array to one register tuple -- 8 registers at C = 1 -- and moved the tuple around between
  // steps, i.e. copied registers whose loads were still in flight; integers stay separate values.)
  unsigned long long Tr[LA][C];
#pragma unroll
  for (int k = 0; k < LA; k++)
#pragma unroll
    for (int c = 0; c < C; c++) Tr[k][c] = 0ull;
  float ov[C];
#pragma unroll
  for (int c = 0; c < C; c++) ov[c] = 0.0f;
  int cur_row = -1, cur_slot = 0;
  bool aborted = false;

  // ---- the owned side's table.  Plain mode: rows of floats (fw_load1 / fw_store1).  HYB: granule form with the row's version as
  // its tag -- OWN_FETCH polls until every granule of row ROW carries version VER, OWN_PUT writes VALS with version VER.
#define OWN_PUT(ROW, VER, VALS)                                                                                       \
  {                                                                                                                   \
    if constexpr (HYB) {                                                                                              \
      const uint32_t wb_ = (uint32_t)(ROW) * (uint32_t)(8 * LD) + g_off;                                              \
      _Pragma("unroll") for (int c = 0; c < C; c++) fw_store2(dob, wb_ + (uint32_t)(c * 512), uint2v{__float_as_uint(VALS(c)), (uint32_t)(VER)}); \
    } else {                                                                                                          \
      _Pragma("unroll") for (int c = 0; c < C; c++)                                                                   \
          fw_store1(dob, (uint32_t)(ROW) * (uint32_t)(4 * LD) + (uint32_t)(c * 256) + o_off, __float_as_uint(VALS(c))); \
    }                                                                                                                 \
  }
#define OWN_FETCH(ROW, VER)                                                                                           \
  {                                                                                                                   \
    if constexpr (HYB) {                                                                                              \
      const uint32_t fb_ = (uint32_t)(ROW) * (uint32_t)(8 * LD) + g_off;                                              \
      long long t_f_ = wall_clock64();                                                                                \
      int sp_ = 0;                                                                                                    \
      for (;;) {                                                                                                      \
        unsigned long long g_[C];                                                                                     \
        _Pragma("unroll") for (int c = 0; c < C; c++) { g_[c] = 0ull; fw_load2(g_[c], dob, fb_ + (uint32_t)(c * 512)); } \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
        bool t_ = true;                                                                                               \
        _Pragma("unroll") for (int c = 0; c < C; c++) { asm volatile("" : "+v"(g_[c])); t_ = t_ && (uint32_t)(g_[c] >> 32) == (uint32_t)(VER); } \
        if (__builtin_amdgcn_ballot_w64(t_) == ~0ull) {                                                               \
          _Pragma("unroll") for (int c = 0; c < C; c++) ov[c] = __uint_as_float((uint32_t)g_[c]);                     \
          break;                                                                                                      \
        }                                                                                                             \
        if ((++sp_ & 63) == 0) {                                                                                      \
          if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { aborted = true; break; }         \
          if (wall_clock64() - t_f_ > 200000000LL) {                                                                  \
            __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                                 \
            aborted = true;                                                                                           \
            break;                                                                                                    \
          }                                                                                                           \
        }                                                                                                             \
      }                                                                                                               \
    } else {                                                                                                          \
      uint32_t in_[C];                                                                                                \
      _Pragma("unroll") for (int c = 0; c < C; c++)                                                                   \
          in_[c] = fw_load1(dob, (uint32_t)(ROW) * (uint32_t)(4 * LD) + (uint32_t)(c * 256) + o_off);                 \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
      _Pragma("unroll") for (int c = 0; c < C; c++) {                                                                 \
        asm volatile("" : "+v"(in_[c]));                                                                              \
        ov[c] = __uint_as_float(in_[c]);                                                                              \
      }                                                                                                               \
    }                                                                                                                 \
  }
  // the owned row of the next visit: ROWY = the record's .y, KA = its rank in that row's chain (hybrid).  The held row goes to its LDS
  // slot, the wanted one comes from its slot, or from the table (where the slot's occupant goes first).
#define OWN_SWITCH(ROWY, KA)                                                                                          \
  {                                                                                                                   \
    const int orow_ = (ROWY) & FL_ROW_MASK, slot_ = (int)((uint32_t)(ROWY) >> FL_SLOT_SHIFT) & (QR - 1);              \
    if (orow_ != cur_row) {                                                                                           \
      if (cur_row >= 0) {                                                                                             \
        _Pragma("unroll") for (int c = 0; c < C; c++) qv[(cur_slot * C + c) * 64 + lane] = ov[c];                     \
        if (HYB && lane == 0) qver[cur_slot] = cur_ver;                                                               \
      }                                                                                                               \
      const int have_ = __builtin_amdgcn_readfirstlane(qt[slot_]);                                                    \
      if (have_ == orow_) {                                                                                           \
        _Pragma("unroll") for (int c = 0; c < C; c++) ov[c] = qv[(slot_ * C + c) * 64 + lane];                        \
      } else {                                                                                                        \
        if (have_ >= 0) {                                                                                             \
          const uint32_t hv_ = HYB ? (uint32_t)__builtin_amdgcn_readfirstlane((int)qver[slot_]) : 0u;                 \
          OWN_PUT(have_, hv_, [&](int c) { return qv[(slot_ * C + c) * 64 + lane]; })                                 \
        }                                                                                                             \
        OWN_FETCH(orow_, KA)                                                                                          \
        if (lane == 0) qt[slot_] = orow_;                                                                             \
      }                                                                                                               \
      cur_row = orow_;                                                                                                \
      cur_slot = slot_;                                                                                               \
    }                                                                                                                 \
  }
#define FW_REQUEST(K, ROWX, LIVE)                                                                                     \
  {                                                                                                                   \
    const uint32_t rb_ = (LIVE) ? (uint32_t)(ROWX) * (uint32_t)(8 * LD) + g_off : FL_OOB;                             \
    _Pragma("unroll") for (int c = 0; c < C; c++) fw_load2(Tr[K][c], dt, (LIVE) ? rb_ + (uint32_t)(c * 512) : FL_OOB); \
  }
#define FW_STEP(K, S, NWAIT)                                                                                          \
  {                                                                                                                   \
    const int s_ = (S);                                                                                               \
    const bool live_ = s_ < nb;                                                                                       \
    uint32_t stbase_ = FL_OOB, ntag_ = 0;                                                                             \
    float tv[C];                                                                                                      \
    _Pragma("unroll") for (int c = 0; c < C; c++) tv[c] = 0.0f;                                                       \
    if (live_ && !aborted) {                                                                                          \
      const int sl_ = s_ & 63;                                                                                        \
      const int rx_ = __builtin_amdgcn_readlane(rec.x, sl_), ry_ = __builtin_amdgcn_readlane(rec.y, sl_);             \
      const int rz_ = __builtin_amdgcn_readlane(rec.z, sl_);                                                          \
      const uint32_t exp_ = (uint32_t)__builtin_amdgcn_readlane(rec.w, sl_);                                          \
      const uint32_t ka_ = HYB ? (uint32_t)__builtin_amdgcn_readlane((int)qav, sl_) >> 1 : 0u;                        \
      OWN_SWITCH(ry_, ka_)                                                                                            \
      cur_ver = ka_ + 1u;                                                                                             \
      /* the row of the other side: requested LA positions ago */                                                     \
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NWAIT) : "memory");                                                    \
      _Pragma("unroll") for (int c = 0; c < C; c++) asm volatile("" : "+v"(Tr[K][c]));                                \
      bool ok_ = true;                                                                                                \
      _Pragma("unroll") for (int c = 0; c < C; c++) ok_ = ok_ && (uint32_t)(Tr[K][c] >> 32) == exp_;                                  \
      if (__builtin_amdgcn_ballot_w64(ok_) != ~0ull) {       /* not there yet: probe the first granule, then read the row again */ \
        const uint32_t rb_ = (uint32_t)rx_ * (uint32_t)(8 * LD);                                                      \
        long long t_last_ = wall_clock64();                                                                           \
        int spins_ = 0;                                                                                               \
        for (;;) {                                                                                                    \
          unsigned long long pr_ = 0ull;                                                                              \
          fw_load2(pr_, dt, lane == 0 ? rb_ : FL_OOB);                                                                \
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(pr_)::"memory");                                                   \
          if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(pr_ >> 32)) == exp_) {                                   \
            _Pragma("unroll") for (int c = 0; c < C; c++) fw_load2(Tr[K][c], dt, rb_ + g_off + (uint32_t)(c * 512));  \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
            _Pragma("unroll") for (int c = 0; c < C; c++) asm volatile("" : "+v"(Tr[K][c]));                          \
            bool t_ = true;                                                                                           \
            _Pragma("unroll") for (int c = 0; c < C; c++) t_ = t_ && (uint32_t)(Tr[K][c] >> 32) == exp_;                              \
            if (__builtin_amdgcn_ballot_w64(t_) == ~0ull) break;                                                      \
          }                                                                                                           \
          if ((++spins_ & 63) == 0) {                                                                                 \
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { aborted = true; break; }       \
            if (wall_clock64() - t_last_ > 200000000LL) {      /* 100 MHz constant clock: 2 s on one queue head */    \
              __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                               \
              aborted = true;                                                                                         \
              break;                                                                                                  \
            }                                                                                                         \
          }                                                                                                           \
        }                                                                                                             \
      }                                                                                                               \
      if (!aborted) {                                                                                                 \
        FL_STAT(2, 1);                                                                                                \
        _Pragma("unroll") for (int c = 0; c < C; c++) tv[c] = __uint_as_float((uint32_t)Tr[K][c]);                            \
        if (own_user) {                                                                                               \
          const float est_ = wide_dot<C>(ov, tv);                                                                     \
          wide_axpys<C, ARITH>(ov, tv, __int_as_float(rz_), est_, lr, uReg, iReg);                                    \
        } else {                                                                                                      \
          const float est_ = wide_dot<C>(tv, ov);                                                                     \
          wide_axpys<C, ARITH>(tv, ov, __int_as_float(rz_), est_, lr, uReg, iReg);                                    \
        }                                                                                                             \
        stbase_ = (uint32_t)rx_ * (uint32_t)(8 * LD) + g_off;                                                         \
        ntag_ = exp_ + 1u;                                                                                            \
      }                                                                                                               \
    }                                                                                                                 \
    /* the other side's row with its new tag, write-through (always C instructions), then the request for position S + LA */ \
    _Pragma("unroll") for (int c = 0; c < C; c++)                                                                     \
        fw_store2(dt, stbase_ == FL_OOB ? FL_OOB : stbase_ + (uint32_t)(c * 512), uint2v{__float_as_uint(tv[c]), ntag_}); \
    {                                                                                                                 \
      const int sn_ = s_ + LA;                                                                                        \
      const bool ln_ = sn_ < nb && !aborted;                                                                          \
      const int rxn_ = __builtin_amdgcn_readlane(rec.x, sn_ & 63);                                                    \
      FW_REQUEST(K, rxn_, ln_)                                                                                        \
    }                                                                                                                 \
  }

#ifdef MFX_FLOW_STATS
  // per queue: {pole visits, polls, generic visits, pole blocks, -, cycles in the kernel, cycles in polls, cycles in counted waits + block starts}
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#define FLW_T0 const unsigned long long st_w = __builtin_amdgcn_s_memtime();
#define FLW_T1(i) st_[i] += __builtin_amdgcn_s_memtime() - st_w;
#else
#define FLW_T0
#define FLW_T1(i)
#endif
  // ---- pole blocks (see fwp_link above; landing registers a0 .. a31, see fwa_load) ------------------------------------------------
  // the probe-then-poll of FW_STEP for landing slot K, row RX, tags EXP: until the row carries its tags (or the launch is aborted);
  // leaves the row's values in VAL
#define FWP_POLL(K, RX, EXP, VAL)                                                                                     \
  {                                                                                                                   \
    const uint32_t rb_ = (uint32_t)(RX) * (uint32_t)(8 * LD);                                                         \
    long long t_last_ = wall_clock64();                                                                               \
    int spins_ = 0;                                                                                                   \
    for (;;) {                                                                                                        \
      bool hit_ = spins_ < pollfull;           /* the first `pollfull` tries read the whole row: one round trip, not two */ \
      if (!hit_) {                                                                                                    \
        unsigned long long pr_ = 0ull;                                                                                \
        fw_load2(pr_, dt, lane == 0 ? rb_ : FL_OOB);                                                                  \
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pr_)::"memory");                                                     \
        hit_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pr_ >> 32)) == (EXP);                                   \
      }                                                                                                               \
      if (hit_) {                                                                                                     \
        fwa_request<K, C>(dt, rb_ + g_off);                                                                           \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
        uint32_t tg_[C];                                                                                              \
        fwa_take<K, C>(VAL, tg_);                                                                                     \
        bool t_ = true;                                                                                               \
        _Pragma("unroll") for (int c = 0; c < C; c++) t_ = t_ && tg_[c] == (EXP);                                     \
        if (__builtin_amdgcn_ballot_w64(t_) == ~0ull) break;                                                          \
      }                                                                                                               \
      if ((++spins_ & 63) == 0) {                                                                                     \
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { aborted = true; break; }           \
        if (wall_clock64() - t_last_ > 200000000LL) {        /* 100 MHz constant clock: 2 s on one queue head */      \
          __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                                   \
          aborted = true;                                                                                             \
          break;                                                                                                      \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
  }
  // One visit of a pole block.  K = landing slot of position S (verified: its values are in tvc, its record in rx_c / rz_c / exp_c),
  // K1 = slot of position S + 1, NW = vector-memory instructions issued behind the request of position S + 1 (2 C, 3 C, then 4 C).
  // Order: (1) wait for the row of S + 1 to have LANDED, read it out and compare its tags -- no branch yet; (2) this visit; (3) its
  // tagged store; (4) the request of position S + 4 into slot K; (5) only now, if S + 1 did not carry its tags: probe and poll
  // (another queue may be waiting for the store of (3)).
#define FWP_STEP(K, K1, S, NW)                                                                                        \
  {                                                                                                                   \
    const int s1_ = ((S) + 1) & 63;                                                                                   \
    FL_STAT(0, 1);                                                                                                    \
    { FLW_T0 asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NW) : "memory"); FLW_T1(7) }                                    \
    uint32_t vn_[C], tn_[C];                                                                                          \
    fwa_take<K1, C>(vn_, tn_);                                                                                        \
    const int rx_n = __builtin_amdgcn_readlane(rec.x, s1_), rz_n = __builtin_amdgcn_readlane(rec.z, s1_);             \
    const uint32_t exp_n = (uint32_t)__builtin_amdgcn_readlane(rec.w, s1_);                                           \
    bool okn_ = true;                                                                                                 \
    _Pragma("unroll") for (int c = 0; c < C; c++) okn_ = okn_ && tn_[c] == exp_n;                                     \
    const bool needn_ = (S) + 1 < nbp && __builtin_amdgcn_ballot_w64(okn_) != ~0ull;                                  \
    /* (2) the visit: other side's row gathered per quad, the chain, the reference's bracket */                      \
    float g_[C][4];                                                                                                   \
    _Pragma("unroll") for (int c = 0; c < C; c++) {                                                                   \
      g_[c][0] = fw_dpp<0x00>(tvc[c]); g_[c][1] = fw_dpp<0x55>(tvc[c]);                                               \
      g_[c][2] = fw_dpp<0xAA>(tvc[c]); g_[c][3] = fw_dpp<0xFF>(tvc[c]);                                               \
    }                                                                                                                 \
    const float est_ = pole_dot<C>(ov, g_);                                                                           \
    if (OWN_U) pole_axpys<C, ARITH>(ov, tvc, __int_as_float(rz_c), est_, lr, uReg, iReg);                             \
    else pole_axpys<C, ARITH>(tvc, ov, __int_as_float(rz_c), est_, lr, uReg, iReg);                                   \
    /* (3) + (4) */                                                                                                   \
    {                                                                                                                 \
      const uint32_t stb_ = (uint32_t)rx_c * (uint32_t)(8 * LD) + g_off;                                              \
      _Pragma("unroll") for (int c = 0; c < C; c++)                                                                   \
          fw_store2(dt, stb_ + (uint32_t)(c * 512), uint2v{__float_as_uint(tvc[c]), exp_c + 1u});                     \
      const int sn_ = (S) + LAP;                                                                                      \
      const int rxq_ = __builtin_amdgcn_readlane(rec.x, sn_ & 63);                                                    \
      fwa_request<K, C>(dt, sn_ < nbp ? (uint32_t)rxq_ * (uint32_t)(8 * LD) + g_off : FL_OOB);                        \
    }                                                                                                                 \
    /* (5) */                                                                                                         \
    if (__builtin_expect(needn_ && !aborted, 0)) { FL_STAT(1, 1); FLW_T0 FWP_POLL(K1, rx_n, exp_n, vn_) FLW_T1(6) }   \
    _Pragma("unroll") for (int c = 0; c < C; c++) tvc[c] = __uint_as_float(vn_[c]);                                   \
    rx_c = rx_n; rz_c = rz_n; exp_c = exp_n;                                                                          \
  }
  // LAP = rows of the other side in flight in a pole block.  The landing pairs cost nothing but accumulation registers: 16 rows in
  // flight at C = 1, 8 at C = 2, 4 beyond (a0 .. a31; LAP divides the 64 positions of a block).  (Measured at C2, rank 64: 4 or 16 in
  // flight, float or double bracket, probes or whole-row polls -- the epoch takes 14.2 - 14.5 ms all the same, because the launch no longer
  // ends with the most popular item's queue but with the chain of the busiest USER's row: 10 717 visits in 10 717 different queues, each a
  // hand-off through memory at ~ 1.3 us (MI355X_MICROARCH.md, handoff-1to1: 0.8 - 1.0 us on an idle chip).  scripts/flow_model.py
  // reproduces it: makespan = max(49 777 x step, 10 717 x (step + hand-off)).  Ranks 128 / 256, where the step was the larger term:
  // 18.6 -> 16.4 ms, 27.7 -> 24.1 ms.)
  constexpr int LAP = C == 1 ? 16 : C == 2 ? 8 : 4;
  static_assert(LAP * C <= 16 && (LAP - 2) * 2 * C <= 63 && LAP <= 16 && NB % LAP == 0, "pole path: landing pairs and the counted waits");
  // vector-memory instructions issued behind the request of position S + 1 when step S waits for it: the rest of the prologue's
  // requests and 2 C per earlier step -- (LAP - 2 + S) C in the first steps of a block, (LAP - 2) 2 C from step LAP - 2 on
#define FWP_NW(PEEL_, S_) (((PEEL_) && (S_) <= LAP - 2) ? (LAP - 2 + (S_)) * C : (LAP - 2) * 2 * C)
#define FWP_ROUND(S0_, PEEL_)                                                                                         \
  {                                                                                                                   \
    if constexpr (LAP > 0) { FWP_STEP(0, (1 % LAP), (S0_) + 0, FWP_NW(PEEL_, 0)) }   \
    if constexpr (LAP > 1) { FWP_STEP(1, (2 % LAP), (S0_) + 1, FWP_NW(PEEL_, 1)) }   \
    if constexpr (LAP > 2) { FWP_STEP(2, (3 % LAP), (S0_) + 2, FWP_NW(PEEL_, 2)) }   \
    if constexpr (LAP > 3) { FWP_STEP(3, (4 % LAP), (S0_) + 3, FWP_NW(PEEL_, 3)) }   \
    if constexpr (LAP > 4) { FWP_STEP(4, (5 % LAP), (S0_) + 4, FWP_NW(PEEL_, 4)) }   \
    if constexpr (LAP > 5) { FWP_STEP(5, (6 % LAP), (S0_) + 5, FWP_NW(PEEL_, 5)) }   \
    if constexpr (LAP > 6) { FWP_STEP(6, (7 % LAP), (S0_) + 6, FWP_NW(PEEL_, 6)) }   \
    if constexpr (LAP > 7) { FWP_STEP(7, (8 % LAP), (S0_) + 7, FWP_NW(PEEL_, 7)) }   \
    if constexpr (LAP > 8) { FWP_STEP(8, (9 % LAP), (S0_) + 8, FWP_NW(PEEL_, 8)) }   \
    if constexpr (LAP > 9) { FWP_STEP(9, (10 % LAP), (S0_) + 9, FWP_NW(PEEL_, 9)) }   \
    if constexpr (LAP > 10) { FWP_STEP(10, (11 % LAP), (S0_) + 10, FWP_NW(PEEL_, 10)) }   \
    if constexpr (LAP > 11) { FWP_STEP(11, (12 % LAP), (S0_) + 11, FWP_NW(PEEL_, 11)) }   \
    if constexpr (LAP > 12) { FWP_STEP(12, (13 % LAP), (S0_) + 12, FWP_NW(PEEL_, 12)) }   \
    if constexpr (LAP > 13) { FWP_STEP(13, (14 % LAP), (S0_) + 13, FWP_NW(PEEL_, 13)) }   \
    if constexpr (LAP > 14) { FWP_STEP(14, (15 % LAP), (S0_) + 14, FWP_NW(PEEL_, 14)) }   \
    if constexpr (LAP > 15) { FWP_STEP(15, (16 % LAP), (S0_) + 15, FWP_NW(PEEL_, 15)) }   \
  }
#define FWP_BLOCK(OWN_U_)                                                                                             \
  {                                                                                                                   \
    constexpr bool OWN_U = OWN_U_;                                                                                    \
    float tvc[C];                                                                                                     \
    int rx_c = __builtin_amdgcn_readlane(rec.x, 0), rz_c = __builtin_amdgcn_readlane(rec.z, 0);                       \
    uint32_t exp_c = (uint32_t)__builtin_amdgcn_readlane(rec.w, 0);                                                   \
    /* the requests of positions 0 .. LAP - 1; position 0: behind its request lie the LAP - 1 others */              \
    FL_STAT(3, 1);                                                                                                    \
    fwp_prologue<0, LAP, C>(dt, rec.x, (uint32_t)(8 * LD), g_off);                                                    \
    { FLW_T0 asm volatile("s_waitcnt vmcnt(%0)" ::"i"((LAP - 1) * C) : "memory"); FLW_T1(7) }                         \
    {                                                                                                                 \
      uint32_t v0_[C], t0_[C];                                                                                        \
      fwa_take<0, C>(v0_, t0_);                                                                                       \
      bool ok0_ = true;                                                                                               \
      _Pragma("unroll") for (int c = 0; c < C; c++) ok0_ = ok0_ && t0_[c] == exp_c;                                   \
      if (__builtin_amdgcn_ballot_w64(ok0_) != ~0ull) { FL_STAT(1, 1); FLW_T0 FWP_POLL(0, rx_c, exp_c, v0_) FLW_T1(6) }   \
      _Pragma("unroll") for (int c = 0; c < C; c++) tvc[c] = __uint_as_float(v0_[c]);                                 \
    }                                                                                                                 \
    if (!aborted) {                                                                                                   \
      FWP_ROUND(0, true)                                                                                              \
      _Pragma("clang loop unroll(disable)")                                                                           \
      for (int s0 = LAP; s0 < nbp && !aborted; s0 += LAP) FWP_ROUND(s0, false)                                        \
    }                                                                                                                 \
    { FLW_T0 asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FLW_T1(7) }   /* nothing in flight into a0 .. a31 when the block is left */ \
  }

  while (pos < end && !aborted) {
    int nb = (int)min((uint32_t)NB, end - pos);
    uint4v rb = fl_load<false>(dq, lane < nb ? (pos + (uint32_t)lane) * 16u : FL_OOB);
    uint32_t qav = 0u;                                   // hybrid: rank of the visit in its owned row's chain << 1 | pub
    if constexpr (HYB) qav = fw_load1(dqa, lane < nb ? (pos + (uint32_t)lane) * 4u : FL_OOB);
    { FLW_T0 asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb), "+v"(qav)::"memory"); FLW_T1(4) }
    const int4 rec = make_int4((int)rb.x, (int)rb.y, (int)rb.z, (int)rb.w);
    bool pub_end = false;
    if constexpr (HYB) {                                 // the block ends behind the first record that publishes its row
      const unsigned long long pm = __builtin_amdgcn_ballot_w64(lane < nb && (qav & 1u));
      if (pm) { nb = min(nb, (int)__builtin_ctzll(pm) + 1); pub_end = true; }
    }
    // a pole block: records that all visit ONE owned row, as many of them as whole rounds of LAP steps (the rest of a block that a
    // publication or the queue's end cut short follows as a block of its own through the generic steps); nothing of it touches the
    // landing registers Tr of the generic steps
    const int ry0 = __builtin_amdgcn_readfirstlane(rec.y);
    const int nbp = nb / LAP * LAP;
    if ((own_flags & 4) && nbp >= LAP && __builtin_amdgcn_ballot_w64(lane >= nbp || rec.y == ry0) == ~0ull) {
      const uint32_t ka0 = HYB ? (uint32_t)__builtin_amdgcn_readfirstlane((int)qav) >> 1 : 0u;
      OWN_SWITCH(ry0, ka0)          // the owned-row switch of FW_STEP, once for the block
      if (own_user) FWP_BLOCK(true) else FWP_BLOCK(false)
      cur_ver = ka0 + (uint32_t)nbp;
      if (HYB && pub_end && nbp == nb && !aborted) {     // the row's next visit lies in another queue: written back with its version, dropped from the cache
        OWN_PUT(cur_row, cur_ver, [&](int c) { return ov[c]; })
        if (lane == 0) qt[cur_slot] = -1;
        cur_row = -1;
      }
      pos += (uint32_t)nbp;
      continue;
    }
#pragma unroll
    for (int k = 0; k < LA; k++) {
      const int rxn = __builtin_amdgcn_readlane(rec.x, k);
      FW_REQUEST(k, rxn, k < nb)
    }
    // the first LA positions: behind the request of position k lie the later requests of the prologue and k full steps
    FW_STEP(0, 0, 3 * C + 0 * P::NSTEP)
    FW_STEP(1, 1, 2 * C + 1 * P::NSTEP)
    FW_STEP(2, 2, 1 * C + 2 * P::NSTEP)
    FW_STEP(3, 3, 0 * C + 3 * P::NSTEP)
    for (int s0 = LA; s0 < nb; s0 += LA) {
      FW_STEP(0, s0, 3 * P::NSTEP)
      FW_STEP(1, s0 + 1, 3 * P::NSTEP)
      FW_STEP(2, s0 + 2, 3 * P::NSTEP)
      FW_STEP(3, s0 + 3, 3 * P::NSTEP)
    }
    // nothing may be in flight when the landing registers leave the pipeline (see the kernel above)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < LA; k++)
#pragma unroll
      for (int c = 0; c < C; c++) asm volatile("" : "+v"(Tr[k][c]));
    if (HYB && pub_end && !aborted && cur_row >= 0) {     // (the block's last record publishes the row it visited: the one in registers)
      OWN_PUT(cur_row, cur_ver, [&](int c) { return ov[c]; })
      if (lane == 0) qt[cur_slot] = -1;
      cur_row = -1;
    }
    pos += (uint32_t)nb;
  }
#undef FW_STEP
#undef FW_REQUEST
#undef FWP_STEP
#undef FWP_BLOCK
#undef FWP_POLL
#undef FWP_ROUND
#undef FWP_NW
#ifdef MFX_FLOW_STATS
  st_[5] = __builtin_amdgcn_s_memtime() - st_t0;
  if (lane == 0)
    for (int i = 0; i < 8; i++) fl_stats[grp * 8 + i] = st_[i];
#endif
  // the owned rows go back to their table: the held one through its LDS slot, then every slot in use
  if (cur_row >= 0) {
#pragma unroll
    for (int c = 0; c < C; c++) qv[(cur_slot * C + c) * 64 + lane] = ov[c];
    if (HYB && lane == 0) { qver[cur_slot] = cur_ver; qt[cur_slot] = cur_row; }
  }
  for (int sI = 0; sI < QR; sI++) {
    const int have = __builtin_amdgcn_readfirstlane(qt[sI]);
    if (have >= 0) {
      if constexpr (HYB) {
        const uint32_t hv = (uint32_t)__builtin_amdgcn_readfirstlane((int)qver[sI]);
        OWN_PUT(have, hv, [&](int c) { return qv[(sI * C + c) * 64 + lane]; })
      } else {
#pragma unroll
        for (int c = 0; c < C; c++) O[(int64_t)have * LD + c * 64 + lane] = qv[(sI * C + c) * 64 + lane];
      }
    }
  }
#undef OWN_PUT
#undef OWN_FETCH
#undef OWN_SWITCH
}

// granule copy of a factor table and back: row r, chunk c, lane j, half h (elements 4j+2h, 4j+2h+1 of the chunk) at byte
// r*8*LD + c*32*L + h*16*L + j*16 as {value, tag, value, tag}
__global__ void flow_tag_kernel(const float* __restrict__ X, int64_t n, int L, int C, float* __restrict__ T) {
  const int64_t per = (int64_t)L * C, total = n * per, stride = (int64_t)gridDim.x * blockDim.x;
  const int LD = 4 * L * C;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t r = t / per;
    const int c = (int)((t % per) / L), j = (int)(t % L);
    const float4v v = *(const float4v*)(X + r * LD + c * 4 * L + 4 * j);
    uint4v* d = (uint4v*)(T + r * 2 * LD) + c * 2 * L + j;
    d[0] = uint4v{__float_as_uint(v.x), 0u, __float_as_uint(v.y), 0u};
    d[L] = uint4v{__float_as_uint(v.z), 0u, __float_as_uint(v.w), 0u};
  }
}
__global__ void flow_untag_kernel(const float* __restrict__ T, int64_t n, int L, int C, float* __restrict__ X) {
  const int64_t per = (int64_t)L * C, total = n * per, stride = (int64_t)gridDim.x * blockDim.x;
  const int LD = 4 * L * C;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t r = t / per;
    const int c = (int)((t % per) / L), j = (int)(t % L);
    const uint4v* d = (const uint4v*)(T + r * 2 * LD) + c * 2 * L + j;
    const uint4v a = d[0], b = d[L];
    *(float4v*)(X + r * LD + c * 4 * L + 4 * j) = float4v{__uint_as_float(a.x), __uint_as_float(a.z), __uint_as_float(b.x), __uint_as_float(b.z)};
  }
}

int build_flow(mfx_ctx* ctx, int64_t first, int64_t count, int64_t groups, bool tagged) {
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
  const int32_t maxU = degU.empty() ? 0 : *std::max_element(degU.begin(), degU.end()), maxI = degI.empty() ? 0 : *std::max_element(degI.begin(), degI.end());
  // owned side: as the device builder chooses it (user rows for a list that keeps a user's ratings together, else the longest chain)
  int own_user = maxU > maxI ? 1 : 0;
  if (const char* e = getenv("MFX_FLOW_OWN")) {
    if (e[0] == 'u') own_user = 1; else if (e[0] == 'i') own_user = 0;
  } else if (count > 1) {
    int64_t au = 0, ai = 0;
    for (int64_t t = 0; t + 1 < count; t++) { au += S->hu[(size_t)t] == S->hu[(size_t)t + 1]; ai += S->hi[(size_t)t] == S->hi[(size_t)t + 1]; }
    if (2 * au > count) own_user = 1; else if (2 * ai > count) own_user = 0;
  }
  const std::vector<int32_t>& degOwn = own_user ? degU : degI;
  const int32_t* hown = own_user ? S->hu.data() : S->hi.data();
  const int32_t* hoth = own_user ? S->hi.data() : S->hu.data();
  const int64_t nOwn = own_user ? ctx->nU : ctx->nI, nOth = own_user ? ctx->nI : ctx->nU;
  // owned rows -> groups: longest chain first onto the least loaded group
  S->assign_gen = ~0ull;            // (the device builder's cached assignment lives in the same arrays)
  S->owner.assign((size_t)nOwn, 0);
  {
    std::vector<int32_t> rows;
    for (int64_t r = 0; r < nOwn; r++)
      if (degOwn[(size_t)r] > 0) rows.push_back((int32_t)r);
    std::sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) { return degOwn[a] != degOwn[b] ? degOwn[a] > degOwn[b] : a < b; });
    typedef std::pair<int64_t, int32_t> Load;   // (ratings so far, group)
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int64_t gI = 0; gI < groups; gI++) heap.push(Load(0, (int32_t)gI));
    std::vector<int32_t> nrows((size_t)groups, 0);
    S->oslot.assign((size_t)nOwn, 0);
    for (int32_t r : rows) {
      Load l = heap.top();
      heap.pop();
      S->owner[(size_t)r] = l.second;
      S->oslot[(size_t)r] = (uint8_t)(nrows[(size_t)l.second]++ & 63);   // the row's place in its queue's LDS cache (tagged schedule)
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
  if (tagged) {       // the LDS slots of the owned rows, packed as the device builder packs them
    if (S->own_cap < std::max<int64_t>(ctx->nU, ctx->nI)) {
      dev_free(S->downer); dev_free(S->dstart);
      S->own_cap = 0;
      if ((rc = dev_alloc(ctx, &S->downer, (size_t)std::max(ctx->nU, ctx->nI))) || (rc = dev_alloc(ctx, &S->dstart, (size_t)std::max(ctx->nU, ctx->nI) + 1)))
        return rc;
      S->own_cap = std::max(ctx->nU, ctx->nI);
    }
    S->hpacked.resize((size_t)nOwn);
    for (int64_t r = 0; r < nOwn; r++) S->hpacked[(size_t)r] = (int32_t)((uint32_t)S->owner[(size_t)r] | (uint32_t)S->oslot[(size_t)r] << FL_SLOT_SHIFT);
    HIPCHK(hipMemcpyAsync(S->downer, S->hpacked.data(), sizeof(int32_t) * (size_t)nOwn, hipMemcpyHostToDevice, ctx->stream));
  }
  const int blocks = (int)std::min<int64_t>((count + 255) / 256, 8192);
  hipLaunchKernelGGL(flow_gather_kernel, dim3(blocks), dim3(256), 0, ctx->stream, S->lpos, S->vexp, count, ctx->eu + first,
                     ctx->ei + first, ctx->er + first, own_user, tagged ? (const int32_t*)S->downer : (const int32_t*)nullptr, S->q);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));    // the staging vectors are reused by the next call
  S->groups = groups; S->longest = longest; S->own_user = own_user; S->tagged = tagged;
  S->prep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (getenv("MFX_DEBUG"))
    fprintf(stderr, "[mfx] dataflow replay: %lld ratings on %lld groups (%s rows owned), longest queue %lld, longest chains %d users / %d items, "
            "host preparation %.1f ms\n", (long long)count, (long long)groups, own_user ? "user" : "item", (long long)longest, maxU, maxI,
            S->prep_ms);
  return MFX_OK;
}

// ---- hybrid ownership, host statement (sgd_flow_wide_kernel<.., HYB>).  The H busiest users (at most `heavy_max`, each with at least
// `heavy_min` ratings in the list) get a queue of their own holding ALL their ratings; every other rating is visited in its item's
// queue as before.  Per record: .x the other side's row, .y the owned row | LDS slot << 26, .z the rating, .w the rank of the visit
// in the OTHER row's chain (its expected version), qa = rank in the OWNED row's chain << 1 | pub, pub = "the owned row's next visit
// lies in another queue (or there is none)": the kernel writes the row back behind that visit and ends its block there.
// Returns 1 (nothing built) for a list that keeps a user's ratings together or has no user heavy enough: the caller builds the
// ordinary queues.
int build_flow_hybrid_host(mfx_ctx* ctx, int64_t first, int64_t count, int64_t groups, int heavy_max, int heavy_min) {
  FlowState* S = fl(ctx);
  const auto t0 = std::chrono::steady_clock::now();
  int rc;
  const int64_t nU = ctx->nU, nI = ctx->nI;
  std::vector<int32_t> hu((size_t)count), hi((size_t)count);
  std::vector<float> hr((size_t)count);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(hu.data(), ctx->eu + first, sizeof(int32_t) * (size_t)count, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hi.data(), ctx->ei + first, sizeof(int32_t) * (size_t)count, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hr.data(), ctx->er + first, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost));
  std::vector<int32_t> degU((size_t)nU, 0), degI((size_t)nI, 0);
  int64_t au = 0;
  for (int64_t t = 0; t < count; t++) {
    degU[(size_t)hu[(size_t)t]]++; degI[(size_t)hi[(size_t)t]]++;
    if (t + 1 < count) au += hu[(size_t)t] == hu[(size_t)t + 1];
  }
  if (2 * au > count) return 1;                              // a user-ordered list: user rows are owned there
  // the heavy users: busiest first
  std::vector<int32_t> heavy;
  for (int64_t u = 0; u < nU; u++)
    if (degU[(size_t)u] >= heavy_min) heavy.push_back((int32_t)u);
  std::sort(heavy.begin(), heavy.end(), [&](int32_t a, int32_t b) { return degU[a] != degU[b] ? degU[a] > degU[b] : a < b; });
  if ((int64_t)heavy.size() > heavy_max) heavy.resize((size_t)heavy_max);
  const int H = (int)std::min<int64_t>((int64_t)heavy.size(), groups / 2);
  if (H == 0) return 1;
  heavy.resize((size_t)H);
  const int64_t g_item = groups - H;
  std::vector<int32_t> uq((size_t)nU, -1);
  for (int h = 0; h < H; h++) uq[(size_t)heavy[(size_t)h]] = h;
  // items -> item queues by the ratings that stay with them, longest first onto the lightest queue; LDS slots as in build_flow
  std::vector<int32_t> degL((size_t)nI, 0);
  for (int64_t t = 0; t < count; t++)
    if (uq[(size_t)hu[(size_t)t]] < 0) degL[(size_t)hi[(size_t)t]]++;
  std::vector<int32_t> owner((size_t)nI, 0);
  std::vector<uint8_t> oslot((size_t)nI, 0);
  {
    std::vector<int32_t> rows;
    for (int64_t r = 0; r < nI; r++)
      if (degI[(size_t)r] > 0) rows.push_back((int32_t)r);
    std::sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) { return degL[a] != degL[b] ? degL[a] > degL[b] : a < b; });
    typedef std::pair<int64_t, int32_t> Load;
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int64_t gI = 0; gI < g_item; gI++) heap.push(Load(0, (int32_t)gI));
    std::vector<int32_t> nrows((size_t)g_item, 0);
    for (int32_t r : rows) {
      Load l = heap.top();
      heap.pop();
      owner[(size_t)r] = l.second;
      oslot[(size_t)r] = (uint8_t)(nrows[(size_t)l.second]++ & 31);
      l.first += degL[(size_t)r];
      heap.push(l);
    }
  }
  // queue of every rating, queue offsets, ranks in both chains, the "next visit of the item lies elsewhere" flags
  std::vector<int64_t> off((size_t)groups + 1, 0);
  std::vector<int32_t> qof((size_t)count);
  for (int64_t t = 0; t < count; t++) {
    const int32_t h = uq[(size_t)hu[(size_t)t]];
    qof[(size_t)t] = h >= 0 ? (int32_t)(g_item + h) : owner[(size_t)hi[(size_t)t]];
    off[(size_t)qof[(size_t)t] + 1]++;
  }
  int64_t longest = 0;
  for (int64_t gI = 0; gI < groups; gI++) { longest = std::max(longest, off[(size_t)gI + 1]); off[(size_t)gI + 1] += off[(size_t)gI]; }
  std::vector<int4> rec((size_t)count);
  std::vector<uint32_t> qa((size_t)count);
  {
    std::vector<int64_t> cur(off.begin(), off.end() - 1);
    std::vector<int32_t> cu((size_t)nU, 0), ci((size_t)nI, 0);
    std::vector<int64_t> lastOfItem((size_t)nI, -1);        // queue slot of the item's previous visit when that was in its own queue
    for (int64_t t = 0; t < count; t++) {
      const int32_t u = hu[(size_t)t], it = hi[(size_t)t];
      const bool hv = uq[(size_t)u] >= 0;
      const int64_t slot = cur[(size_t)qof[(size_t)t]]++;
      const uint32_t ru = (uint32_t)cu[(size_t)u]++, ri = (uint32_t)ci[(size_t)it]++;
      int rbits;
      memcpy(&rbits, &hr[(size_t)t], 4);
      if (hv) {
        // the heavy user's queue: user row owned (slot 0), item row through its tags; published behind the user's last rating
        rec[(size_t)slot] = make_int4(it, u, rbits, (int)ri);
        qa[(size_t)slot] = ru << 1 | (ru + 1 == (uint32_t)degU[(size_t)u] ? 1u : 0u);
        // the item's previous visit, if it was in the item's own queue, must publish: this visit reads the row from the table
        if (lastOfItem[(size_t)it] >= 0) { qa[(size_t)lastOfItem[(size_t)it]] |= 1u; lastOfItem[(size_t)it] = -1; }
      } else {
        rec[(size_t)slot] = make_int4(u, (int)((uint32_t)it | (uint32_t)oslot[(size_t)it] << FL_SLOT_SHIFT), rbits, (int)ru);
        qa[(size_t)slot] = ri << 1 | (ri + 1 == (uint32_t)degI[(size_t)it] ? 1u : 0u);
        lastOfItem[(size_t)it] = slot;
      }
    }
  }
  // device copies
  if (S->cap < count) {
    dev_free(S->q); dev_free(S->lpos); dev_free(S->vexp);
    S->cap = 0;
    if ((rc = dev_alloc(ctx, &S->q, (size_t)count)) || (rc = dev_alloc(ctx, &S->lpos, (size_t)count)) || (rc = dev_alloc(ctx, &S->vexp, (size_t)count)))
      return rc;
    S->cap = count;
  }
  if (S->qa_cap < count) {
    dev_free(S->qa);
    S->qa_cap = 0;
    if ((rc = dev_alloc(ctx, &S->qa, (size_t)count))) return rc;
    S->qa_cap = count;
  }
  if (S->goff_cap < groups + 1) {
    dev_free(S->qoff);
    S->goff_cap = 0;
    if ((rc = dev_alloc(ctx, &S->qoff, (size_t)groups + 1))) return rc;
    S->goff_cap = groups + 1;
  }
  if (!S->flag && (rc = dev_alloc(ctx, &S->flag, (size_t)1))) return rc;
  HIPCHK(hipMemsetAsync(S->flag, 0, sizeof(unsigned), ctx->stream));
  HIPCHK(hipMemcpyAsync(S->qoff, off.data(), sizeof(int64_t) * ((size_t)groups + 1), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(S->q, rec.data(), sizeof(int4) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(S->qa, qa.data(), sizeof(uint32_t) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  S->hoff.assign(off.begin(), off.end());
  S->assign_gen = ~0ull;
  S->groups = groups; S->longest = longest; S->own_user = 0; S->tagged = true;
  S->hybrid = true; S->g_item = (int)g_item; S->n_heavy = H;
  S->prep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (getenv("MFX_DEBUG"))
    fprintf(stderr, "[mfx] dataflow replay, hybrid ownership: %lld ratings, %lld item queues + %d heavy users' queues (busiest %d, lightest %d ratings), "
            "longest queue %lld, host preparation %.1f ms\n", (long long)count, (long long)g_item, H, degU[(size_t)heavy[0]], degU[(size_t)heavy[(size_t)H - 1]],
            (long long)longest, S->prep_ms);
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
// how often consecutive list positions share their user / their item: out[0], out[1]
__global__ void flow_adjacent_kernel(const int32_t* __restrict__ eu, const int32_t* __restrict__ ei, int64_t n, unsigned long long* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned au = 0, ai = 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t + 1 < n; t += stride) {
    au += eu[t] == eu[t + 1];
    ai += ei[t] == ei[t + 1];
  }
  for (int o = 32; o > 0; o >>= 1) { au += __shfl_down(au, o, 64); ai += __shfl_down(ai, o, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(out, (unsigned long long)au); atomicAdd(out + 1, (unsigned long long)ai); }
}
// owned rows of a list that keeps a row's visits together: first list position of every row ...
__global__ void flow_first_kernel(const int32_t* __restrict__ rows, int64_t n, uint32_t* __restrict__ first) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
    if (t == 0 || rows[t] != rows[t - 1]) atomicMin(first + rows[t], (uint32_t)t);
}
__global__ void flow_iota_kernel(uint32_t* __restrict__ v, int64_t n) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) v[r] = (uint32_t)r;
}
// ... and, the rows sorted by it, queue = rank mod groups (the rows that are at work at the same time sit in different queues),
// LDS slot = its place in that queue mod 64; rows the list does not visit (first = ~0, at the end of the order) keep queue 0
__global__ void flow_deal_kernel(const uint32_t* __restrict__ first_sorted, const uint32_t* __restrict__ row_sorted, int64_t n, int64_t groups,
                                 int32_t* __restrict__ owner) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    const bool used = first_sorted[k] != 0xffffffffu;
    owner[row_sorted[k]] = used ? (int32_t)((uint32_t)(k % groups) | (uint32_t)((k / groups) & 63) << FL_SLOT_SHIFT) : 0;
  }
}
// key = row of `side` for every list position (or the queue of its owned row), value = the position
__global__ void flow_keys_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ owner, int64_t n, uint32_t* __restrict__ key,
                                 uint32_t* __restrict__ val) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    key[t] = owner ? (uint32_t)owner[rows[t]] & (uint32_t)FL_ROW_MASK : (uint32_t)rows[t];
    val[t] = (uint32_t)t;
  }
}
// the queue record of every list position, in LIST order (sequential reads and writes): {other-side row, owned row | LDS slot << 26,
// rating bits, 0}; owner != NULL (tagged schedule): the LDS slot of the owned row rides in the top six bits of .y
__global__ void flow_pack_kernel(const int32_t* __restrict__ eu, const int32_t* __restrict__ ei, const float* __restrict__ er, int64_t n, int own_user,
                                 const int32_t* __restrict__ owner, int4* __restrict__ pack) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    const int u = eu[t], i = ei[t];
    const uint32_t slot = owner ? (uint32_t)owner[own_user ? u : i] >> FL_SLOT_SHIFT : 0u;
    pack[t] = make_int4(own_user ? i : u, (int)((uint32_t)(own_user ? u : i) | slot << FL_SLOT_SHIFT), __float_as_int(er[t]), 0);
  }
}
// sorted by other-side row (stable: list order inside a row): rank of position v1[k] in its row's chain = k - start[row], written
// into the record of that position (.w: the version the visit expects)
__global__ void flow_rank_kernel(const uint32_t* __restrict__ k1, const uint32_t* __restrict__ v1, const int64_t* __restrict__ start, int64_t n,
                                 int4* __restrict__ pack) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) pack[v1[k]].w = (int)(uint32_t)(k - start[k1[k]]);
}
// queue order: ONE 16-byte record per position (round 3 first gathered user, item, rating and rank from four arrays: 1.4 ms of the
// 4 ms a C2 call spends building its queues)
__global__ void flow_gather2_kernel(const uint32_t* __restrict__ lpos, const int4* __restrict__ pack, int64_t n, int4* __restrict__ q) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) q[t] = pack[lpos[t]];
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
// start[r] = number of sorted keys < r, r = 0 .. nr - 1 (the chain starts of the rows in the sorted order)
__global__ void flow_bounds32_kernel(const uint32_t* __restrict__ keys, int64_t n, int64_t nr, int64_t* __restrict__ start) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nr; r += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)keys[mid] < r) lo = mid + 1; else hi = mid;
    }
    start[r] = lo;
  }
}
static int bits_for(uint64_t n) {
  int b = 1;
  while (b < 32 && ((uint64_t)1 << b) < n) b++;
  return b;
}

// the buffers of the device-side constructions (build_flow_device, build_flow_hybrid_device)
static int flow_device_buffers(mfx_ctx* ctx, int64_t count, int64_t groups) {
  FlowState* S = fl(ctx);
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
    dev_free(S->k0); dev_free(S->k1); dev_free(S->v0); dev_free(S->v1); dev_free(S->pack); dev_free(S->sort_tmp);
    S->dcap = 0;
    if ((rc = dev_alloc(ctx, &S->k0, (size_t)count)) || (rc = dev_alloc(ctx, &S->k1, (size_t)count)) || (rc = dev_alloc(ctx, &S->v0, (size_t)count)) ||
        (rc = dev_alloc(ctx, &S->v1, (size_t)count)) || (rc = dev_alloc(ctx, &S->pack, (size_t)count)))
      return rc;
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, S->k0, S->k1, S->v0, S->v1, (size_t)count, 0, 32, ctx->stream));
    if ((rc = dev_alloc(ctx, &S->sort_tmp, bytes))) return rc;
    S->sort_tmp_bytes = bytes;
    S->dcap = count;
  }
  // (the two capacities are separate: a context reused with a differently skewed model can grow max(nU, nI) while nU + nI shrinks)
  if (S->deg_cap < nU + nI) {
    dev_free(S->degU);
    S->deg_cap = 0;
    if ((rc = dev_alloc(ctx, &S->degU, (size_t)(nU + nI)))) return rc;
    S->deg_cap = nU + nI;
  }
  if (S->own_cap < std::max(nU, nI)) {
    dev_free(S->downer); dev_free(S->dstart);
    S->own_cap = 0; S->assign_gen = ~0ull;
    if ((rc = dev_alloc(ctx, &S->downer, (size_t)std::max(nU, nI))) || (rc = dev_alloc(ctx, &S->dstart, (size_t)std::max(nU, nI) + 1))) return rc;
    S->own_cap = std::max(nU, nI);
  }
  S->degI = S->degU + nU;
  if (S->goff_cap < groups + 1) {
    dev_free(S->qoff);
    S->goff_cap = 0;
    if ((rc = dev_alloc(ctx, &S->qoff, (size_t)groups + 1))) return rc;
    S->goff_cap = groups + 1;
  }
  if (!S->flag && (rc = dev_alloc(ctx, &S->flag, (size_t)1))) return rc;
  return MFX_OK;
}

int build_flow_device(mfx_ctx* ctx, int64_t first, int64_t count, int64_t groups, bool tagged) {
  FlowState* S = fl(ctx);
  const auto t0 = std::chrono::steady_clock::now();
  int rc;
  const int64_t nU = ctx->nU, nI = ctx->nI;
  if ((rc = flow_device_buffers(ctx, count, groups))) return rc;
  const int32_t *eu = ctx->eu + first, *ei = ctx->ei + first;
  const float* er = ctx->er + first;
  const int grid = (int)std::min<int64_t>((count + 255) / 256, 8192);
  // Owned rows -> queues (longest chain first onto the least loaded queue) and their LDS slots: ONCE per train matrix and queue
  // count, from the chain lengths of the first list replayed over it (round 2 recomputed them every epoch: 2.1 ms of global atomics
  // at C2, a copy back, a host sort and a copy forth -- a third of the queue construction).  The lists replayed over one matrix are
  // permutations of its ratings (std::shuffle orders, shuffled users, the stratified rounds); for any other list (a sub-range)
  // the cached assignment is merely less balanced -- ownership, not balance, is what correctness rests on.
  // Which side is owned.  A list that keeps the ratings of one user together (trainUShuffle's user-ordered epochs, the blocks of
  // trainSGDPar's rounds) is a chain of back-to-back visits of that user's row: owned by ONE queue they stay in registers; with the
  // item rows owned every one of them is a hand-off through memory between two queues, and the queues advance like one sequential
  // sweep (measured: 5.8 s per C2 epoch of trainUShuffle, 3.4 M updates/s).  So: user rows when more than half of the consecutive
  // list positions share their user; else the side with the longest chain (items, on rating data); MFX_FLOW_OWN=user|item forces.
  int force_own = -1;
  if (const char* e = getenv("MFX_FLOW_OWN")) force_own = e[0] == 'u' ? 1 : (e[0] == 'i' ? 0 : -1);
  int adj_own = -1;
  if (force_own < 0 && count > 1) {
    unsigned long long* adj = (unsigned long long*)S->dstart;        // (scratch: dstart is written further down)
    unsigned long long hadj[2] = {0, 0};
    HIPCHK(hipMemsetAsync(adj, 0, sizeof hadj, ctx->stream));
    hipLaunchKernelGGL(flow_adjacent_kernel, dim3(std::min(grid, 1024)), dim3(256), 0, ctx->stream, eu, ei, count, adj);
    HIPCHK(hipMemcpyAsync(hadj, adj, sizeof hadj, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (2 * hadj[0] > (unsigned long long)count) adj_own = 1;
    else if (2 * hadj[1] > (unsigned long long)count) adj_own = 0;
  }
  const int want_own = force_own >= 0 ? force_own : adj_own;          // -1: by the longest chain
  // A list that keeps a row's visits together (adj_own >= 0) is worked off like a wave front: the rows at work at any time are
  // CONSECUTIVE in the order of their first visit (an item's chain passes from one of them to the next).  Dealt round-robin in that
  // order they sit in different queues; dealt by chain length (below) a queue's next row is often thousands of rows ahead of the
  // front and its wave waits.  Per call, on the device (the order changes with every epoch); MFX_FLOW_DEAL=0 keeps the dealing by
  // chain length.
  const char* dealEnv = getenv("MFX_FLOW_DEAL");
  const int64_t nOwnD = adj_own == 1 ? nU : nI;
  const bool deal = adj_own >= 0 && want_own == adj_own && !(dealEnv && dealEnv[0] == '0') && nOwnD <= count;
  if (deal) {
    size_t need = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, need, S->k0, S->k1, S->v0, S->v1, (size_t)nOwnD, 0, 32, ctx->stream));
    NEED(need <= S->sort_tmp_bytes, MFX_E_STATE, "sgd dataflow: sort workspace for %lld rows exceeds the one for %lld ratings", (long long)nOwnD,
         (long long)count);
    const int rgrid = (int)std::min<int64_t>((nOwnD + 255) / 256, 4096);
    HIPCHK(hipMemsetAsync(S->k0, 0xff, sizeof(uint32_t) * (size_t)nOwnD, ctx->stream));
    hipLaunchKernelGGL(flow_first_kernel, dim3(grid), dim3(256), 0, ctx->stream, adj_own ? eu : ei, count, S->k0);
    hipLaunchKernelGGL(flow_iota_kernel, dim3(rgrid), dim3(256), 0, ctx->stream, S->v0, nOwnD);
    size_t bytes0 = S->sort_tmp_bytes;
    HIPCHK(rocprim::radix_sort_pairs(S->sort_tmp, bytes0, S->k0, S->k1, S->v0, S->v1, (size_t)nOwnD, 0, 32, ctx->stream));
    hipLaunchKernelGGL(flow_deal_kernel, dim3(rgrid), dim3(256), 0, ctx->stream, S->k1, S->v1, nOwnD, groups, S->downer);
    HIPCHK(hipGetLastError());
    S->assign_gen = ~0ull;                 // the cached dealing by chain length is gone from downer
    S->hy_gen = ~0ull;
    S->assign_own_user = adj_own;
    S->assign_maxU = S->assign_maxI = 0;   // (not looked at on this path)
  } else if (S->assign_gen != ctx->train_gen || S->assign_groups != groups || S->assign_nU != nU || S->assign_nI != nI || S->assign_want != want_own ||
             S->assign_count != count) {
    // (keyed on the list length too: an assignment computed from a sub-range or from one round's list of trainSGDPar knows the chain
    //  lengths of THAT list only; a later whole epoch gets its own.  Round-3 advice.)
    HIPCHK(hipMemsetAsync(S->degU, 0, sizeof(int32_t) * (size_t)(nU + nI), ctx->stream));
    hipLaunchKernelGGL(flow_degrees_kernel, dim3(grid), dim3(256), 0, ctx->stream, eu, ei, count, S->degU, S->degI);
    S->hdeg.resize((size_t)(nU + nI));
    HIPCHK(hipMemcpyAsync(S->hdeg.data(), S->degU, sizeof(int32_t) * (size_t)(nU + nI), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int32_t *hdU = S->hdeg.data(), *hdI = S->hdeg.data() + nU;
    S->assign_maxU = nU > 0 ? *std::max_element(hdU, hdU + nU) : 0;
    S->assign_maxI = nI > 0 ? *std::max_element(hdI, hdI + nI) : 0;
    S->assign_own_user = want_own >= 0 ? want_own : (S->assign_maxU > S->assign_maxI ? 1 : 0);
    S->assign_want = want_own;
    const int32_t* degOwn = S->assign_own_user ? hdU : hdI;
    const int64_t nOwnA = S->assign_own_user ? nU : nI;
    S->owner.assign((size_t)nOwnA, 0);
    std::vector<int32_t> rows;
    for (int64_t r = 0; r < nOwnA; r++)
      if (degOwn[r] > 0) rows.push_back((int32_t)r);
    std::sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) { return degOwn[a] != degOwn[b] ? degOwn[a] > degOwn[b] : a < b; });
    typedef std::pair<int64_t, int32_t> Load;
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int64_t gI = 0; gI < groups; gI++) heap.push(Load(0, (int32_t)gI));
    std::vector<int32_t> nrows((size_t)groups, 0);
    for (int32_t r : rows) {
      Load l = heap.top();
      heap.pop();
      // bits 26-31: the row's place in its queue's LDS cache (tagged schedule)
      S->owner[(size_t)r] = (int32_t)((uint32_t)l.second | (uint32_t)(nrows[(size_t)l.second]++ & 63) << FL_SLOT_SHIFT);
      l.first += degOwn[r];
      heap.push(l);
    }
    // rows this list does not visit: dealt round the queues, not piled onto queue 0 (another list replayed with this assignment
    // -- same matrix, same length -- may visit them; ownership, not balance, is what correctness needs)
    {
      int64_t rr = 0;
      for (int64_t r = 0; r < nOwnA; r++)
        if (degOwn[r] == 0) {
          const int32_t gq = (int32_t)(rr++ % groups);
          S->owner[(size_t)r] = (int32_t)((uint32_t)gq | (uint32_t)(nrows[(size_t)gq]++ & 63) << FL_SLOT_SHIFT);
        }
    }
    HIPCHK(hipMemcpyAsync(S->downer, S->owner.data(), sizeof(int32_t) * (size_t)nOwnA, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));          // S->owner may be reassigned by the host builder
    S->assign_gen = ctx->train_gen; S->assign_groups = groups; S->assign_nU = nU; S->assign_nI = nI; S->assign_count = count;
    S->hy_gen = ~0ull;                     // (downer is shared with the hybrid builder's item queues)
  }
  const int own_user = S->assign_own_user;
  const int32_t maxU = S->assign_maxU, maxI = S->assign_maxI;
  const int64_t nOth = own_user ? nI : nU;
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
  // chain starts of the other side in the sorted order: dstart[r] = number of sorted keys < r
  hipLaunchKernelGGL(flow_bounds32_kernel, dim3((unsigned)std::min<int64_t>((nOth + 256) / 256, 4096)), dim3(256), 0, ctx->stream, S->k1, count,
                     nOth + 1, S->dstart);
  hipLaunchKernelGGL(flow_pack_kernel, dim3(grid), dim3(256), 0, ctx->stream, eu, ei, er, count, own_user,
                     tagged ? (const int32_t*)S->downer : (const int32_t*)nullptr, S->pack);
  hipLaunchKernelGGL(flow_rank_kernel, dim3(grid), dim3(256), 0, ctx->stream, S->k1, S->v1, S->dstart, count, S->pack);
  // queue order: stable sort of the list positions by the queue of their owned row
  hipLaunchKernelGGL(flow_keys_kernel, dim3(grid), dim3(256), 0, ctx->stream, own_rows, (const int32_t*)S->downer, count, S->k0, S->v0);
  bytes = S->sort_tmp_bytes;
  HIPCHK(rocprim::radix_sort_pairs(S->sort_tmp, bytes, S->k0, S->k1, S->v0, S->lpos, (size_t)count, 0, bits_for((uint64_t)groups), ctx->stream));
  hipLaunchKernelGGL(flow_bounds_kernel, dim3((unsigned)std::min<int64_t>((groups + 256) / 256, 4096)), dim3(256), 0, ctx->stream, S->k1, count,
                     groups + 1, S->qoff);
  hipLaunchKernelGGL(flow_gather2_kernel, dim3(grid), dim3(256), 0, ctx->stream, S->lpos, (const int4*)S->pack, count, S->q);
  HIPCHK(hipGetLastError());
  S->hoff.resize((size_t)groups + 1);
  HIPCHK(hipMemcpyAsync(S->hoff.data(), S->qoff, sizeof(int64_t) * ((size_t)groups + 1), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int64_t longest = 0;
  for (int64_t gI = 0; gI < groups; gI++) longest = std::max(longest, S->hoff[(size_t)gI + 1] - S->hoff[(size_t)gI]);
  S->groups = groups; S->longest = longest; S->own_user = own_user; S->tagged = tagged;
  S->prep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (getenv("MFX_DEBUG"))
    fprintf(stderr, "[mfx] dataflow replay (device construction): %lld ratings on %lld queues (%s rows owned), longest queue %lld, longest chains "
            "%d users / %d items, preparation %.1f ms\n", (long long)count, (long long)groups, own_user ? "user" : "item", (long long)longest, maxU,
            maxI, S->prep_ms);
  return MFX_OK;
}

// ---- hybrid ownership built on the device (the default; MFX_FLOW_HYBRID=host keeps build_flow_hybrid_host as the cross-check) ----
// ratings per item that stay in the item's queue (their user is not heavy)
__global__ void hy_light_degrees_kernel(const int32_t* __restrict__ eu, const int32_t* __restrict__ ei, int64_t n, const int32_t* __restrict__ uq,
                                        int32_t* __restrict__ degL) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
    if (uq[eu[t]] < 0) atomicAdd(degL + ei[t], 1);
}
// sorted by user (stable): rank of list position v1[k] in its user's chain
__global__ void hy_user_rank_kernel(const uint32_t* __restrict__ k1, const uint32_t* __restrict__ v1, const int64_t* __restrict__ start, int64_t n,
                                    uint32_t* __restrict__ ru) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) ru[v1[k]] = (uint32_t)(k - start[k1[k]]);
}
// key = item, value = list position | (its user is heavy) << 31: the item-sorted order then knows of every visit whether it lies in a
// heavy user's queue without going back to the list
__global__ void hy_item_keys_kernel(const int32_t* __restrict__ ei, const int32_t* __restrict__ eu, const int32_t* __restrict__ uq, int64_t n,
                                    uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    key[t] = (uint32_t)ei[t];
    val[t] = (uint32_t)t | (uq[eu[t]] >= 0 ? 0x80000000u : 0u);
  }
}
// sorted by item (stable): rank in the item's chain << 1 | "the item's next visit lies in another queue (a heavy user's), or there is none"
__global__ void hy_item_rank_kernel(const uint32_t* __restrict__ k1, const uint32_t* __restrict__ v1, const int64_t* __restrict__ start, int64_t n,
                                    uint32_t* __restrict__ ri) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    const uint32_t it = k1[k];
    const bool next_same = k + 1 < n && k1[k + 1] == it;
    const bool pub = !next_same || (v1[k + 1] >> 31) != 0u;
    ri[v1[k] & 0x7fffffffu] = (uint32_t)(k - start[it]) << 1 | (pub ? 1u : 0u);
  }
}
// the queue record, rank | pub and queue of every list position (list order)
__global__ void hy_pack_kernel(const int32_t* __restrict__ eu, const int32_t* __restrict__ ei, const float* __restrict__ er, int64_t n,
                               const int32_t* __restrict__ uq, const int32_t* __restrict__ owner, const int32_t* __restrict__ degU,
                               const uint32_t* __restrict__ ru, const uint32_t* __restrict__ ri, int g_item, int4* __restrict__ pack,
                               uint32_t* __restrict__ qa, uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    const int u = eu[t], it = ei[t];
    const int h = uq[u];
    const uint32_t a = ru[t], b = ri[t];
    if (h >= 0) {      // the heavy user's queue: user row owned, the item row through its tags; published behind the user's last rating
      pack[t] = make_int4(it, u, __float_as_int(er[t]), (int)(b >> 1));
      qa[t] = a << 1 | (a + 1 == (uint32_t)degU[u] ? 1u : 0u);
      key[t] = (uint32_t)(g_item + h);
    } else {
      const uint32_t ow = (uint32_t)owner[it];
      pack[t] = make_int4(u, (int)((uint32_t)it | (ow >> FL_SLOT_SHIFT) << FL_SLOT_SHIFT), __float_as_int(er[t]), (int)a);
      qa[t] = b;
      key[t] = ow & (uint32_t)FL_ROW_MASK;
    }
    val[t] = (uint32_t)t;
  }
}
__global__ void hy_gather_kernel(const uint32_t* __restrict__ lpos, const int4* __restrict__ pack, const uint32_t* __restrict__ qa_list, int64_t n,
                                 int4* __restrict__ q, uint32_t* __restrict__ qa) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) { q[t] = pack[lpos[t]]; qa[t] = qa_list[lpos[t]]; }
}

// Same lists as build_flow_hybrid_host.  Cached per train matrix, queue count and list length: the heavy users (uq), the item queues and
// LDS slots (downer) -- they follow from the chain lengths, which a permutation of the same ratings does not change.  Per call: three
// stable radix sorts (by user: ranks in the users' chains; by item: ranks in the items' chains and the publish flags; by queue).
// Returns 1 (nothing built) for a user-ordered list or when no user is heavy enough.
int build_flow_hybrid_device(mfx_ctx* ctx, int64_t first, int64_t count, int64_t groups, int heavy_max, int heavy_min) {
  FlowState* S = fl(ctx);
  const auto t0 = std::chrono::steady_clock::now();
  int rc;
  const int64_t nU = ctx->nU, nI = ctx->nI;
  NEED(count < ((int64_t)1 << 31), MFX_E_ARG, "sgd dataflow (hybrid): list of %lld ratings", (long long)count);
  if ((rc = flow_device_buffers(ctx, count, groups))) return rc;
  if (S->hy_cap < count) {
    dev_free(S->hy_ri); dev_free(S->hy_qa);
    S->hy_cap = 0;
    if ((rc = dev_alloc(ctx, &S->hy_ri, (size_t)count)) || (rc = dev_alloc(ctx, &S->hy_qa, (size_t)count))) return rc;
    S->hy_cap = count;
  }
  if (S->qa_cap < count) {
    dev_free(S->qa);
    S->qa_cap = 0;
    if ((rc = dev_alloc(ctx, &S->qa, (size_t)count))) return rc;
    S->qa_cap = count;
  }
  if (S->duq_cap < nU) {
    dev_free(S->duq);
    S->duq_cap = 0; S->hy_gen = ~0ull;
    if ((rc = dev_alloc(ctx, &S->duq, (size_t)nU))) return rc;
    S->duq_cap = nU;
  }
  const int32_t *eu = ctx->eu + first, *ei = ctx->ei + first;
  const float* er = ctx->er + first;
  const int grid = (int)std::min<int64_t>((count + 255) / 256, 8192);
  if (count > 1) {          // a user-ordered list: user rows are owned there (the ordinary queues)
    unsigned long long* adj = (unsigned long long*)S->dstart;
    unsigned long long hadj[2] = {0, 0};
    HIPCHK(hipMemsetAsync(adj, 0, sizeof hadj, ctx->stream));
    hipLaunchKernelGGL(flow_adjacent_kernel, dim3(std::min(grid, 1024)), dim3(256), 0, ctx->stream, eu, ei, count, adj);
    HIPCHK(hipMemcpyAsync(hadj, adj, sizeof hadj, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (2 * hadj[0] > (unsigned long long)count) return 1;
  }
  if (S->hy_gen != ctx->train_gen || S->hy_groups != groups || S->hy_nU != nU || S->hy_nI != nI || S->hy_count != count ||
      S->hy_heavy_max != heavy_max || S->hy_heavy_min != heavy_min) {
    HIPCHK(hipMemsetAsync(S->degU, 0, sizeof(int32_t) * (size_t)(nU + nI), ctx->stream));
    hipLaunchKernelGGL(flow_degrees_kernel, dim3(grid), dim3(256), 0, ctx->stream, eu, ei, count, S->degU, S->degI);
    S->hdeg.resize((size_t)(nU + nI));
    HIPCHK(hipMemcpyAsync(S->hdeg.data(), S->degU, sizeof(int32_t) * (size_t)(nU + nI), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int32_t *degU = S->hdeg.data(), *degI = S->hdeg.data() + nU;
    std::vector<int32_t> heavy;
    for (int64_t u = 0; u < nU; u++)
      if (degU[u] >= heavy_min) heavy.push_back((int32_t)u);
    std::sort(heavy.begin(), heavy.end(), [&](int32_t a, int32_t b) { return degU[a] != degU[b] ? degU[a] > degU[b] : a < b; });
    if ((int64_t)heavy.size() > heavy_max) heavy.resize((size_t)heavy_max);
    const int H = (int)std::min<int64_t>((int64_t)heavy.size(), groups / 2);
    S->hy_gen = ~0ull;
    if (H == 0) {                          // remembered: the next call over this matrix asks no more than the key
      S->hy_gen = ctx->train_gen; S->hy_groups = groups; S->hy_nU = nU; S->hy_nI = nI; S->hy_count = count;
      S->hy_heavy_max = heavy_max; S->hy_heavy_min = heavy_min; S->hy_H = 0;
      return 1;
    }
    S->assign_gen = ~0ull;                 // downer no longer holds the ordinary assignment
    heavy.resize((size_t)H);
    const int64_t g_item = groups - H;
    std::vector<int32_t> uq((size_t)nU, -1);
    for (int h = 0; h < H; h++) uq[(size_t)heavy[(size_t)h]] = h;
    HIPCHK(hipMemcpyAsync(S->duq, uq.data(), sizeof(int32_t) * (size_t)nU, hipMemcpyHostToDevice, ctx->stream));
    // items -> item queues by the ratings that stay with them (longest first onto the lightest queue), LDS slots as the host statement
    int32_t* degL = (int32_t*)S->k0;       // (scratch: the sort buffers are written further down)
    HIPCHK(hipMemsetAsync(degL, 0, sizeof(int32_t) * (size_t)nI, ctx->stream));
    hipLaunchKernelGGL(hy_light_degrees_kernel, dim3(grid), dim3(256), 0, ctx->stream, eu, ei, count, (const int32_t*)S->duq, degL);
    std::vector<int32_t> hdegL((size_t)nI);
    HIPCHK(hipMemcpyAsync(hdegL.data(), degL, sizeof(int32_t) * (size_t)nI, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    S->hpacked.assign((size_t)nI, 0);
    {
      std::vector<int32_t> rows;
      for (int64_t r = 0; r < nI; r++)
        if (degI[r] > 0) rows.push_back((int32_t)r);
      std::sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) { return hdegL[a] != hdegL[b] ? hdegL[a] > hdegL[b] : a < b; });
      typedef std::pair<int64_t, int32_t> Load;
      std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
      for (int64_t gI = 0; gI < g_item; gI++) heap.push(Load(0, (int32_t)gI));
      std::vector<int32_t> nrows((size_t)g_item, 0);
      for (int32_t r : rows) {
        Load l = heap.top();
        heap.pop();
        S->hpacked[(size_t)r] = (int32_t)((uint32_t)l.second | (uint32_t)(nrows[(size_t)l.second]++ & 31) << FL_SLOT_SHIFT);
        l.first += hdegL[(size_t)r];
        heap.push(l);
      }
    }
    HIPCHK(hipMemcpyAsync(S->downer, S->hpacked.data(), sizeof(int32_t) * (size_t)nI, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    S->hy_gen = ctx->train_gen; S->hy_groups = groups; S->hy_nU = nU; S->hy_nI = nI; S->hy_count = count;
    S->hy_heavy_max = heavy_max; S->hy_heavy_min = heavy_min; S->hy_H = H;
  }
  const int H = S->hy_H;
  if (H == 0) return 1;
  const int64_t g_item = groups - H;
  HIPCHK(hipMemsetAsync(S->flag, 0, sizeof(unsigned), ctx->stream));
  size_t bytes;
  // ranks in the users' chains
  hipLaunchKernelGGL(flow_keys_kernel, dim3(grid), dim3(256), 0, ctx->stream, eu, (const int32_t*)nullptr, count, S->k0, S->v0);
  bytes = S->sort_tmp_bytes;
  HIPCHK(rocprim::radix_sort_pairs(S->sort_tmp, bytes, S->k0, S->k1, S->v0, S->v1, (size_t)count, 0, bits_for((uint64_t)nU), ctx->stream));
  hipLaunchKernelGGL(flow_bounds32_kernel, dim3((unsigned)std::min<int64_t>((nU + 256) / 256, 4096)), dim3(256), 0, ctx->stream, S->k1, count, nU + 1, S->dstart);
  hipLaunchKernelGGL(hy_user_rank_kernel, dim3(grid), dim3(256), 0, ctx->stream, S->k1, S->v1, S->dstart, count, S->vexp);
  // ranks in the items' chains and the publish flags of the visits that stay in the item queues
  hipLaunchKernelGGL(hy_item_keys_kernel, dim3(grid), dim3(256), 0, ctx->stream, ei, eu, (const int32_t*)S->duq, count, S->k0, S->v0);
  bytes = S->sort_tmp_bytes;
  HIPCHK(rocprim::radix_sort_pairs(S->sort_tmp, bytes, S->k0, S->k1, S->v0, S->v1, (size_t)count, 0, bits_for((uint64_t)nI), ctx->stream));
  hipLaunchKernelGGL(flow_bounds32_kernel, dim3((unsigned)std::min<int64_t>((nI + 256) / 256, 4096)), dim3(256), 0, ctx->stream, S->k1, count, nI + 1, S->dstart);
  hipLaunchKernelGGL(hy_item_rank_kernel, dim3(grid), dim3(256), 0, ctx->stream, S->k1, S->v1, S->dstart, count, S->hy_ri);
  // records, queues, queue order
  hipLaunchKernelGGL(hy_pack_kernel, dim3(grid), dim3(256), 0, ctx->stream, eu, ei, er, count, (const int32_t*)S->duq, (const int32_t*)S->downer,
                     (const int32_t*)S->degU, (const uint32_t*)S->vexp, (const uint32_t*)S->hy_ri, (int)g_item, S->pack, S->hy_qa, S->k0, S->v0);
  bytes = S->sort_tmp_bytes;
  HIPCHK(rocprim::radix_sort_pairs(S->sort_tmp, bytes, S->k0, S->k1, S->v0, S->lpos, (size_t)count, 0, bits_for((uint64_t)groups), ctx->stream));
  hipLaunchKernelGGL(flow_bounds_kernel, dim3((unsigned)std::min<int64_t>((groups + 256) / 256, 4096)), dim3(256), 0, ctx->stream, S->k1, count,
                     groups + 1, S->qoff);
  hipLaunchKernelGGL(hy_gather_kernel, dim3(grid), dim3(256), 0, ctx->stream, S->lpos, (const int4*)S->pack, (const uint32_t*)S->hy_qa, count, S->q, S->qa);
  HIPCHK(hipGetLastError());
  S->hoff.resize((size_t)groups + 1);
  HIPCHK(hipMemcpyAsync(S->hoff.data(), S->qoff, sizeof(int64_t) * ((size_t)groups + 1), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int64_t longest = 0;
  for (int64_t gI = 0; gI < groups; gI++) longest = std::max(longest, S->hoff[(size_t)gI + 1] - S->hoff[(size_t)gI]);
  S->groups = groups; S->longest = longest; S->own_user = 0; S->tagged = true;
  S->hybrid = true; S->g_item = (int)g_item; S->n_heavy = H;
  S->prep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (getenv("MFX_DEBUG"))
    fprintf(stderr, "[mfx] dataflow replay, hybrid ownership (device construction): %lld ratings, %lld item queues + %d heavy users' queues, longest queue "
            "%lld, preparation %.1f ms\n", (long long)count, (long long)g_item, H, (long long)longest, S->prep_ms);
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

// ---- tagged schedule: granule copy of the other side, the launch, and the copy back ---------------------------------
template <int L, int C, int ARITH>
int launch_flow_tag_lca(mfx_ctx* ctx, const mfx_sgd_opts* o, int blocks) {
  typedef F2<L, C> P;
  FlowState* S = fl(ctx);
  const int64_t nOth = S->own_user ? ctx->nI : ctx->nU;
  float* X = S->own_user ? ctx->V : ctx->U;          // the other side: read and written through its granule copy
  float* O = S->own_user ? ctx->U : ctx->V;          // the owned side
  // rows of 64 C floats: one element per lane and chunk (sgd_flow_wide_kernel; MFX_FLOW_WIDE=0 keeps the 16-lane kernel, the cross-check)
  typedef void (*FlowKern)(const int4*, const int64_t*, uint32_t, float*, uint32_t, float*, uint32_t, int, float, float, float, unsigned*,
                           const uint32_t*, int);
  FlowKern kern = sgd_flow_tag_kernel<L, C, ARITH>;
  int which = 0;
  if constexpr (L == 16) {
    static_assert(FW<C>::LDS == P::LDS && FW<C>::WGS == P::WGS && FW<C>::QR == P::QR, "the two tagged kernels share the launch shape");
    const char* we = getenv("MFX_FLOW_WIDE");
    if (!(we && we[0] == '0')) { kern = sgd_flow_wide_kernel<C, ARITH>; which = 1; }
    if (S->hybrid) { kern = sgd_flow_wide_kernel<C, ARITH, true>; which = 2; }
  } else {
    NEED(!S->hybrid, MFX_E_STATE, "sgd dataflow: hybrid ownership runs on the wide kernel (ranks above 32)");
  }
  static bool attr_done[3] = {false, false, false};  // per instantiation
  if (!attr_done[which]) {
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS));
    attr_done[which] = true;
  }
  int per_cu = 0;
  HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, FL_WG, (size_t)P::LDS));
  int dev = 0, cus = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  NEED((int64_t)per_cu * cus >= blocks, MFX_E_STATE, "sgd dataflow (tagged): %d workgroups do not fit the device (%d per CU x %d CUs)", blocks,
       per_cu, cus);
  static const bool prio = [] { const char* e = getenv("MFX_FLOW_PRIO"); return e && atoi(e) != 0; }();   // (measured: no effect; off)
  const char* pe_ = getenv("MFX_FLOW_POLE");                                                                // (read per call: tests switch it)
  const bool pole = !pe_ || atoi(pe_) != 0;
  const char* pf_ = getenv("MFX_FLOW_POLLFULL");
  const int pollfull = pf_ ? std::max(0, std::min(65535, atoi(pf_))) : 0;
  const int64_t total = nOth * L * C;
  const int tgrid = (int)std::min<int64_t>((total + 255) / 256, 8192);
  if (S->hybrid) {
    // both tables in granule form for the launch (users' = S->tagbuf, items' = S->tagbuf2), both copied back behind it
    const int64_t need2 = (int64_t)ctx->nI * 2 * P::LD;
    if (S->tag2_cap < need2) {
      dev_free(S->tagbuf2);
      S->tag2_cap = 0;
      int rc = dev_alloc(ctx, &S->tagbuf2, (size_t)need2);
      if (rc) return rc;
      S->tag2_cap = need2;
    }
    const int t2grid = (int)std::min<int64_t>(((int64_t)ctx->nI * L * C + 255) / 256, 8192);
    hipLaunchKernelGGL(flow_tag_kernel, dim3(tgrid), dim3(256), 0, ctx->stream, (const float*)ctx->U, (int64_t)ctx->nU, L, C, S->tagbuf);
    hipLaunchKernelGGL(flow_tag_kernel, dim3(t2grid), dim3(256), 0, ctx->stream, (const float*)ctx->V, (int64_t)ctx->nI, L, C, S->tagbuf2);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(FL_WG), (size_t)P::LDS, ctx->stream, (const int4*)S->q, S->qoff,
                       (uint32_t)((uint64_t)S->hoff.back() * 16u), S->tagbuf, (uint32_t)((uint64_t)ctx->nU * 8u * P::LD), S->tagbuf2,
                       (uint32_t)((uint64_t)ctx->nI * 8u * P::LD), (prio ? 2 : 0) | (pole ? 4 : 0) | (pollfull << 8), o->learnRate, o->uReg, o->iReg,
                       S->flag, (const uint32_t*)S->qa, S->g_item);
    hipLaunchKernelGGL(flow_untag_kernel, dim3(tgrid), dim3(256), 0, ctx->stream, (const float*)S->tagbuf, (int64_t)ctx->nU, L, C, ctx->U);
    hipLaunchKernelGGL(flow_untag_kernel, dim3(t2grid), dim3(256), 0, ctx->stream, (const float*)S->tagbuf2, (int64_t)ctx->nI, L, C, ctx->V);
    HIPCHK(hipGetLastError());
    return MFX_OK;
  }
  hipLaunchKernelGGL(flow_tag_kernel, dim3(tgrid), dim3(256), 0, ctx->stream, (const float*)X, nOth, L, C, S->tagbuf);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(FL_WG), (size_t)P::LDS, ctx->stream, (const int4*)S->q, S->qoff,
                     (uint32_t)((uint64_t)S->hoff.back() * 16u), S->tagbuf, (uint32_t)((uint64_t)nOth * 8u * P::LD), O,
                     (uint32_t)((uint64_t)(S->own_user ? ctx->nU : ctx->nI) * 4u * P::LD), (S->own_user ? 1 : 0) | (prio ? 2 : 0) | (pole ? 4 : 0) | (pollfull << 8), o->learnRate, o->uReg, o->iReg,
                     S->flag, (const uint32_t*)nullptr, 0);
  hipLaunchKernelGGL(flow_untag_kernel, dim3(tgrid), dim3(256), 0, ctx->stream, (const float*)S->tagbuf, nOth, L, C, X);
  HIPCHK(hipGetLastError());
  return MFX_OK;
}
template <int L, int C>
int launch_flow_tag_lc(mfx_ctx* ctx, const mfx_sgd_opts* o, int blocks) {
  if constexpr (C <= 4) {
    FlowState* S = fl(ctx);
    const int64_t need = (S->own_user ? (int64_t)ctx->nI : (int64_t)ctx->nU) * 2 * F2<L, C>::LD;      // (hybrid: own_user == 0, the users' table)
    if (S->tag_cap < need) {
      dev_free(S->tagbuf);
      S->tag_cap = 0;
      int rc = dev_alloc(ctx, &S->tagbuf, (size_t)need);
      if (rc) return rc;
      S->tag_cap = need;
    }
    ProfScope ps(ctx, MFX_K_SGD);
    switch (o->arith) {
      case MFX_ARITH_REF64: return launch_flow_tag_lca<L, C, MFX_ARITH_REF64>(ctx, o, blocks);
      case MFX_ARITH_REF64F: return launch_flow_tag_lca<L, C, MFX_ARITH_REF64F>(ctx, o, blocks);
      default: return launch_flow_tag_lca<L, C, MFX_ARITH_F32>(ctx, o, blocks);
    }
  } else {
    return mfx_fail(ctx, MFX_E_ARG, "sgd dataflow (tagged): K <= 256");
  }
}
// workgroups per CU of the tagged kernel for a rank shape (0: not built)
int flow_tag_wgs(int L, int C) {
  if (L == 4) return F2<4, 1>::WGS;
  if (L == 8) return F2<8, 1>::WGS;
  switch (C) {
    case 1: return F2<16, 1>::WGS;
    case 2: return F2<16, 2>::WGS;
    case 3: return F2<16, 3>::WGS;
    case 4: return F2<16, 4>::WGS;
  }
  return 0;
}

}  // namespace

void mfx_flow_free_internal(mfx_ctx* ctx) {
  FlowState* S = fl(ctx);
  if (!S) return;
  dev_free(S->q); dev_free(S->qoff); dev_free(S->lpos); dev_free(S->vexp); dev_free(S->ver); dev_free(S->flag);
  dev_free(S->duq); dev_free(S->hy_ri); dev_free(S->hy_qa);
  dev_free(S->k0); dev_free(S->k1); dev_free(S->v0); dev_free(S->v1); dev_free(S->pack); dev_free(S->sort_tmp);
  dev_free(S->degU); dev_free(S->downer); dev_free(S->dstart); dev_free(S->tagbuf); dev_free(S->tagbuf2); dev_free(S->qa);
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
  // the tagged schedule serves the plain update up to K = 256 while a buffer descriptor reaches the granule copy (twice the
  // table) and the queue; MFX_FLOW_TAGGED=0 keeps the version-counter kernel (the cross-check, and what the variants run on)
  const bool variant = ctx->ifw || ctx->tmf_u || ctx->dimreg;
  const char* tg = getenv("MFX_FLOW_TAGGED");
  const bool tagged = !variant && flow_tag_wgs(L, C) > 0 && !(tg && tg[0] == '0') &&
                      (uint64_t)std::max(ctx->nU, ctx->nI) * ctx->ld * 8 < (1ull << 32) && count < ((int64_t)1 << 28);
  int wgs = tagged ? flow_tag_wgs(L, C) : FL_WG_PER_CU;
  if (const char* e = getenv("MFX_FLOW_WGS")) wgs = std::max(1, std::min(wgs, atoi(e)));
  // every workgroup resident: the queues cannot starve each other; no more queues than there are rows to own
  const int64_t per_block = tagged ? (int64_t)(FL_WG / 64) : (int64_t)(FL_WG / 64) * (64 / L);   // tagged: one queue per wavefront
  int blocks = (int)std::min<int64_t>((int64_t)std::max(1, cus) * wgs, (std::max(ctx->nU, ctx->nI) + per_block - 1) / per_block);
  if (const char* e = getenv("MFX_FLOW_BLOCKS")) blocks = std::max(1, std::min(blocks, atoi(e)));   // test knob: few queues, many owned rows each
  const int64_t groups = (int64_t)blocks * per_block;
  // queues and versions are built on the device; MFX_FLOW_HOST=1 keeps the host statement of the same lists (the cross-check)
  // hybrid ownership (wide kernel, plain update): the busiest users get queues of their own.  Built on the device by default;
  // MFX_FLOW_HYBRID=host: the host statement of the same lists (the cross-check), MFX_FLOW_HYBRID=0: item rows owned throughout.
  int rc = 1;
  fl(ctx)->hybrid = false;
  {
    const char* hy = getenv("MFX_FLOW_HYBRID");
    const bool off = hy && hy[0] == '0';
    if (!off && tagged && L == 16 && C <= 4 && !getenv("MFX_FLOW_OWN") && !getenv("MFX_FLOW_HOST") &&
        !(getenv("MFX_FLOW_WIDE") && getenv("MFX_FLOW_WIDE")[0] == '0') && (uint64_t)ctx->nI * ctx->ld * 8 < (1ull << 32)) {
      const char* hm = getenv("MFX_FLOW_HEAVY");
      const int heavy_max = hm ? std::max(0, atoi(hm)) : 32;      // (C2: 32 queues 11.1 ms, 128: 11.6, all 290 qualifying users: 12.6)
      // a user is worth a queue when its chain is long next to a queue's share of the list (at 1.3 us per hand-off a chain of half an
      // average queue already costs more than that queue's own work)
      const int heavy_min = (int)std::max<int64_t>(256, count / (2 * std::max<int64_t>(groups, 1)));
      const bool host = (hy && hy[0] == 'h') || ctx->nI > count || ctx->nU > count;      // (the device statement borrows list-sized buffers for per-row tables)
      if (heavy_max > 0) rc = host ? build_flow_hybrid_host(ctx, first, count, groups, heavy_max, heavy_min)
                                   : build_flow_hybrid_device(ctx, first, count, groups, heavy_max, heavy_min);
      if (rc < 0) return rc;
    }
  }
  if (rc == 1) rc = getenv("MFX_FLOW_HOST") ? build_flow(ctx, first, count, groups, tagged) : build_flow_device(ctx, first, count, groups, tagged);
  if (rc) return rc;
#define MFX_FLOW_GO(LL, CC) rc = tagged ? launch_flow_tag_lc<LL, CC>(ctx, o, blocks) : launch_flow_lc<LL, CC>(ctx, o, blocks)
  if (L == 4) MFX_FLOW_GO(4, 1);
  else if (L == 8) MFX_FLOW_GO(8, 1);
  else switch (C) {
    case 1: MFX_FLOW_GO(16, 1); break;
    case 2: MFX_FLOW_GO(16, 2); break;
    case 3: MFX_FLOW_GO(16, 3); break;
    case 4: MFX_FLOW_GO(16, 4); break;
    case 5: rc = launch_flow_lc<16, 5>(ctx, o, blocks); break;
    case 6: rc = launch_flow_lc<16, 6>(ctx, o, blocks); break;
    case 7: rc = launch_flow_lc<16, 7>(ctx, o, blocks); break;
    case 8: rc = launch_flow_lc<16, 8>(ctx, o, blocks); break;
    default: return mfx_fail(ctx, MFX_E_ARG, "sgd dataflow: unsupported rank shape L=%d C=%d", L, C);
  }
#undef MFX_FLOW_GO
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
  int rc = getenv("MFX_FLOW_HOST") ? build_flow(ctx, first, count, (int64_t)blocks * FL_WG, false)     // one lane = one queue
                                   : build_flow_device(ctx, first, count, (int64_t)blocks * FL_WG, false);
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

#ifdef MFX_FLOW_STATS
extern "C" int mfx_debug_flow_stats(unsigned long long* out, int64_t ngroups) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fl_stats), sizeof(unsigned long long) * 8 * (size_t)ngroups) == hipSuccess ? 0 : -1;
}
#endif

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
  if (S->tagged)      // the LDS slot of the owned row rides in the top six bits of the owned row's field
    for (int64_t t = 0; t < off.back(); t++) records[4 * t + 1] &= FL_ROW_MASK;
  if (qoff) memcpy(qoff, off.data(), sizeof(int64_t) * off.size());
  return MFX_OK;
}

