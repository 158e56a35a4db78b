// sgd_common.h -- device helpers shared by the SGD kernels (sgd.hip, sgd_slots.hip).
#ifndef MFX_SGD_COMMON_H_
#define MFX_SGD_COMMON_H_
#include "mfx_internal.h"

typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// Cache policy of the factor-row accesses (experiment knob MFX_SGD_POLICY):
//   0 plain global loads/stores (L1 + write-back L2)
//   1 sc1 loads and stores: agent scope -- loads bypass the CU's L1, stores write through
//     the XCD's L2 (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & visibility")
//   2 nt loads and stores   3 sc1 loads, plain stores   4 sc0 sc1 (system scope) both
template <int POL>
struct Rows {
  float* base;
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ Rows(float* b, uint32_t bytes) : base(b) {
    if (POL != 0) rs = __builtin_amdgcn_make_buffer_rsrc(b, 0, bytes, 0x00020000);
  }
  static constexpr int AUXL = POL == 1 ? 16 : POL == 2 ? 2 : POL == 3 ? 16 : 17;
  static constexpr int AUXS = POL == 1 ? 16 : POL == 2 ? 2 : POL == 3 ? 0 : 17;
  __device__ __forceinline__ float4v ld(int64_t elt) const {
    if (POL == 0) return *(const float4v*)(base + elt);
    uint4v v = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(elt * 4), 0, AUXL);
    return __builtin_bit_cast(float4v, v);
  }
  __device__ __forceinline__ void st(int64_t elt, float4v v) const {
    if (POL == 0) { *(float4v*)(base + elt) = v; return; }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, v), rs, (uint32_t)(elt * 4), 0, AUXS);
  }
  // the same through a 32-bit BYTE offset of the row (one v_lshl_or instead of a 64-bit multiply, a narrowing and a shift) plus a
  // constant that the compiler folds into the instruction's 12-bit immediate offset; buffer policies only.  (The constant was first
  // handed over as the instruction's SGPR offset: correct on every matrix of the test suite, and half-zeroed user rows on a
  // 1.25 M x 1 M matrix at every rank above 64 -- seen as NaN in bench.py's C5 record; tests/test_fullsize_gpu.py now runs a
  // tall matrix with learning rate 0 and wants the factors back bit for bit.  Round 4, from the evidence in hand: what failed were
  // the rows whose byte offset in the table is >= 2^30 -- a 1.25 M-row table of 1 KB rows has 201 424 of them, 16 % of the users,
  // the "fifth" of scripts/nan_rows.py, and no matrix of the test suite has a table beyond 1 GiB -- and only their accesses with
  // a NON-ZERO SGPR offset (chunks 1 ...; chunk 0 passed 0).  Vector offset >= 2^30 plus a scalar offset is treated as out of
  // range (loads return 0, stores are dropped) although the sum is far below num_records; with the constant in the immediate
  // field the same address is in range.  The rule kept here: row offsets beyond 2^30 never meet an SGPR offset.  The other user
  // of the SGPR-offset form, als_wide.hip's BufF, works on per-wave bases with offsets of a few hundred KB and checks it.)
  __device__ __forceinline__ float4v ldb(uint32_t byte, int konst) const {
    static_assert(POL != 0, "byte offsets go through the buffer descriptor");
    return __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rs, byte + (uint32_t)konst, 0, AUXL));
  }
  __device__ __forceinline__ void stb(uint32_t byte, int konst, float4v v) const {
    static_assert(POL != 0, "byte offsets go through the buffer descriptor");
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, v), rs, byte + (uint32_t)konst, 0, AUXS);
  }
};

// All-reduce sum over the L lanes of a group with DPP (no LDS traffic), levels xor 1, 2, 4, 8.  For the
// levels 1 and 2 every lane adds its xor partner (quad_perm); from then on the four lanes of a quad (the
// eight of a half) hold the same bits, so any lane of the partner quad / half serves as the xor partner
// (row_half_mirror, row_mirror): bit for bit the ascending xor butterfly documented in include/mfx.h.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
template <int L>
__device__ __forceinline__ float group_sum(float s) {
  static_assert(L == 4 || L == 8 || L == 16, "group size");
  s = s + dpp_f<0xB1>(s);                 // quad_perm [1,0,3,2]                 (xor 1)
  s = s + dpp_f<0x4E>(s);                 // quad_perm [2,3,0,1]                 (xor 2)
  if (L >= 8) s = s + dpp_f<0x141>(s);    // row_half_mirror: lane i <-> 7-i    (partner quad; xor 4 level)
  if (L >= 16) s = s + dpp_f<0x140>(s);   // row_mirror:      lane i <-> 15-i   (partner half; xor 8 level)
  return s;
}

// p.q in device order (include/mfx.h)
template <int L, int C>
__device__ __forceinline__ float group_dot(const float4v (&p)[C], const float4v (&q)[C]) {
  float a = 0.0f;
#pragma unroll
  for (int c = 0; c < C; c++) {
    a = __builtin_fmaf(p[c].x, q[c].x, a);
    a = __builtin_fmaf(p[c].y, q[c].y, a);
    a = __builtin_fmaf(p[c].z, q[c].z, a);
    a = __builtin_fmaf(p[c].w, q[c].w, a);
  }
  return group_sum<L>(a);
}

// x -= learnRate * (-2.0*diff*y + 2.0*reg*x)      modelMF.cpp:96 / :102
__device__ __forceinline__ float upd_ref64(float x, float y, double m2diff, double reg2, double lr) {
  return (float)((double)x - lr * (m2diff * (double)y + reg2 * (double)x));
}
// row -= learnRate*(-2.0*diff*other + 2.0*reg*row) with the scalars narrowed to
// float by the Eigen row expression                modelMF.cpp:1759 / :1762
__device__ __forceinline__ float upd_f32(float x, float y, float c1, float c2, float lr) {
  return x - lr * (c1 * y + c2 * x);
}


// The reference's update of one rating on the two rows held in registers:
// user first, then the item with the ALREADY UPDATED user row (modelMF.cpp:94-103).
template <int C, int ARITH>
__device__ __forceinline__ void sgd_axpys(float4v (&p)[C], float4v (&q)[C], float r, float est, float lr,
                                          float uReg, float iReg) {
  if (ARITH == MFX_ARITH_F32) {
    // (float)(-2.0 * ((double)r - (double)est)): r - est is exact in double and scaling by 2 commutes with the
    // rounding to float, so the float expression below is the same number (no f64 instructions)
    const float c1 = -2.0f * (r - est);
    const float cu = 2.0f * uReg, ci = 2.0f * iReg;
#pragma unroll
    for (int c = 0; c < C; c++) {
#pragma unroll
      for (int e = 0; e < 4; e++) p[c][e] = upd_f32(p[c][e], q[c][e], c1, cu, lr);
#pragma unroll
      for (int e = 0; e < 4; e++) q[c][e] = upd_f32(q[c][e], p[c][e], c1, ci, lr);
    }
  } else {
    double diff;
    if (ARITH == MFX_ARITH_REF64F) { const float d = r - est; diff = (double)d; }
    else diff = (double)r - (double)est;
    const double m2 = -2.0 * diff;
    const double ru = 2.0 * (double)uReg, ri = 2.0 * (double)iReg, lrd = (double)lr;
#pragma unroll
    for (int c = 0; c < C; c++) {
#pragma unroll
      for (int e = 0; e < 4; e++) p[c][e] = upd_ref64(p[c][e], q[c][e], m2, ru, lrd);
#pragma unroll
      for (int e = 0; e < 4; e++) q[c][e] = upd_ref64(q[c][e], p[c][e], m2, ri, lrd);
    }
  }
}

// Grid barrier of a persistent launch whose workgroups are ALL resident: every workgroup arrives once per phase;
// `target` = workgroups x phases completed.  A waiter that sees no progress for
// ~2 s raises the abort flag and everybody leaves: a grid barrier must not be able to hang the device.
template <int POL>
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned target) {
  // this wave's row stores are visible device-wide (POL 1: acknowledged write-through stores, only the wait is
  // needed; POL 0: L2 write-back) ...
  // (POL 1: an explicit wait -- a workgroup-scope release fence emits no s_waitcnt for global stores, and the arrival below
  // must not overtake them)
  if (POL == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();                                       // ... for every wave of the workgroup, before it arrives
  __shared__ int s_abort;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ab = 0;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ab = 1; break; }
      if (wall_clock64() - t0 > 200000000LL) {           // 100 MHz constant clock: 2 s
        __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ab = 1;
        break;
      }
    }
    s_abort = ab;
  }
  __syncthreads();
  if (POL != 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop stale lines before the next level's row loads
  return s_abort != 0;
}


// ---- per-rating attributes of the sibling models, derived from one pair per user and one per item ----
// ModelInvPopMF (modelInvPopMF.cpp:161-166): (freq, score) pairs -> float wt = 1/(1 + rhoRMS*score of the rarer side)
__device__ __forceinline__ float mfx_ifw_weight(float2 ua, float2 ia, float rho) {
  float wt = ia.y;                       // float wt = invPopI[item]
  if (ia.x > ua.x) wt = ua.y;            // itemFreq[item] > userFreq[u]
  return (float)(1.0 / (1.0 + (double)(rho * wt)));
}
// ModelDropoutSigmoid (modelDropoutSigmoid.cpp:158-160): (freq bits, rank) pairs -> rank of the rarer side
__device__ __forceinline__ int mfx_tmf_rank(int2 ua, int2 ia) {
  return __int_as_float(ua.x) < __int_as_float(ia.x) ? ua.y : ia.y;
}
// ModelPoissonDropout (modelPoissonDropout.cpp:199-207): the update rank of a visit is a Poisson(lambda) draw clipped to
// [1, K].  The reference draws from std::poisson_distribution on one mt19937 per OpenMP thread, i.e. a stream that
// depends on the thread count; here the draw is a pure function of (seed, epoch, user, item): a 32-bit hash mapped to
// (0,1) and inverted through the Poisson CDF by sequential search in double (expneg = exp(-lambda) from the host, so
// host and device run the same IEEE operations).
__host__ __device__ static inline uint32_t mfx_draw_hash(uint32_t seed, uint32_t epoch, uint32_t u, uint32_t item) {
  uint32_t h = mfx_mix32(seed * 0x9e3779b1U + epoch * 0x85ebca6bU + 0x2545f491U);
  h = mfx_mix32(h ^ (u * 0xc2b2ae35U + 0x27d4eb2fU));
  h = mfx_mix32(h ^ (item * 0x165667b1U + 0x9e3779b9U));
  return h;
}
__host__ __device__ static inline int mfx_poisson_rank(int lambda, double expneg, uint32_t h, int K) {
  const double x = ((double)h + 0.5) * (1.0 / 4294967296.0);
  double p = expneg, F = p;
  int k = 0;
  while (x > F && k < 4 * K + 64) {      // the bound only guards against a CDF that stalls below x by round-off
    k++;
    p = p * (double)lambda / (double)k;
    F += p;
  }
  return k < 1 ? 1 : (k > K ? K : k);     // updRank > facDim -> facDim; updRank < EPS -> 1 (:202-207)
}

#endif
