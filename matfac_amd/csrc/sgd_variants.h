// sgd_variants.h -- the per-rating update of the three SGD variants, shared by their own kernels (sgd_ifw.hip, sgd_tmf.hip,
// svd.hip: flat lock-free + one-group serial) and by the dataflow replay (sgd_flow.hip).
#ifndef MFX_SGD_VARIANTS_H_
#define MFX_SGD_VARIANTS_H_
#include "sgd_common.h"

// ModelInvPopMF (modelInvPopMF.cpp:152-178): the error term carries the rating's weight
template <int L, int C, int POL>
__device__ __forceinline__ void visit_ifw(const Rows<POL>& Um, const Rows<POL>& Vm, int64_t pe, int64_t qe, float r, float wt,
                                          float lr, float uReg, float iReg) {
  float4v p[C], q[C];
#pragma unroll
  for (int c = 0; c < C; c++) {
    p[c] = Um.ld(pe + c * 4 * L);
    q[c] = Vm.ld(qe + c * 4 * L);
  }
  const float est = group_dot<L, C>(p, q);
  const double diff = (double)r - (double)est;
  const double m2 = (-2.0 * (double)wt) * diff, ru = 2.0 * (double)uReg, ri = 2.0 * (double)iReg, lrd = (double)lr;
#pragma unroll
  for (int c = 0; c < C; c++) {
#pragma unroll
    for (int e = 0; e < 4; e++) p[c][e] = upd_ref64(p[c][e], q[c][e], m2, ru, lrd);
#pragma unroll
    for (int e = 0; e < 4; e++) q[c][e] = upd_ref64(q[c][e], p[c][e], m2, ri, lrd);
  }
#pragma unroll
  for (int c = 0; c < C; c++) {
    Um.st(pe + c * 4 * L, p[c]);
    Vm.st(qe + c * 4 * L, q[c]);
  }
}

// ModelDropoutSigmoid / ModelPoissonDropout (modelDropoutSigmoid.cpp:158-188): the first `rank` dimensions only
template <int L, int C, int POL>
__device__ __forceinline__ void visit_tmf(const Rows<POL>& Um, const Rows<POL>& Vm, int64_t pe, int64_t qe, float r, int rank, int j,
                                          float lr, float uReg, float iReg) {
  float4v p[C], q[C];
#pragma unroll
  for (int c = 0; c < C; c++) {
    p[c] = Um.ld(pe + c * 4 * L);
    q[c] = Vm.ld(qe + c * 4 * L);
  }
  float a = 0.0f;
#pragma unroll
  for (int c = 0; c < C; c++)
#pragma unroll
    for (int x = 0; x < 4; x++)
      if (c * 4 * L + 4 * j + x < rank) a = __builtin_fmaf(p[c][x], q[c][x], a);
  const float est = group_sum<L>(a);
  const float d = r - est;                                        // float diff (:176)
  const double m2 = -2.0 * (double)d, ru = 2.0 * (double)uReg, ri = 2.0 * (double)iReg, lrd = (double)lr;
#pragma unroll
  for (int c = 0; c < C; c++) {
#pragma unroll
    for (int x = 0; x < 4; x++)
      if (c * 4 * L + 4 * j + x < rank) p[c][x] = upd_ref64(p[c][x], q[c][x], m2, ru, lrd);
#pragma unroll
    for (int x = 0; x < 4; x++)
      if (c * 4 * L + 4 * j + x < rank) q[c][x] = upd_ref64(q[c][x], p[c][x], m2, ri, lrd);
  }
#pragma unroll
  for (int c = 0; c < C; c++)
    if (c * 4 * L + 4 * j < rank) {        // pieces entirely beyond the rank are not written back
      Um.st(pe + c * 4 * L, p[c]);
      Vm.st(qe + c * 4 * L, q[c]);
    }
}

// trainSGDParSVD (modelMF.cpp:489-507): per-dimension regulariser, float diff
template <int L, int C, int POL>
__device__ __forceinline__ void visit_dimreg(const Rows<POL>& Um, const Rows<POL>& Vm, int64_t pe, int64_t qe, float r,
                                             float lr, const float4v (&rk)[C]) {
  float4v p[C], q[C];
#pragma unroll
  for (int c = 0; c < C; c++) {
    p[c] = Um.ld(pe + c * 4 * L);
    q[c] = Vm.ld(qe + c * 4 * L);
  }
  const float est = group_dot<L, C>(p, q);
  const float d = r - est;                       // float diff (modelMF.cpp:494)
  const double m2 = -2.0 * (double)d, lrd = (double)lr;
#pragma unroll
  for (int c = 0; c < C; c++) {
#pragma unroll
    for (int e = 0; e < 4; e++) p[c][e] = upd_ref64(p[c][e], q[c][e], m2, 2.0 * (double)rk[c][e], lrd);
#pragma unroll
    for (int e = 0; e < 4; e++) q[c][e] = upd_ref64(q[c][e], p[c][e], m2, 2.0 * (double)rk[c][e], lrd);
  }
#pragma unroll
  for (int c = 0; c < C; c++) {
    Um.st(pe + c * 4 * L, p[c]);
    Vm.st(qe + c * 4 * L, q[c]);
  }
}

#endif
