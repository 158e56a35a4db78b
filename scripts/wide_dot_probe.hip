// Probe for sgd_flow_wide_kernel's dot product: the 16-lane device order (group_dot, 4 elements per lane) against the one-element-
// per-lane form (quad hops + row mirrors + v_permlane16_swap / v_permlane32_swap), bit for bit, and the lane maps of the two swaps.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I matfac_amd/csrc -I include scripts/wide_dot_probe.hip -o /tmp/wdp && /tmp/wdp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "sgd_common.h"

template <int CTRL>
__device__ __forceinline__ float fw_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int C>
__device__ __forceinline__ float wide_dot(const float (&p)[C], const float (&q)[C]) {
  float a = 0.0f;
#pragma unroll
  for (int c = 0; c < C; c++)
#pragma unroll
    for (int x = 0; x < 4; x++) {
      if (c || x) a = fw_dpp<0x93>(a);
      a = __builtin_fmaf(p[c], q[c], a);
    }
  float s = fw_dpp<0xFF>(a);
  s = s + dpp_f<0x141>(s);
  s = s + dpp_f<0x140>(s);
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return s;
}
template <int C>
__global__ void probe(const float* P, const float* Q, float* out, unsigned* maps) {
  const int lane = threadIdx.x;
  // narrow: lanes 0..15 hold 4 elements per chunk
  float4v p4[C], q4[C];
  float pw[C], qw[C];
  for (int c = 0; c < C; c++) {
    const int j = lane & 15;
    p4[c] = *(const float4v*)(P + c * 64 + 4 * j);
    q4[c] = *(const float4v*)(Q + c * 64 + 4 * j);
    pw[c] = P[c * 64 + lane];
    qw[c] = Q[c * 64 + lane];
  }
  out[lane] = group_dot<16, C>(p4, q4);
  out[64 + lane] = wide_dot<C>(pw, qw);
  if (C == 1) {
    auto r = __builtin_amdgcn_permlane16_swap((unsigned)lane, 100u + (unsigned)lane, false, false);
    maps[lane] = r[0]; maps[64 + lane] = r[1];
    auto r2 = __builtin_amdgcn_permlane32_swap((unsigned)lane, 100u + (unsigned)lane, false, false);
    maps[128 + lane] = r2[0]; maps[192 + lane] = r2[1];
  }
}
template <int C>
int run(const float* dP, const float* dQ, float* dO, unsigned* dM) {
  hipLaunchKernelGGL(probe<C>, dim3(1), dim3(64), 0, 0, dP, dQ, dO, dM);
  float o[128];
  hipMemcpy(o, dO, sizeof o, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) bad += memcmp(&o[l], &o[64 + l], 4) != 0 || memcmp(&o[l], &o[0], 4) != 0;
  printf("C=%d narrow %.9g wide %.9g lanes that differ: %d\n", C, o[0], o[64], bad);
  return bad;
}
int main() {
  float hP[256], hQ[256];
  srand(7);
  for (int i = 0; i < 256; i++) { hP[i] = (float)rand() / RAND_MAX - 0.5f; hQ[i] = (float)rand() / RAND_MAX - 0.5f; }
  float *dP, *dQ, *dO; unsigned* dM;
  hipMalloc(&dP, sizeof hP); hipMalloc(&dQ, sizeof hQ); hipMalloc(&dO, 128 * 4); hipMalloc(&dM, 256 * 4);
  hipMemcpy(dP, hP, sizeof hP, hipMemcpyHostToDevice); hipMemcpy(dQ, hQ, sizeof hQ, hipMemcpyHostToDevice);
  int bad = run<1>(dP, dQ, dO, dM) + run<2>(dP, dQ, dO, dM) + run<4>(dP, dQ, dO, dM);
  unsigned m[256];
  hipMemcpy(m, dM, sizeof m, hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane16_swap vdst'", "permlane16_swap src0'", "permlane32_swap vdst'", "permlane32_swap src0'"};
  for (int k = 0; k < 4; k++) {
    printf("%s (vdst = lane, src0 = 100 + lane):", names[k]);
    for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, m[k * 64 + l]);
    printf("\n");
  }
  return bad != 0;
}
