// f32 MFMA issue-rate probe for gfx950 (diagnostic, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// v_mfma_f32_32x32x2_f32 with NACC independent accumulators per wave and EXTRA dependent vector fmas per
// iteration, at 1, 2 and 4 waves per SIMD.  Result on MI355X (profiles/README.md): vector instructions do not
// overlap with the f32 MFMA, their time adds -- which is what the ALS accumulation loop is written around.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int EXTRA>
__global__ __launch_bounds__(64) void k(float* out, int iters, float seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; i++)
    for (int r = 0; r < 16; r++) acc[i][r] = 0;
  float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x, e = seed;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int x = 0; x < EXTRA; x++) e = __builtin_fmaf(e, 1.0001f, 0.5f);
  }
  float s = e;
  for (int i = 0; i < NACC; i++)
    for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NACC, int EXTRA>
void run(const char* name, int blocks) {
  float* out;
  if (hipMalloc(&out, (size_t)blocks * 64 * 4) != hipSuccess) return;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL((k<NACC, EXTRA>), dim3(blocks), dim3(64), 0, 0, out, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, EXTRA>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s blocks %5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, (double)blocks * iters * NACC * 4096.0 / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4, 0>("4 acc, 1 wave/SIMD", 1024);
  run<4, 0>("4 acc, 2 waves/SIMD", 2048);
  run<1, 0>("1 acc, 2 waves/SIMD", 2048);
  run<4, 12>("4 acc + 12 VALU, 1 wave", 1024);
  run<4, 12>("4 acc + 12 VALU, 2 waves", 2048);
  run<4, 48>("4 acc + 48 VALU, 2 waves", 2048);
  run<4, 48>("4 acc + 48 VALU, 4 waves", 4096);
  return 0;
}
