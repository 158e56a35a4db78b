// vector-instruction issue-rate probe for gfx950 (diagnostic, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -w scripts/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
// Exactly counted instruction streams (inline asm, 8 independent registers) at 1, 2, 4 and 8 waves per SIMD with
// nothing waiting on memory: what the chip issues per second.  Prices the `valu_issue` entry of bench.py's roofline
// and the DPP / conversion instructions the SGD step is made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[8], t[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; t[i] = 0.f; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(1.0001f), "v"(0.5f));
        REP8(X)
#undef X
      } else if (KIND == 1) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(t[i]) : "v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == 2) {
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(t[i]) : "v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == 3) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t[i]) : "v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == 4) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(t[i]) : "v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == 5) {
#define X(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(t[i]) : "v"(a[i]));
        REP8(X)
#undef X
      } else if (KIND == 6) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t[i]) : "v"(a[i]), "v"(1.0001f));
        REP8(X)
#undef X
      } else {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&t[(i) & 6]) : "v"(*(double*)&a[(i) & 6]), "v"(*(double*)&a[((i) + 2) & 6]));
        REP8(X)
#undef X
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + t[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name) {
  for (int w : {1, 2, 4, 8}) {
    const int blocks = 256 * w;
    float* out;
    if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess) return;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * 4 * iters * 32;
    printf("%-28s waves/SIMD %d  %7.3f ms  %7.1f G wave-instr/s  (%.2f cycles per instruction per SIMD at 2.4 GHz)\n", name, w, ms,
           winst / ms / 1e6, 256.0 * 4 * 2.4e9 / (winst / (ms * 1e-3)));
    hipFree(out);
  }
}
int main() {
  run<0>("v_fma_f32");
  run<6>("v_mul_f32");
  run<7>("v_pk_fma_f32 (2 fma each)");
  run<4>("v_cvt_f32_i32");
  run<5>("v_cvt_i32_f32");
  run<1>("v_mov_b32_dpp quad_perm");
  run<2>("v_add_f32_dpp quad_perm");
  run<3>("v_mov_b32_dpp row_share");
  return 0;
}
