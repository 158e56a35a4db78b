// lds_bcast_probe.hip -- what a BROADCAST ds_read_b128 (all 64 lanes, one address) costs on gfx950 next to a lane-distinct one, and
// whether v_pk_fma_f32 of the same waves issues underneath it: sizing for the ALS K <= 64 solve (pivot row through LDS instead of
// 2 016 v_readlane).  8 single-wave workgroups per CU (2 per SIMD), like als_segment_kernel.
//   hipcc -O3 --offload-arch=gfx950 scripts/lds_bcast_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define N_IT 4096

template <int WHICH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(float* out, float seed) {
  __shared__ __attribute__((aligned(16))) float buf[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) buf[i] = seed + i;
  __syncthreads();
  f32x2 acc[16];
#pragma unroll
  for (int p = 0; p < 16; p++) acc[p] = f32x2{seed, seed + p};
  f32x2 l2 = {seed * 0.5f, seed * 0.5f};
  float sink = 0.0f;
  for (int it = 0; it < N_IT; it++) {
    const int base = (it & 7) * 32;                      // floats; uniform
    f32x4 v[8];
    const unsigned lds0 = (unsigned)(uintptr_t)buf;      // LDS byte address
    const unsigned ubase = lds0 + base * 4, lbase = lds0 + 16 * lane;
    if (WHICH == 0 || WHICH == 3 || WHICH == 4) {         // broadcast: every lane the same address
#pragma unroll
      for (int q = 0; q < 8; q++) asm volatile("ds_read_b128 %0, %1" : "=v"(v[q]) : "v"(ubase + 16 * q));
    }
    if (WHICH == 1) {                                    // lane-distinct, conflict-free
#pragma unroll
      for (int q = 0; q < 8; q++) asm volatile("ds_read_b128 %0, %1" : "=v"(v[q]) : "v"(lbase + ((base * 4 + 1024 * q) & 4095)));
    }
    if (WHICH == 5) {                                    // broadcast b32
#pragma unroll
      for (int q = 0; q < 8; q++) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(ubase + 4 * q)); v[q] = f32x4{t, t, t, t}; }
    }
    if (WHICH == 2) {
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = f32x4{seed, seed, seed, seed};
    }
    if (WHICH != 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (WHICH == 2 || WHICH == 3) {                      // 16 v_pk_fma_f32 on the values
#pragma unroll
      for (int q = 0; q < 8; q++) {
        acc[2 * q] = __builtin_elementwise_fma(l2, f32x2{v[q][0], v[q][1]}, acc[2 * q]);
        acc[2 * q + 1] = __builtin_elementwise_fma(l2, f32x2{v[q][2], v[q][3]}, acc[2 * q + 1]);
      }
      asm volatile("" : "+v"(l2));
    } else if (WHICH == 4) {                             // 16 pk_fma with SGPR-pair sources after 32 v_readlane (today's pattern) next to the reads
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const f32x2 row = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc[q][0]), q)),
                           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc[q][1]), q))};
        acc[q] = __builtin_elementwise_fma(l2, row, acc[q]);
      }
#pragma unroll
      for (int q = 0; q < 8; q++) sink += v[q][0] + v[q][3];
    } else {
#pragma unroll
      for (int q = 0; q < 8; q++) sink += v[q][0] + v[q][3];
    }
  }
  float s = sink;
#pragma unroll
  for (int p = 0; p < 16; p++) s += acc[p][0] + acc[p][1];
  out[blockIdx.x * 64 + lane] = s;
}

int main() {
  float* out;
  const int blocks = 256 * 8;
  hipMalloc(&out, (size_t)blocks * 64 * 4);
  const char* names[] = {"8 broadcast ds_read_b128", "8 lane-distinct ds_read_b128", "16 v_pk_fma_f32 (VGPR sources)",
                         "8 broadcast ds_read_b128 + 16 v_pk_fma_f32 on them", "8 broadcast reads + 32 v_readlane + 16 v_pk_fma (SGPR pairs)",
                         "8 broadcast ds_read_b32"};
  void (*k[])(float*, float) = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 6; w++) {
    fprintf(stderr, "probe %d\n", w);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k[w], dim3(blocks), dim3(64), 0, 0, out, 1.0f);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    // 2 waves per SIMD: cycles per iteration of a SIMD's pair of waves at 2.4 GHz, and per CU (8 waves)
    const double cyc = best * 1e-3 * 2.4e9 / N_IT;
    printf("%-64s %8.3f ms  %7.1f cycles per iteration (all 8 waves of a CU each did one)\n", names[w], best, cyc); fflush(stdout);
  }
  return 0;
}
