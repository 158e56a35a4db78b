// What a read-only stream reaches on this device: sum of N floats, 16 bytes per lane per load, U loads in flight per lane.
//   hipcc -O3 --offload-arch=gfx950 scripts/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
// Pattern 0: a wave reads 1 KB contiguous per load, consecutive waves consecutive KBs (grid-stride).
// Pattern 1: every 16-lane group reads its own contiguous range (256 B per load): 4 streams per wave, as many streams as groups.
// Pattern 2: as 1, but the 64 groups of a workgroup share a window: group g reads 256 B pieces g, g + 64, ... of the window.
// Pattern 3: as 1 with every piece shifted by 48 bytes: a 256-byte piece touches three 128-byte lines (the CCD++ trips are
// 16-byte aligned, not line aligned).  Pattern 4: as 3 plus a second array read 8 bytes per lane (the 16-bit ids).
// Pattern 5: as 4 plus THREE 4-byte loads per step that every lane of a group takes from the same address (the trip records
// of the CCD++ passes): bytes unchanged, 5 instead of 2 vector-memory instructions per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int PAT, int U>
__global__ __launch_bounds__(1024) void probe(const f4* __restrict__ a, long n4, float* out) {
  f4 acc = {0, 0, 0, 0};
  const int lane = threadIdx.x & 63, j = lane & 15;
  if (PAT == 0) {
    const long stride = (long)gridDim.x * blockDim.x;
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; q + (U - 1) * stride < n4; q += U * stride) {
      f4 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = a[q + u * stride];
#pragma unroll
      for (int u = 0; u < U; u++) acc += v[u];
    }
  } else {
    const long ngrp = (long)gridDim.x * (blockDim.x >> 4);
    const long g = (long)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const long per = n4 / 16 / ngrp;              // 256-byte pieces per group
    if (PAT == 1 || PAT == 3) {
      const long p0 = g * per;
      for (long p = 0; p + U <= per; p += U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = a[(p0 + p + u) * 16 + j + (PAT == 3 ? 3 : 0)];
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
      }
    } else if (PAT == 5) {
      const long per4 = per * 2 / 3;
      const long p0 = g * per4;
      const float2* b = (const float2*)(a + (n4 / 16 * 2 / 3 + 16) * 16);
      const float* r0 = (const float*)a + g * 64, *r1 = r0 + 16 * 1024 * 1024, *r2 = r1 + 16 * 1024 * 1024;
      for (long p = 0; p + U <= per4; p += U) {
        f4 v[U];
        float2 w[U];
        float x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          v[u] = a[(p0 + p + u) * 16 + j + 3]; w[u] = b[(p0 + p + u) * 16 + j + 5];
          x[u] = r0[(p + u) & 63]; y[u] = r1[(p + u) & 63]; z[u] = r2[(p + u) & 63];
        }
#pragma unroll
        for (int u = 0; u < U; u++) { acc += v[u]; acc[0] += w[u].x + w[u].y + x[u] + y[u] + z[u]; }
      }
    } else if (PAT == 4) {
      // two thirds of the buffer as the 16-byte stream, one third as the 8-byte stream
      const long per4 = per * 2 / 3;
      const long p0 = g * per4;
      const float2* b = (const float2*)(a + (n4 / 16 * 2 / 3 + 16) * 16);
      for (long p = 0; p + U <= per4; p += U) {
        f4 v[U];
        float2 w[U];
#pragma unroll
        for (int u = 0; u < U; u++) { v[u] = a[(p0 + p + u) * 16 + j + 3]; w[u] = b[(p0 + p + u) * 16 + j + 5]; }
#pragma unroll
        for (int u = 0; u < U; u++) { acc += v[u]; acc[0] += w[u].x + w[u].y; }
      }
    } else {
      const long w0 = (long)blockIdx.x * (blockDim.x >> 4) * per;   // window of this workgroup, in pieces
      const int gl = threadIdx.x >> 4, G = blockDim.x >> 4;
      for (long p = 0; p + U <= per; p += U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = a[(w0 + (p + u) * G + gl) * 16 + j];
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
      }
    }
  }
  float s = acc[0] + acc[1] + acc[2] + acc[3];
  if (s == 12345.678f) out[0] = s;
}
template <int PAT, int U>
void run(const f4* a, long n4, float* out, int blocks, int threads) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((probe<PAT, U>), dim3(blocks), dim3(threads), 0, 0, a, n4, out);
  hipEventRecord(e0);
  const int R = 10;
  for (int r = 0; r < R; r++) hipLaunchKernelGGL((probe<PAT, U>), dim3(blocks), dim3(threads), 0, 0, a, n4, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("pattern %d  U=%d  blocks %5d x %4d  %.1f MB  %.4f ms  %.2f TB/s\n", PAT, U, blocks, threads, n4 * 16 / 1e6, ms / R, n4 * 16.0 / (ms / R * 1e-3) / 1e12);
}
int main() {
  const long n4 = 640L * 1024 * 1024 / 16;
  f4* a; float* out;
  hipMalloc(&a, n4 * 16 + 4096); hipMalloc(&out, 64);
  hipMemset(a, 0, n4 * 16);
  for (int blocks : {256, 512, 1024, 2048}) {
    run<0, 1>(a, n4, out, blocks, 1024);
    run<0, 2>(a, n4, out, blocks, 1024);
    run<0, 4>(a, n4, out, blocks, 1024);
  }
  for (int blocks : {512, 2048}) {
    run<0, 4>(a, n4, out, blocks * 4, 256);
    run<1, 2>(a, n4, out, blocks, 1024);
    run<1, 4>(a, n4, out, blocks, 1024);
    run<2, 2>(a, n4, out, blocks, 1024);
    run<2, 4>(a, n4, out, blocks, 1024);
    run<3, 2>(a, n4, out, blocks, 1024);
    run<3, 4>(a, n4, out, blocks, 1024);
    run<4, 2>(a, n4, out, blocks, 1024);
    run<5, 1>(a, n4, out, blocks, 1024);
    run<5, 2>(a, n4, out, blocks, 1024);
    run<4, 1>(a, n4, out, blocks, 1024);
  }
  return 0;
}
