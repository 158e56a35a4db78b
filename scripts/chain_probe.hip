// chain_probe.hip -- latency of DEPENDENT instructions for ONE wave on gfx950 (what the hottest queue of the exact replay pays per
// link of its chain): cycles per link of chains of v_fma_f32, v_fmac with a DPP source, v_mov_dpp, f64 mul / add / fma, the f32 <-> f64
// conversions, v_permlane32_swap + add, v_readlane -> v_add, and s_cbranch taken / not taken.
//   hipcc -O3 --offload-arch=gfx950 scripts/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
#define N_IT 256

template <int WHICH>
__global__ void probe(float* out, unsigned long long* cyc, float seed) {
  float a = seed + threadIdx.x, b = 1.0000001f, c = 0.5f;
  double d = (double)a, e = 1.00000001, f = 0.25;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < N_IT; it++) {
    if (WHICH == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (WHICH == 1) { REP16(asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c));) }
    if (WHICH == 2) { REP16(asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 quad_perm:[3,0,1,2] row_mask:0xf bank_mask:0xf" : "+v"(a));) }
    if (WHICH == 3) { REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(e));) }
    if (WHICH == 4) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(f));) }
    if (WHICH == 5) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(e), "v"(f));) }
    if (WHICH == 6) { REP16(asm volatile("v_cvt_f64_f32 %1, %0\n\tv_cvt_f32_f64 %0, %1" : "+v"(a), "+v"(d));) }   // 2 links
    if (WHICH == 7) { REP16(asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1" : "+v"(a), "+v"(b));) }
    if (WHICH == 8) { REP16(asm volatile("v_readlane_b32 s20, %0, 5\n\ts_nop 3\n\tv_add_f32 %0, s20, %0" : "+v"(a) : : "s20");) }
    if (WHICH == 9) { REP16(asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a));) }
    if (WHICH == 10) { REP16(asm volatile("s_cmp_eq_u32 s20, s20\n\ts_cbranch_scc0 1f\n\ts_nop 0\n1:" ::: "s20", "scc");) }            // not taken
    if (WHICH == 11) { REP16(asm volatile("s_cmp_eq_u32 s20, s20\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n1:" ::: "s20", "scc");) }   // taken, short
    if (WHICH == 12) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));) }
    if (WHICH == 13) { REP16(asm volatile("v_mul_f64 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(d), "+v"(a) : "v"(e), "v"(c));) }      // f64 chain + independent f32 chain
    if (WHICH == 14) { REP16(asm volatile("v_cmp_eq_u32 vcc, %0, %0\n\ts_cmp_eq_u64 vcc, -1\n\ts_cbranch_scc0 1f\n1:" :: "v"(a) : "vcc", "scc");) }  // VALU compare -> scalar branch
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  out[threadIdx.x] = a + (float)d + b;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 16);   // (a sixteenth probe, v_readlane -> s_lshl -> v_or without wait states, and a 200 000-iteration calibration loop were removed: one of them did not return)
  const char* names[] = {"v_fma_f32 chain", "v_fmac_f32_dpp chain", "nop1 + v_mov_dpp chain", "v_mul_f64 chain", "v_add_f64 chain", "v_fma_f64 chain",
                         "cvt f32->f64->f32 (2 links)", "mov+swap32+add (xor-32 level)", "readlane -> nop3 -> v_add (sgpr)", "nop1 + v_add_dpp row_mirror",
                         "s_cmp + branch not taken", "s_cmp + branch taken over 4 nops", "v_add_f32 chain", "mul_f64 chain + add_f32 chain (2 instr)",
                         "v_cmp -> s_cmp vcc -> branch"};
  void (*k[])(float*, unsigned long long*, float) = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>, probe<9>,
                                                      probe<10>, probe<11>, probe<12>, probe<13>, probe<14>};
  for (int w = 0; w < 15; w++) {
    fprintf(stderr, "probe %d\n", w);
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 5; rep++) {
      hipLaunchKernelGGL(k[w], dim3(1), dim3(64), 0, 0, out, cyc, 1.0f);
      unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      if (h < best) best = h;
    }
    printf("%-42s %8.2f s_memtime ticks per group of the 16 x %d\n", names[w], (double)best / (16.0 * N_IT), N_IT); fflush(stdout);
  }
  return 0;
}
