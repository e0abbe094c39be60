// Dev probe (not part of the product): issue cost in cycles of the instructions the pair kernel is made of,
// one wavefront per SIMD, long dependent-free streams.  hipcc --offload-arch=gfx950 -O3 tools/probe_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
template <int OP> __global__ void k(double* out, unsigned long long* cyc, double seed) {
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = seed + i * 0.125 + threadIdx.x * 1e-3;
  double b = seed * 0.5, c = seed * 0.25;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) a[i] = fma(a[i], b, c);
        if (OP == 1) a[i] = a[i] * b;
        if (OP == 2) a[i] = a[i] + b;
        if (OP == 3) a[i] = __builtin_amdgcn_rcp(a[i]);
        if (OP == 4) { int lo = __double2loint(a[i]); lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xf, 0xf, false); a[i] = __hiloint2double(__double2hiint(a[i]), lo); }
        if (OP == 5) a[i] = fmax(a[i], b);
        if (OP == 6) { float f = (float)a[i]; f = __builtin_amdgcn_rcpf(f); a[i] = (double)f; }
        if (OP == 7) a[i] = __builtin_amdgcn_sqrt(a[i]);
        if (OP == 8) a[i] = (a[i] > b) ? c : a[i];
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0; for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name, int waves_per_simd) {
  double* d; unsigned long long* c;
  int blocks = 256, threads = 256 * waves_per_simd;
  hipMalloc(&d, blocks * threads * 8); hipMalloc(&c, blocks * 8);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, c, 1.0001);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, c, 1.0001);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), c, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  double n_inst = 256.0 * REP;
  printf("%-28s waves/SIMD %d: %6.2f counter-ticks per wave-instr; wall %.3f ms -> %.2f ns per instr per SIMD-wave\n", name, waves_per_simd, avg / n_inst, ms, ms * 1e6 / n_inst / waves_per_simd);
  hipFree(d); hipFree(c);
}
int main() {
  for (int w = 1; w <= 2; w++) {
    run<0>("v_fma_f64", w); run<1>("v_mul_f64", w); run<2>("v_add_f64", w); run<3>("v_rcp_f64", w);
    run<4>("v_mov_b32_dpp", w); run<5>("v_max_f64", w); run<6>("cvt+v_rcp_f32+cvt", w); run<7>("v_sqrt_f64", w);
    run<8>("v_cmp_f64+2 cndmask", w);
  }
  return 0;
}
