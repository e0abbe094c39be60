// Dev probe: does v_rcp_f64 (16 cycles/wave-instr standalone) overlap with independent v_fma_f64 work?
// Streams per iteration: A = 8 rcp; B = 64 fma; C = 8 rcp + 64 fma interleaved (1 rcp : 8 fma), all independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void k(double* out, double seed) {
  double r[8], a[8];
  for (int i = 0; i < 8; i++) { r[i] = seed + i * 0.37 + threadIdx.x * 1e-3; a[i] = seed * 0.5 + i; }
  const double b = 0.9999, c = 1e-9;
  for (int it = 0; it < 4096; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE != 1) r[i] = __builtin_amdgcn_rcp(r[i]) + 1.0;   // keeps the value near 1..2 (adds one v_add)
      if (MODE != 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = fma(a[j], b, c);
      }
    }
  }
  double s = 0; for (int i = 0; i < 8; i++) s += r[i] + a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> float run(int threads) {
  double* d; hipMalloc(&d, 256 * threads * 8);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.0001); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.0001); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d); return ms;
}
int main() {
  for (int threads : {256, 512}) {
    float a = run<0>(threads), b = run<1>(threads), c = run<2>(threads);
    printf("waves/SIMD %d: 8 rcp(+add) %.3f ms | 64 fma %.3f ms | both interleaved %.3f ms (sum %.3f)\n", threads / 256, a, b, c, a + b);
  }
}
