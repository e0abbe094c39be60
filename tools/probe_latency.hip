// Dev probe (not part of the product): DEPENDENT-issue cost of the instruction patterns on the pair kernel's
// critical path (one EM iteration is one long dependency chain: f -> W -> s -> 1/s -> R -> t -> reduction -> f).
// CH independent chains of the same pattern are interleaved; CH = 1 is the pure latency, and the CH at which the
// cost per instruction stops falling is the instruction-level parallelism a single wavefront needs to run at the
// issue rate.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_latency.hip -o /tmp/probe_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double mk(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fold32(double x, double y) {
  u32x2 l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  u32x2 h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return mk(l[0], h[0]) + mk(l[1], h[1]);
}
__device__ __forceinline__ double rl(double v, int lane) {
  return mk((unsigned)__builtin_amdgcn_readlane(__double2loint(v), lane),
            (unsigned)__builtin_amdgcn_readlane(__double2hiint(v), lane));
}

// OP: 0 fma, 1 rcp, 2 dpp-mov pair + add (one reduction level), 3 permlane32 swap pair + add, 4 readlane pair + mul,
//     5 mul, 6 ds_swizzle pair + add (LDS crossbar instead of DPP), 7 v_mov_b64 row_newbcast + add
template <int OP, int CH>
__global__ void k(double *out, double seed, int iters) {
  double a[CH];
  for (int i = 0; i < CH; i++) a[i] = seed + i * 0.125 + threadIdx.x * 1e-3;
  const double b = seed * 0.999, c = seed * 1e-3;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int i = 0; i < CH; i++) {
        if (OP == 0) a[i] = fma(a[i], b, c);
        if (OP == 1) a[i] = __builtin_amdgcn_rcp(a[i]);
        if (OP == 2) a[i] = a[i] + dpp_mov<0x128>(a[i]);
        if (OP == 3) a[i] = fold32(a[i], a[i]);
        if (OP == 4) a[i] = b * rl(a[i], 0);
        if (OP == 5) a[i] = a[i] * b;
        if (OP == 6) {
          int lo = __builtin_amdgcn_ds_swizzle(__double2loint(a[i]), 0x041F);  // swap adjacent lanes
          int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(a[i]), 0x041F);
          a[i] = a[i] + __hiloint2double(hi, lo);
        }
      }
    }
  }
  double s = 0;
  for (int i = 0; i < CH; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int CH>
void run(const char *name, int waves_per_simd, double ghz) {
  double *d;
  const int blocks = 256, threads = 256 * waves_per_simd, iters = 2048;
  hipMalloc(&d, (size_t)blocks * threads * 8);
  hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(threads), 0, 0, d, 1.0001, 16);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(threads), 0, 0, d, 1.0001, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 16 * CH;  // pattern instances per wavefront
  printf("%-34s chains %d waves/SIMD %d: %7.2f cycles per pattern per wave, %7.2f per pattern per SIMD\n", name, CH,
         waves_per_simd, ms * 1e-3 * ghz * 1e9 / n, ms * 1e-3 * ghz * 1e9 / n / waves_per_simd);
  hipFree(d);
}

template <int OP>
void sweep(const char *name, double ghz) {
  run<OP, 1>(name, 1, ghz);
  run<OP, 2>(name, 1, ghz);
  run<OP, 3>(name, 1, ghz);
  run<OP, 4>(name, 1, ghz);
  run<OP, 8>(name, 1, ghz);
  run<OP, 1>(name, 2, ghz);
  run<OP, 2>(name, 2, ghz);
  run<OP, 4>(name, 2, ghz);
}

int main() {
  const double ghz = 2.4;  // nominal; cycles below are wall time x 2.4 GHz
  sweep<0>("v_fma_f64", ghz);
  sweep<5>("v_mul_f64", ghz);
  sweep<1>("v_rcp_f64", ghz);
  sweep<2>("2 v_mov_b32_dpp + v_add_f64", ghz);
  sweep<3>("2 v_permlane32_swap + v_add_f64", ghz);
  sweep<4>("2 v_readlane + v_mul_f64", ghz);
  sweep<6>("2 ds_swizzle + v_add_f64", ghz);
  return 0;
}
