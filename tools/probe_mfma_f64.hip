// Dev probe (not part of the product): operand layouts of the two f64 MFMA instructions of gfx950 and what they cost
// inside a stream of f64 VALU work -- the facts behind wave_sum3's matrix-pipe reduction (ld_device.h).  Build + run on
// the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_mfma_f64.hip -o /tmp/probe_mfma && /tmp/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

// mode 0: 4x4x4_4b, mode 1: 16x16x4 (register 0 of the result; all four are written to out4)
__global__ void mfma_once(const double *a, const double *b, double *out, double *out4, int mode) {
  const int l = threadIdx.x;
  if (mode == 0) {
    out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
  } else {
    v4d c = {0.0, 0.0, 0.0, 0.0};
    v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
    out[l] = d[0];
    for (int r = 0; r < 4; ++r) out4[4 * l + r] = d[r];
  }
}

// the candidate reduction: three per-lane values -> three wave totals, in every lane with (lane & 3) == v
__device__ __forceinline__ double sum3_mfma(double t1, double t2, double t3) {
  const int l = threadIdx.x & 63;
  const double one = 1.0;
  // stage a: inside every 16-lane block, sums over the four lanes that differ in k = (lane / 4) % 4
  const double e1 = __builtin_amdgcn_mfma_f64_4x4x4f64(t1, one, 0.0, 0, 0, 0);
  const double e2 = __builtin_amdgcn_mfma_f64_4x4x4f64(t2, one, 0.0, 0, 0, 0);
  const double e3 = __builtin_amdgcn_mfma_f64_4x4x4f64(t3, one, 0.0, 0, 0, 0);
  // the results are replicated over j = lane % 4: column j keeps value j
  const int j = l & 3;
  const double w = j == 0 ? e1 : (j == 1 ? e2 : e3);
  // stage b: sums over k again, now as the B operand (column j stays apart)
  const double f = __builtin_amdgcn_mfma_f64_4x4x4f64(one, w, 0.0, 0, 0, 0);
  // stage c: across the four 16-lane rows
  v4d c = {0.0, 0.0, 0.0, 0.0};
  const v4d g = __builtin_amdgcn_mfma_f64_16x16x4f64(one, f, c, 0, 0, 0);
  return g[0];
}

__global__ void sum3_test(const double *in, double *out) {
  const int l = threadIdx.x;
  out[l] = sum3_mfma(in[l], in[64 + l], in[128 + l]);
}

// cost: NF independent-chain FMAs per round, with / without the five MFMAs of the reduction in between
template <bool WITH>
__global__ void cost(const double *in, double *out, long long *cyc, int rounds) {
  const int l = threadIdx.x & 63;
  double x[8];
  for (int q = 0; q < 8; ++q) x[q] = in[l] + q;
  double acc = 0.0;
  const double m = in[64 + l];
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    double red = 0.0;
    if (WITH) red = sum3_mfma(x[0], x[1], x[2]);
#pragma unroll
    for (int k = 0; k < 25; ++k)
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = fma(x[q], m, 1e-3);
    if (WITH) acc += red;
  }
  long long t1 = __builtin_readcyclecounter();
  double s = acc;
  for (int q = 0; q < 8; ++q) s += x[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  double *da, *db, *dout, *dout4;
  long long *dc;
  hipMalloc(&da, 256 * 8); hipMalloc(&db, 256 * 8); hipMalloc(&dout, 1 << 20); hipMalloc(&dout4, 256 * 8 * 4);
  hipMalloc(&dc, 64);
  std::vector<double> a(64), b(64), o(64), o4(256);
  for (int mode = 0; mode < 2; ++mode) {
    printf("== %s ==\n", mode == 0 ? "v_mfma_f64_4x4x4_4b" : "v_mfma_f64_16x16x4");
    // A one-hot, B ones: which D lanes see A-lane x?  (same block, same i, all j)
    for (int which = 0; which < 2; ++which) {
      printf("%s one-hot lane -> D lanes (reg 0) that receive it:\n", which == 0 ? "A" : "B");
      for (int x = 0; x < 64; ++x) {
        for (int l = 0; l < 64; ++l) { a[l] = which == 0 ? (l == x) : 1.0; b[l] = which == 1 ? (l == x) : 1.0; }
        hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, da, db, dout, dout4, mode);
        hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
        hipMemcpy(o4.data(), dout4, 2048, hipMemcpyDeviceToHost);
        if (x < 20 || x % 16 == 0) {
          printf("  %2d:", x);
          for (int l = 0; l < 64; ++l) if (o[l] != 0.0) printf(" %d", l);
          if (mode == 1) {
            printf("   | regs of lane 0:");
            for (int r = 0; r < 4; ++r) printf(" %g", o4[r]);
            printf("  lane 16:");
            for (int r = 0; r < 4; ++r) printf(" %g", o4[64 + r]);
          }
          printf("\n");
        }
      }
    }
    // k index: A one-hot lane x and B one-hot lane y meet iff same block and same k
    printf("A-lane x meets B-lane y (first 20 x; list of y):\n");
    for (int x = 0; x < 20; ++x) {
      printf("  %2d:", x);
      for (int y = 0; y < 64; ++y) {
        for (int l = 0; l < 64; ++l) { a[l] = (l == x); b[l] = (l == y); }
        hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, da, db, dout, dout4, mode);
        hipMemcpy(o4.data(), dout4, 2048, hipMemcpyDeviceToHost);
        hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
        bool hit = false;
        for (int l = 0; l < 64; ++l) hit = hit || o[l] != 0.0;
        if (mode == 1) for (int q = 0; q < 256; ++q) hit = hit || o4[q] != 0.0;
        if (hit) printf(" %d", y);
      }
      printf("\n");
    }
  }
  // the candidate reduction against long-double sums
  std::vector<double> in(192), out(64);
  unsigned long long st = 0x9E3779B97F4A7C15ull;
  for (auto &v : in) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = (double)(st >> 11) / 9007199254740992.0; }
  double *din;
  hipMalloc(&din, 192 * 8);
  hipMemcpy(din, in.data(), 192 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(sum3_test, dim3(1), dim3(64), 0, 0, din, dout);
  hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost);
  long double ref[3] = {0, 0, 0};
  for (int v = 0; v < 3; ++v) for (int l = 0; l < 64; ++l) ref[v] += in[64 * v + l];
  printf("== candidate reduction ==\nexpected %.17Lg %.17Lg %.17Lg\n", ref[0], ref[1], ref[2]);
  for (int l = 0; l < 64; ++l) printf("%s%.17g", l % 4 == 0 ? "\n  " : "  ", out[l]);
  printf("\n");
  // cost inside a VALU stream, one wavefront per SIMD and two
  for (int waves = 1; waves <= 2; ++waves) {
    long long c0 = 0, c1 = 0;
    const int rounds = 2000;
    hipLaunchKernelGGL((cost<false>), dim3(1), dim3(256 * waves), 0, 0, din, dout, dc, rounds);
    hipMemcpy(&c0, dc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((cost<true>), dim3(1), dim3(256 * waves), 0, 0, din, dout, dc, rounds);
    hipMemcpy(&c1, dc, 8, hipMemcpyDeviceToHost);
    printf("%d wavefront(s) per SIMD: 200 FMA per round %.1f cycles, + reduction %.1f cycles (difference %.1f)\n", waves,
           (double)c0 / rounds, (double)c1 / rounds, (double)(c1 - c0) / rounds);
  }
  return 0;
}
