// Dev probe (not part of the product): accuracy of v_rcp_f64 and of the refinement variants the pair
// kernel can choose from, over s in (1e-30, 1].  Build+run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_rcp.hip -o /tmp/probe_rcp && /tmp/probe_rcp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* in, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = in[i];
  double r0 = __builtin_amdgcn_rcp(s);
  double e = fma(-s, r0, 1.0);
  double r1 = fma(r0, e, r0);                 // one Newton step
  double t = fma(e, e, e);
  double rc = fma(r0, t, r0);                 // cubic: r0 (1 + e + e^2)
  double e1 = fma(-s, r1, 1.0);
  double r2 = fma(r1, e1, r1);                // two Newton steps
  out[4 * i] = r0; out[4 * i + 1] = r1; out[4 * i + 2] = rc; out[4 * i + 3] = r2;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> h(n), o(4 * (size_t)n);
  unsigned long long st = 88172645463325252ull;
  for (int i = 0; i < n; i++) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    double u = (double)(st >> 11) / 9007199254740992.0;
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    double ex = -30.0 * ((double)(st >> 11) / 9007199254740992.0);
    h[i] = (0.5 + 0.5 * u) * std::pow(10.0, ex);
  }
  double *di, *dd;
  hipMalloc(&di, n * 8); hipMalloc(&dd, 4 * (size_t)n * 8);
  hipMemcpy(di, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, di, dd, n);
  hipMemcpy(o.data(), dd, 4 * (size_t)n * 8, hipMemcpyDeviceToHost);
  const char* names[4] = {"v_rcp_f64", "1 Newton", "cubic", "2 Newton"};
  for (int v = 0; v < 4; v++) {
    long double mx = 0; int exact = 0;
    for (int i = 0; i < n; i++) {
      long double ref = 1.0L / (long double)h[i];
      long double rel = fabsl(((long double)o[4 * (size_t)i + v] - ref) / ref);
      if (rel > mx) mx = rel;
      if (o[4 * (size_t)i + v] == (double)ref) exact++;
    }
    printf("%-10s max rel err %.3Le (2^%.1Lf)  correctly rounded %.2f%%\n", names[v], mx, log2l(mx), 100.0 * exact / n);
  }
  return 0;
}
