// probe_init.hip -- where a process' first 0.2-0.3 s with the HIP runtime go: each first call timed.
// hipcc -O2 --offload-arch=gfx950 tools/probe_init.hip -o tools/bin/probe_init
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(int *p) { if (p) *p = 1; }

int main() {
  double t = now(), t0 = t;
  auto lap = [&](const char *what) {
    const double n = now();
    std::printf("%-44s %8.2f ms\n", what, (n - t) * 1e3);
    t = n;
  };
  int n = 0;
  hipGetDeviceCount(&n); lap("hipGetDeviceCount (runtime init)");
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); lap("hipGetDeviceProperties");
  hipSetDevice(0); lap("hipSetDevice");
  hipStream_t s[4];
  for (int i = 0; i < 4; ++i) hipStreamCreate(&s[i]);
  lap("4 x hipStreamCreate");
  hipEvent_t e[6];
  for (int i = 0; i < 6; ++i) hipEventCreateWithFlags(&e[i], hipEventDisableTiming);
  lap("6 x hipEventCreate");
  int *d = nullptr;
  hipMalloc(&d, 4); lap("first hipMalloc (4 bytes)");
  char *big = nullptr;
  hipMalloc(&big, 1200ull << 20); lap("hipMalloc 1.2 GB");
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[0], d); lap("first kernel launch (code object load)");
  hipStreamSynchronize(s[0]); lap("sync of that stream");
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[1], d); hipStreamSynchronize(s[1]); lap("launch + sync on a second stream");
  hipMemsetAsync(big, 0, 1200ull << 20, s[2]); hipStreamSynchronize(s[2]); lap("memset 1.2 GB on a third stream");
  hipFree(big); lap("hipFree 1.2 GB");
  std::printf("%-44s %8.2f ms\n", "total", (now() - t0) * 1e3);
  return 0;
}
