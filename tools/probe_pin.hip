// probe_pin.hip -- what a pinned host buffer costs: hipHostMalloc / hipHostFree against huge-page memory + hipHostRegister /
// hipHostUnregister (untouched and touched), and whether a kernel can write through the registered mapping.
// hipcc -O2 --offload-arch=gfx950 tools/probe_pin.hip -o tools/bin/probe_pin
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void fill(uint64_t *p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

int main(int argc, char **argv) {
  const size_t bytes = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 400ull) << 20;
  hipFree(nullptr);
  char *d = nullptr;
  hipMalloc(&d, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    void *p = nullptr;
    hipHostMalloc(&p, bytes);
    double t1 = now();
    hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost);
    double t2 = now();
    hipHostFree(p);
    double t3 = now();
    std::printf("hipHostMalloc %zu MB: alloc %.1f ms, D2H into it %.1f ms (%.1f GB/s), free %.1f ms\n", bytes >> 20, (t1 - t0) * 1e3,
                (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
  }
  for (int touch = 0; touch < 2; ++touch)
    for (int rep = 0; rep < 3; ++rep) {
      double t0 = now();
      void *p = nullptr;
      if (posix_memalign(&p, 2 << 20, bytes)) return 1;
      madvise(p, bytes, MADV_HUGEPAGE);
      if (touch) std::memset(p, 0, bytes);
      double t1 = now();
      hipError_t e = hipHostRegister(p, bytes, hipHostRegisterMapped);
      double t2 = now();
      void *dp = nullptr;
      hipHostGetDevicePointer(&dp, p, 0);
      hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost);
      double t3 = now();
      hipLaunchKernelGGL(fill, dim3((unsigned)((bytes / 8 + 255) / 256)), dim3(256), 0, 0, (uint64_t *)dp, bytes / 8);
      hipDeviceSynchronize();
      double t4 = now();
      const bool ok = ((uint64_t *)p)[12345] == 12345 && ((uint64_t *)p)[bytes / 8 - 1] == bytes / 8 - 1;
      hipHostUnregister(p);
      double t5 = now();
      free(p);
      double t6 = now();
      std::printf("huge pages %s + hipHostRegister (%s): alloc%s %.1f ms, register %.1f ms, D2H %.1f ms (%.1f GB/s), kernel writes through the "
                  "mapping %.1f ms (%.1f GB/s, %s), unregister %.1f ms, free %.1f ms\n",
                  touch ? "touched" : "untouched", hipGetErrorString(e), touch ? " + memset" : "", (t1 - t0) * 1e3, (t2 - t1) * 1e3,
                  (t3 - t2) * 1e3, bytes / (t3 - t2) / 1e9, (t4 - t3) * 1e3, bytes / (t4 - t3) / 1e9, ok ? "values right" : "VALUES WRONG",
                  (t5 - t4) * 1e3, (t6 - t5) * 1e3);
    }
  return 0;
}
