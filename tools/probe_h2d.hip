// probe_h2d.hip -- how fast a pageable host matrix reaches the device: one hipMemcpy, T host threads each copying a slice on a
// stream of their own, and the same from pinned memory (the ceiling).  hipcc -O2 --offload-arch=gfx950 tools/probe_h2d.hip -o /tmp/probe_h2d -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const size_t bytes = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1200ull) << 20;
  void *h = nullptr;
  if (posix_memalign(&h, 2 << 20, bytes)) return 1;
  madvise(h, bytes, MADV_HUGEPAGE);
  std::memset(h, 1, bytes);
  char *d = nullptr;
  hipMalloc(&d, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
    std::printf("one hipMemcpy from pageable memory: %.1f ms, %.1f GB/s\n", (now() - t0) * 1e3, bytes / (now() - t0) / 1e9);
  }
  for (int T : {2, 4, 8}) {
    std::vector<hipStream_t> st((size_t)T);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; ++rep) {
      double t0 = now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
          hipSetDevice(0);
          const size_t a = bytes * (size_t)t / (size_t)T, b = bytes * (size_t)(t + 1) / (size_t)T;
          const size_t step = 64ull << 20;
          for (size_t o = a; o < b; o += step) hipMemcpyAsync(d + o, (char *)h + o, std::min(step, b - o), hipMemcpyHostToDevice, st[(size_t)t]);
          hipStreamSynchronize(st[(size_t)t]);
        });
      for (auto &x : th) x.join();
      std::printf("%d threads, a stream each: %.1f ms, %.1f GB/s\n", T, (now() - t0) * 1e3, bytes / (now() - t0) / 1e9);
    }
    for (auto &s : st) hipStreamDestroy(s);
  }
  {  // T threads memcpy into two pinned staging buffers, DMA from there
    const size_t stage = 64ull << 20;
    char *p[2];
    double t0 = now();
    hipHostMalloc((void **)&p[0], stage);
    hipHostMalloc((void **)&p[1], stage);
    std::printf("two pinned staging buffers of 64 MB: %.1f ms to allocate\n", (now() - t0) * 1e3);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t ev[2];
    hipEventCreate(&ev[0]);
    hipEventCreate(&ev[1]);
    for (int T : {4, 8}) {
      t0 = now();
      size_t k = 0;
      for (size_t o = 0; o < bytes; o += stage, ++k) {
        const size_t n = std::min(stage, bytes - o);
        const int b = (int)(k & 1);
        if (k >= 2) hipEventSynchronize(ev[b]);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
          th.emplace_back([&, t] {
            const size_t a = n * (size_t)t / (size_t)T, e = n * (size_t)(t + 1) / (size_t)T;
            std::memcpy(p[b] + a, (char *)h + o + a, e - a);
          });
        for (auto &x : th) x.join();
        hipMemcpyAsync(d + o, p[b], n, hipMemcpyHostToDevice, s);
        hipEventRecord(ev[b], s);
      }
      hipStreamSynchronize(s);
      std::printf("%d threads memcpy into pinned staging + DMA: %.1f ms, %.1f GB/s\n", T, (now() - t0) * 1e3, bytes / (now() - t0) / 1e9);
    }
  }
  {
    void *pin = nullptr;
    double t0 = now();
    hipHostMalloc(&pin, bytes);
    std::printf("hipHostMalloc of the whole matrix: %.1f ms\n", (now() - t0) * 1e3);
    std::memset(pin, 1, bytes);
    t0 = now();
    hipMemcpy(d, pin, bytes, hipMemcpyHostToDevice);
    std::printf("one hipMemcpy from pinned memory: %.1f ms, %.1f GB/s\n", (now() - t0) * 1e3, bytes / (now() - t0) / 1e9);
    t0 = now();
    hipHostRegister(h, bytes, hipHostRegisterDefault);
    std::printf("hipHostRegister of the pageable matrix: %.1f ms\n", (now() - t0) * 1e3);
  }
  return 0;
}
