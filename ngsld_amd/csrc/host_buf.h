// host_buf.h -- gigabytes of raw genotype values in host memory: written once by a reader, read once by the upload (and by
// the exact-order replay of the few pairs the kernels flag).  Not a std::vector -- zero-filling a buffer before the reader
// overwrites it was a third of the read time of configs[2] and 4 s of the streamed configs[4] run -- and on transparent huge
// pages where the host offers them (2 MB alignment + MADV_HUGEPAGE: 600 page faults per 1.2 GB instead of 300,000 on the
// way in, as many fewer pages to give back on the way out; ignored where huge pages are off).
#pragma once

#include <stdlib.h>
#include <sys/mman.h>

#include <cstddef>

namespace ngsld {

inline double *alloc_host_matrix(size_t count) {  // free() releases it; nullptr on failure
  const size_t huge = (size_t)2 << 20, bytes = (count * sizeof(double) + huge - 1) / huge * huge;
  void *q = nullptr;
  if (posix_memalign(&q, huge, bytes ? bytes : huge) != 0) return nullptr;
  (void)madvise(q, bytes, MADV_HUGEPAGE);
  return static_cast<double *>(q);
}

struct HostMatrix {
  double *p = nullptr;
  HostMatrix() = default;
  HostMatrix(const HostMatrix &) = delete;
  HostMatrix &operator=(const HostMatrix &) = delete;
  ~HostMatrix() { free(p); }
  bool alloc(size_t count) {
    free(p);
    p = alloc_host_matrix(count);
    return p != nullptr;
  }
  double *data() const { return p; }
};

}  // namespace ngsld
