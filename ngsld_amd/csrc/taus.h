// taus.h -- gsl_rng_taus, the generator behind the reference's --rnd_sample (ngsLD.cpp:69-70,165-166,277):
// L'Ecuyer's 3-component Tausworthe generator, restated from the algorithm published in GSL's rng/taus.c (GSL is an
// external dependency of the reference and is not in its tree).  "taus", not "taus2": no minimum is forced on the
// state after seeding.  Shared by the host (master stream -> per-row seeds) and the device (per-row streams).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define NGSLD_HD __host__ __device__ __forceinline__
#else
#define NGSLD_HD inline
#endif

namespace ngsld {

struct Taus {
  uint32_t s1, s2, s3;
  NGSLD_HD uint32_t get() {
    s1 = ((s1 & 4294967294u) << 12) ^ (((s1 << 13) ^ s1) >> 19);
    s2 = ((s2 & 4294967288u) << 4) ^ (((s2 << 2) ^ s2) >> 25);
    s3 = ((s3 & 4294967280u) << 17) ^ (((s3 << 3) ^ s3) >> 11);
    return s1 ^ s2 ^ s3;
  }
  // gsl_rng_set: seed 0 -> 1; s1 = LCG(seed), s2 = LCG(s1), s3 = LCG(s2), LCG(n) = (69069 n) mod 2^32; six warm-up draws
  NGSLD_HD void set(uint64_t seed) {
    if (seed == 0) seed = 1;
    s1 = (uint32_t)(69069ull * seed);
    s2 = 69069u * s1;
    s3 = 69069u * s2;
    for (int k = 0; k < 6; ++k) get();
  }
  NGSLD_HD double uniform() { return get() / 4294967296.0; }  // gsl_rng_uniform
  // draw_rnd(r, 0, INF) truncated to unsigned long (gen_func.cpp:117-119, ngsLD.cpp:166): a row's seed
  NGSLD_HD uint64_t row_seed() { return (uint64_t)(0 + uniform() * (double)1000000000000000ull); }
};

}  // namespace ngsld
