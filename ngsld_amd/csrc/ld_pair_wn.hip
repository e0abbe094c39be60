// ld_pair_wn.hip -- instantiations of the multi-wavefront-per-pair kernel (512 < n_ind <= 4608).
#include "ld_device.h"

namespace ngsld {

template <int SLOTS, int WAVES>
static hipError_t launch_sw(bool masked, bool prefetch, const PairArgs &a, hipStream_t stream) {
  const uint64_t blocks = a.n_items;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  const dim3 grid((unsigned)blocks), block(WAVES * 64);
  // every slot but the last one of the last wavefront full?  Otherwise the kernel with per-slot pads (see pair_ld_kernel, PADS)
  const bool clean = (uint64_t)a.n_ind > (uint64_t)(WAVES * SLOTS - 1) * 64u;
  if (prefetch) {
    if (masked)
      hipLaunchKernelGGL((pair_ld_kernel<SLOTS, WAVES, true, true>), grid, block, 0, stream, a);
    else if (clean)
      hipLaunchKernelGGL((pair_ld_kernel<SLOTS, WAVES, false, true>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((pair_ld_kernel<SLOTS, WAVES, false, true, true>), grid, block, 0, stream, a);
  } else if constexpr (SLOTS <= 8) {  // (nine slots: the prefetching kernel only)
    if (masked)
      hipLaunchKernelGGL((pair_ld_kernel<SLOTS, WAVES, true, false>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((pair_ld_kernel<SLOTS, WAVES, false, false>), grid, block, 0, stream, a);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int WAVES>
static hipError_t launch_w(int slots, bool masked, bool prefetch, const PairArgs &a, hipStream_t stream) {
  switch (slots) {  // pair_config never picks fewer than 5 slots once it needs more than one wavefront
    case 5: return launch_sw<5, WAVES>(masked, prefetch, a, stream);
    case 6: return launch_sw<6, WAVES>(masked, prefetch, a, stream);
    case 7: return launch_sw<7, WAVES>(masked, prefetch, a, stream);
    case 8: return launch_sw<8, WAVES>(masked, prefetch, a, stream);
    case 9: return prefetch ? launch_sw<9, WAVES>(masked, prefetch, a, stream) : hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_pair_wn(int slots, int waves, bool masked, bool prefetch, const PairArgs &a, hipStream_t stream) {
  switch (waves) {
    case 2: return launch_w<2>(slots, masked, prefetch, a, stream);
    case 4: return launch_w<4>(slots, masked, prefetch, a, stream);
    case 8: return launch_w<8>(slots, masked, prefetch, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ngsld
