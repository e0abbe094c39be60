// ld_pair_wn.hip -- instantiations of the multi-wavefront-per-pair kernel (960 < n_ind <= 5120; from 513 on with NGSLD_PAIR_KERNEL=multi).
#include "ld_kernel_multi.h"
#include "ld_dispatch.h"

namespace ngsld {

template <int SLOTS, int WAVES>
static hipError_t launch_sw(bool masked, const PairArgs &a, hipStream_t stream) {
  const uint64_t blocks = a.n_items;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  const dim3 grid((unsigned)blocks), block(WAVES * 64);
  if (WAVES == 2 && a.skip_degenerate && a.flags != nullptr) {  // (a matrix with degenerate sites: their pairs' EM is left to the replay)
    if (masked)
      hipLaunchKernelGGL((pair_ld_kernel<SLOTS, WAVES == 2 ? 2 : WAVES, true, WAVES == 2>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((pair_ld_kernel<SLOTS, WAVES == 2 ? 2 : WAVES, false, WAVES == 2>), grid, block, 0, stream, a);
    return hipGetLastError();
  }
  if (masked)
    hipLaunchKernelGGL((pair_ld_kernel<SLOTS, WAVES, true>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((pair_ld_kernel<SLOTS, WAVES, false>), grid, block, 0, stream, a);
  return hipGetLastError();
}

template <int WAVES>
static hipError_t launch_w(int slots, bool masked, const PairArgs &a, hipStream_t stream) {
  switch (slots) {  // pair_config never picks fewer than 5 slots once it needs more than one wavefront
    case 5: return launch_sw<5, WAVES>(masked, a, stream);
    case 6: return launch_sw<6, WAVES>(masked, a, stream);
    case 7: return launch_sw<7, WAVES>(masked, a, stream);
    case 8: return launch_sw<8, WAVES>(masked, a, stream);
    case 9: return launch_sw<9, WAVES>(masked, a, stream);
    case 10: return launch_sw<10, WAVES>(masked, a, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_pair_wn(int slots, int waves, bool masked, const PairArgs &a, hipStream_t stream) {
  switch (waves) {
    case 2: return launch_w<2>(slots, masked, a, stream);
    case 4: return launch_w<4>(slots, masked, a, stream);
    case 8: return launch_w<8>(slots, masked, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ngsld
