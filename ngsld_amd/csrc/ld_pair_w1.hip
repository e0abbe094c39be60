// ld_pair_w1.hip -- instantiations of the one-wavefront-per-pair kernel (n_ind <= 512) and the launcher.
#include "ld_device.h"

namespace ngsld {

// n_ind -> (individuals per lane, wavefronts per pair).  One wavefront holds up to 8*64 = 512
// individuals as 18*8 = 144 VGPRs of P; above that 2..8 wavefronts of a workgroup share the pair.
bool pair_config(uint64_t n_ind, int *slots, int *waves) {
  if (n_ind == 0 || n_ind > 4096) return false;  // 8 wavefronts x 8 slots x 64 lanes
  int w = 1;
  while ((n_ind + 64ull * w - 1) / (64ull * w) > 8) w *= 2;
  *waves = w;
  *slots = (int)((n_ind + 64ull * w - 1) / (64ull * w));
  return true;
}

template <int SLOTS>
static hipError_t launch_s(bool masked, bool prefetch, const PairArgs &a, hipStream_t stream) {
  const uint64_t blocks = prefetch ? a.n_items : (a.n_items + 3) / 4;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  const dim3 grid((unsigned)blocks), block(256);
  if (prefetch) {
    if (masked)
      hipLaunchKernelGGL((pair_ld_pf_kernel<SLOTS, true>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((pair_ld_pf_kernel<SLOTS, false>), grid, block, 0, stream, a);
  } else {
    if (masked)
      hipLaunchKernelGGL((pair_ld_kernel<SLOTS, 1, true, false>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((pair_ld_kernel<SLOTS, 1, false, false>), grid, block, 0, stream, a);
  }
  return hipGetLastError();
}

hipError_t launch_pair_wn(int slots, int waves, bool masked, bool prefetch, const PairArgs &a, hipStream_t stream);

hipError_t launch_pair_kernel(int slots, int waves, bool masked, bool prefetch, const PairArgs &a,
                              hipStream_t stream) {
  if (waves != 1) return launch_pair_wn(slots, waves, masked, prefetch, a, stream);
  switch (slots) {
    case 1: return launch_s<1>(masked, prefetch, a, stream);
    case 2: return launch_s<2>(masked, prefetch, a, stream);
    case 3: return launch_s<3>(masked, prefetch, a, stream);
    case 4: return launch_s<4>(masked, prefetch, a, stream);
    case 5: return launch_s<5>(masked, prefetch, a, stream);
    case 6: return launch_s<6>(masked, prefetch, a, stream);
    case 7: return launch_s<7>(masked, prefetch, a, stream);
    case 8: return launch_s<8>(masked, prefetch, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ngsld
