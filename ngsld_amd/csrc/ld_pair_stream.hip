// ld_pair_stream.hip -- instantiations and launcher of the streaming kernel with the candidate's vector resident
// (pair_ld_bres_kernel, ld_device.h): 5,121 .. 10,240 individuals, eight wavefronts per pair, 11 .. 20 blocks of 64
// individuals per wavefront; beyond that kBresTailSlots blocks per wavefront stay resident and the rest is streamed.
// NGSLD_PAIR_KERNEL=stream keeps the plain streaming kernel (tests, A/B).
#include <cstdlib>
#include <cstring>

#include "ld_kernel_stream.h"
#include "ld_dispatch.h"

namespace ngsld {

template <int SLOTS>
static hipError_t launch_bres_s(bool masked, const PairArgs &a, hipStream_t stream) {
  if (masked)
    hipLaunchKernelGGL((pair_ld_bres_kernel<SLOTS, true>), dim3((unsigned)a.n_items), dim3(512), 0, stream, a);
  else
    hipLaunchKernelGGL((pair_ld_bres_kernel<SLOTS, false>), dim3((unsigned)a.n_items), dim3(512), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_pair_bres(int slots, bool masked, const PairArgs &a, hipStream_t stream) {
  if (a.n_items == 0) return hipSuccess;
  // (planes padded to the next 64 individuals and no further: the kernel relies on the cohort ending inside the last block)
  if (a.n_items > 0x7fffffffull || a.np % 64u != 0 || a.n_ind > a.np || a.np - a.n_ind >= 64u || slots < kBresMinSlots)
    return hipErrorInvalidValue;
  if (slots > kBresMaxSlots) {  // more than 10,240 individuals: 10,240 of them resident (kBresTailSlots = 20 blocks of 64 in each of eight wavefronts), the rest streamed
    if (masked)
      hipLaunchKernelGGL((pair_ld_bres_kernel<kBresTailSlots, true, true>), dim3((unsigned)a.n_items), dim3(512), 0, stream, a);
    else
      hipLaunchKernelGGL((pair_ld_bres_kernel<kBresTailSlots, false, true>), dim3((unsigned)a.n_items), dim3(512), 0, stream, a);
    return hipGetLastError();
  }
  switch (slots) {
    case 11: return launch_bres_s<11>(masked, a, stream);
    case 12: return launch_bres_s<12>(masked, a, stream);
    case 13: return launch_bres_s<13>(masked, a, stream);
    case 14: return launch_bres_s<14>(masked, a, stream);
    case 15: return launch_bres_s<15>(masked, a, stream);
    case 16: return launch_bres_s<16>(masked, a, stream);
    case 17: return launch_bres_s<17>(masked, a, stream);
    case 18: return launch_bres_s<18>(masked, a, stream);
    case 19: return launch_bres_s<19>(masked, a, stream);
    case 20: return launch_bres_s<20>(masked, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ngsld
