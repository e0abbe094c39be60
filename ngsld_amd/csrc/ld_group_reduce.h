// ld_group_reduce.h -- sums over the 8, 16 or 32 lanes of a group, by DPP steps inside the group in a fixed order (the group
// kernel, ld_kernel_group.h; the genotype-combination kernel, ld_pair_hard.hip).
#pragma once

#include "ld_common.h"

namespace ngsld {

template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) { return v + dpp_mov<CTRL>(v); }

template <int G>
__device__ __forceinline__ double group_sum(double v) {  // sum over the G lanes of a group, result in every lane of it
  v = dpp_add<0xB1>(v);               // quad_perm:[1,0,3,2]
  v = dpp_add<0x4E>(v);               // quad_perm:[2,3,0,1]
  if (G == 8) return dpp_add<0x141>(v);  // row_half_mirror: lane l <-> 7 - l inside each 8-lane half row
  v = dpp_add<0x124>(v);              // row_ror:4
  v = dpp_add<0x128>(v);              // row_ror:8
  if (G == 32) v = fold16(v, v);      // odd rows trade places with even rows of the copy: row0+row1 | row2+row3
  return v;
}

// Three values at once (G = 16 or 32): after the first DPP level a value sits twice in every lane pair, after the second
// four times in every quad -- so the second level runs on TWO registers (t1 | t2 packed by lane parity, and t3) and the
// remaining ones on ONE (lane % 4 == 0: t1, 1: t2, 2 and 3: t3); three quad broadcasts hand the totals back to every lane
// of the group.  G = 16: 31 instructions instead of 36 (7 f64 adds instead of 12); G = 32: one v_permlane16_swap fold
// instead of three.  (G = 8 has only three levels: packing does not pay there.)
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {  // quad_perm broadcast of one lane of every quad
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int G>
__device__ __forceinline__ void group_sum3(double &t1, double &t2, double &t3, bool odd, bool upper) {
  // odd = lane & 1, upper = lane & 2 (loop invariants of the caller)
  if (G == 8) {
    t1 = group_sum<G>(t1); t2 = group_sum<G>(t2); t3 = group_sum<G>(t3);
    return;
  }
  t1 = dpp_add<0xB1>(t1); t2 = dpp_add<0xB1>(t2); t3 = dpp_add<0xB1>(t3);  // quad_perm:[1,0,3,2]
  double u = odd ? t2 : t1;
  u = dpp_add<0x4E>(u); t3 = dpp_add<0x4E>(t3);                              // quad_perm:[2,3,0,1]
  double w = upper ? t3 : u;
  w = dpp_add<0x124>(w);  // row_ror:4
  w = dpp_add<0x128>(w);  // row_ror:8
  if (G == 32) w = fold16(w, w);
  t1 = dpp_quad<0x00>(w);  // quad_perm:[0,0,0,0]
  t2 = dpp_quad<0x55>(w);  // quad_perm:[1,1,1,1]
  t3 = dpp_quad<0xAA>(w);  // quad_perm:[2,2,2,2]
}

}  // namespace ngsld
