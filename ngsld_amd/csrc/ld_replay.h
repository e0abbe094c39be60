// ld_replay.h -- launch interface of the device-side exact-order replay for called-genotype matrices (ld_replay.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ngsld.h"

namespace ngsld {

struct ReplayHardArgs {
  uint32_t *flags;           // the launch's flag buffer (ld_device.h: counter, list, bitmap); entries gain kFlagDone
  uint32_t flag_cap;
  const uint64_t *row_off;   // [n_sites + 1] plan: records before each row
  const uint64_t *item_off;  // [n_sites + 1] plan: items before each row
  const ngsld_item *items;   // the plan's items
  uint32_t n_sites;
  uint64_t rec_base;         // plan index of the launch's record 0
  const uint64_t *masks;     // [n_sites][4][words] genotype bit sets (classify_hard_kernel)
  uint32_t words, n_ind;
  int ignore_miss;
  int miss_ok;               // "no data" individuals are call_geno's canonical triple: u_lkl / u_pp are the host's values of it
  double u_lkl, u_pp;        // exp(log(1/3)) and the est_maf posterior of that triple, computed by the HOST's libm
  ngsld_rec_std *out_std;    // the launch's records (device memory, or pinned host memory written in place)
  ngsld_rec_ext *out_ext;    // may be null
  int *status;
};

// one lane per listed pair; n_records bounds the grid (a launch cannot flag more pairs than it has)
hipError_t launch_replay_hard(const ReplayHardArgs &a, uint64_t n_records, hipStream_t stream);

}  // namespace ngsld
