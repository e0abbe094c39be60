// ld_prep.h -- launch interface of the per-site preprocessing / item kernels (see ld_prep.hip).
#pragma once

#include "ld_device.h"

namespace ngsld {

struct PrepArgs {
  const double *raw;     // [n_sites][n_ind][3] as on disk (or already normalised, normal space)
  const double *maf_in;  // only with normalised_input
  double *planes;        // [n_sites][3][np]
  uint64_t site_stride;  // 3 * np
  uint64_t n_sites;
  uint32_t np, n_ind;
  int log_scale, ignore_miss, normalised_input;
  int text_semantics, call_geno;  // read_data.cpp:83-99 rules; ngsLD.cpp:92-98
  double N_thresh, call_thresh;
  double *maf, *mean_e, *rsx;  // [n_sites]; rsx = 1/sqrt(sum (e - mean)^2)
  int *status;
};

hipError_t launch_prep(const PrepArgs &a, hipStream_t stream);
hipError_t launch_build_items(const uint32_t *row_end, const uint64_t *item_off, uint32_t n_sites, uint32_t ch,
                              Item *items, hipStream_t stream);
hipError_t launch_selftest(const double *in, double *out, hipStream_t stream);

}  // namespace ngsld
