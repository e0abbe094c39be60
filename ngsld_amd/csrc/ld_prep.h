// ld_prep.h -- launch interface of the per-site preprocessing / item kernels (see ld_prep.hip).
#pragma once

#include "ld_common.h"

namespace ngsld {

struct PrepArgs {
  const double *raw;     // [n_sites][n_ind][3] as on disk (or already normalised, normal space); a chunk of sites
  uint64_t site0;        // global index of the chunk's first site: outputs go to site0 + k
  const double *maf_in;  // only with normalised_input
  double *planes;        // [n_sites][3][np]
  uint64_t site_stride;  // 3 * np
  uint64_t n_sites;
  uint32_t np, n_ind;
  int log_scale, ignore_miss, normalised_input;
  int text_semantics, call_geno;  // read_data.cpp:83-99 rules; ngsLD.cpp:92-98
  int exact_chain;                // NGSLD_TEST_PREP_EXACT=1 (tests): every triple through the reference's log / exp chain, no fast path
  double N_thresh, call_thresh;
  double *maf, *mean_e, *rsx;  // [n_sites]; rsx = 1/sqrt(sum (e - mean)^2)
  int *status;
  // text genotypes (log scale, no --call_geno): *odd_missing is set when an individual without data -- three (nearly) equal raw
  // values -- is anything but the reader's own triple, missing_canon three times (read_data.cpp:94: log(1/3) through the HOST's
  // libm).  Null: not looked for.  What the device-side replay of called genotypes needs to know (ld_replay.hip: miss_ok).
  int *odd_missing;
  double missing_canon;
};

hipError_t launch_prep(const PrepArgs &a, hipStream_t stream);
// Pair-space enumeration on the device, one thread per row s1 (the s2 walk of ngsLD.cpp:240-282 after the
// distance/SNP cuts, which the host has already turned into row_end):
//   count_only: row_count[s1] = number of candidates that survive the maf[s2] skip and, when rnd_sample < 1,
//               the row's Tausworthe draws;  otherwise: write the row's items (mask + first_record).
struct ItemArgs {
  const uint32_t *row_end;    // [n_sites]
  const uint8_t *keep;        // [n_sites] maf[s] >= min_maf
  const uint64_t *row_seed;   // [n_sites] or null when not sampling
  const uint64_t *row_off;    // [n_sites + 1] records before each row (fill pass)
  const uint64_t *item_off;   // [n_sites + 1] items before each row (fill pass)
  uint64_t *row_count;        // [n_sites] (count pass)
  Item *items;
  uint32_t n_sites, span;
  double rnd_sample;
  int count_only;
};
hipError_t launch_items(const ItemArgs &a, hipStream_t stream);
// skip (may be null): the sites site_skip_kernel marked, see ld_prep.hip
hipError_t launch_pack_scalars(const double *maf, const double *mean_e, const double *rsx, const uint8_t *skip, double *sc4, uint64_t n,
                               hipStream_t stream);
// skip[s] = 1 where the one-locus EM of site s ends below kSkipBelow (a site whose pairs the exact-order replay will settle:
// the pair kernels leave their EM out); *count (preset to 0) = how many
hipError_t launch_site_skip(const double *planes, uint64_t site_stride, uint32_t np, uint32_t n_ind, int ignore_miss, const double *maf,
                            uint64_t n_sites, uint8_t *skip, uint32_t *count, hipStream_t stream);
hipError_t launch_selftest(const double *in, double *out, hipStream_t stream);

}  // namespace ngsld
