// ld_replay.h -- launch interfaces of the device-side exact-order replays: called-genotype matrices (ld_replay.hip) and
// genotype-likelihood matrices (ld_replay_lkl.hip).
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
  // a launch that flags more pairs than its list holds: its located pairs (launch_replay_expand), null where that route is not set up
  const struct ReplayEntry *list;
  uint32_t *host_bits;       // the launch's host-only bitmap (replay_hard_list_kernel adds what it cannot settle)
};

// one lane per listed pair; n_records bounds the grid (a launch cannot flag more pairs than it has)
hipError_t launch_replay_hard(const ReplayHardArgs &a, uint64_t n_records, hipStream_t stream);
// the overflow route: a persistent grid over a.list (does nothing unless the launch flagged more pairs than its list holds)
hipError_t launch_replay_hard_list(const ReplayHardArgs &a, int n_cus, hipStream_t stream);

// ---- genotype likelihoods (ld_replay_lkl.hip) ----
struct ReplayLklArgs {
  uint32_t *bits;             // the launch's flag bitmap: one bit per record (ld_device.h); launch_replay_expand clears the bits it lists
  uint32_t *host_bits;        // ... and the pairs among them that stay with the host (PairArgs::flags_host); the kernel adds the
                              // pairs whose r2_ExpG it cannot settle itself (see below)
  uint32_t *flags;            // the launch's flag buffer (its counters and host-only list, ld_device.h)
  uint32_t flag_cap;
  uint32_t flag_text;         // the records become text: r2_ExpG values on a sixth-decimal rounding point are the host's
  const double *mean_e, *rsx; // [n_sites] the pair kernels' per-site moments (which pairs they flagged for their Pearson moment)
  uint64_t n_records;         // records in the launch
  uint32_t chunk_words;       // bitmap words per claim (set by launch_replay_lkl)
  uint32_t after_lanes;       // launch_replay_expand has run: flags[6] counts the bits it left in the bitmap
  uint32_t lane_iter_cap;     // replay_lane_kernel: a pair not converged after this many EM steps goes back into the bitmap, for the wavefront-per-pair kernel (0: never)
  uint32_t only_if_overflow;  // launch_replay_expand: do nothing unless the launch flagged more pairs than its list holds (called genotypes)
  uint32_t *work;             // chunk counter of the persistent teams, zero at launch
  uint32_t *done;             // receives the number of pairs replayed (added to)
  const uint64_t *row_off;    // [n_sites + 1] plan: records before each row
  const uint64_t *item_off;   // [n_sites + 1] plan: items before each row
  const ngsld_item *items;    // the plan's items
  uint64_t n_items;
  uint32_t n_sites;
  uint64_t rec_base;          // plan index of the launch's record 0
  // the exact store: normal-space likelihoods and est_maf as the REFERENCE holds them when calc_pair_LD runs (the host's
  // libm, the sequential est_maf), laid out like the pair kernels' planes
  const double *xplanes;      // [n_sites][3][np]
  uint64_t site_stride;       // 3 * np
  uint32_t np;
  const double *xmaf;         // [n_sites]
  uint64_t xt_sites;          // sites of the individual-major copy (lane-per-pair kernel): xT[(i * xt_sites + s) * 3 + g]
  const uint32_t *xperm;      // [n_sites] where a site stands in that copy (may be null: in place): rare sites first, the others behind, each
                              // class in site order -- the rare sites of a stretch of the genome, the shared sites of a wavefront's pairs, are
                              // then neighbours in memory instead of one cache line each (engine_replay.hip, alloc_lane_store)
  const uint32_t *xdepth;     // [n_sites] how far below 1 the site's smallest nonzero likelihood lies, -floor(log2), or >= 4096 for a site with
                              // a value that is no ordinary number in [0, 2) (ld_replay_lkl.hip, plain_depth; may be null: no site vouched for)
  uint32_t n_ind;
  int ignore_miss;
  ngsld_rec_std *out_std;     // the launch's records (device memory, or pinned host memory written in place)
  ngsld_rec_ext *out_ext;     // may be null
  int *status;
};

// ---- the lane-per-pair form (ld_replay_lkl.hip): one LANE per flagged pair, every lane walking the individuals in order ----
struct ReplayEntry {  // a flagged pair, located
  uint64_t slot;      // its record in the launch
  uint32_t s1, s2;
};
// individual-major copy of the exact store: xT[(i * n_sites + s) * 3 + g] = xplanes[s][g][i], sites [site_begin, site_end)
// xdepth (preset to zero, may be null), xperm (may be null): ReplayLklArgs::xdepth / xperm of the sites copied
hipError_t launch_transpose_store(const double *xplanes, uint64_t site_stride, uint32_t np, uint32_t n_ind, uint64_t n_sites,
                                  double *xT, uint32_t *xdepth, const uint32_t *xperm, hipStream_t stream, uint64_t site_begin = 0, uint64_t site_end = ~0ull);
// Walks the launch's bitmap (a thread per word), appends the pairs the lane-per-pair kernel takes to `list` (at most list_cap;
// counter: flags[4]) and CLEARS their bits: what stays set -- pairs whose Pearson moment is ill conditioned, pairs beyond the
// list's capacity -- is the wavefront-per-pair kernel's.
hipError_t launch_replay_expand(const ReplayLklArgs &a, ReplayEntry *list, uint64_t list_cap, hipStream_t stream);
// The list's entries ordered by (the pair's rarer site, its other site): keys_b / vals_b receive the sorted keys and the entries'
// indices in that order (ALL list_cap positions are sorted -- the list's length is known to the device only --, unused ones
// behind the real ones).  temp: replay_sort_temp_bytes(list_cap, n_sites) bytes.
size_t replay_sort_temp_bytes(uint64_t list_cap, uint32_t n_sites);
hipError_t launch_replay_sort(const ReplayLklArgs &a, const ReplayEntry *list, uint64_t list_cap, uint64_t *keys_a, uint64_t *keys_b,
                              uint32_t *vals_a, uint32_t *vals_b, void *temp, size_t temp_bytes, hipStream_t stream);
// the lane-per-pair kernel over list[order[0 .. flags[4])] (work counter: flags[5])
// waves_per_simd: 4 where the launch has pairs for every lane many times over; 1 for a launch of a few hundred thousand records (a
// lane's pair then takes a quarter of the time, and the launch lasts as long as its slowest lane)
hipError_t launch_replay_lanes(const ReplayLklArgs &a, const ReplayEntry *list, const uint32_t *order, const double *xT, int n_cus,
                               int waves_per_simd, hipStream_t stream);

// wavefronts per pair for a cohort of n_ind individuals (1 up to 512, then 2 / 4 / 8); 0: beyond the kernel (host replay)
uint32_t replay_lkl_waves(uint32_t n_ind);
// cohorts the wavefront-per-pair kernel does not take (replay_lkl_waves == 0): what the lanes left in the bitmap becomes the host's
hipError_t launch_replay_leftover(const ReplayLklArgs &a, hipStream_t stream);
// a persistent grid sized for n_cus compute units walks the bitmap
hipError_t launch_replay_lkl(const ReplayLklArgs &a, int n_cus, hipStream_t stream);

}  // namespace ngsld
