// ld_device.h -- the pair-LD device code of gfx950 (MI355X, CDNA4), all of it, in order.  Replaces calc_pair_LD / haplo_freq /
// pair_freq_iter / pearson_r of the reference (ngsLD.cpp:229-367, shared/gen_func.cpp:1027-1119).  The translation units that
// instantiate kernels include the family headers they need; this umbrella is for tools and experiments.
#pragma once

#include "ld_common.h"          // constants, flag layout, PairArgs, cross-lane and LDS primitives
#include "ld_em.h"              // reciprocal tree, stage_pair, em_pair, write_pair
#include "ld_kernel_multi.h"    // pair_ld_kernel: 2 / 4 / 8 wavefronts per pair (P form)
#include "ld_run_pipeline.h"    // site copy into LDS, result ring, claim list of a run of items
#include "ld_kernel_run.h"      // pair_ld_run_kernel: one wavefront per pair (the headline)
#include "ld_kernel_group.h"    // pair_ld_group_kernel: several pairs per wavefront in lockstep
#include "ld_kernel_stream.h"   // pair_ld_stream_kernel, pair_ld_bres_kernel
#include "ld_dispatch.h"        // kernel families, PairConfig, launchers
