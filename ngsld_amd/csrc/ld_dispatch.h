// ld_dispatch.h -- host side of the pair kernels: the kernel families, PairConfig (pair_config picks by cohort size, by
// measurement), the launchers the engine calls.
#pragma once

#include "ld_common.h"

namespace ngsld {

constexpr int kBresMinSlots = 11, kBresMaxSlots = 20;  // 8 wavefronts x 64 lanes x 11..20 blocks: 5,121 .. 10,240 individuals
constexpr int kBresTailSlots = 20;                      // beyond: 20 blocks per wavefront resident (10,240 individuals), the rest streamed

// host-callable launchers, defined in ld_pair_w1.hip / ld_pair_wn.hip
// Kernel families (pair_config picks by cohort size, by measurement: profiles/r03/sweep_513_1024.txt):
//   kGroup  8 / 16 / 32 lanes per pair, several pairs per wavefront in lockstep (n_ind <= 128, some shapes up to 224)
//   kRun    one wavefront per pair, the row vector shared in LDS, runs of items (n_ind <= 640: up to TEN individuals per lane)
//   kRunAB  one wavefront per pair, EM step in its a/b form, run pipeline (ld_pair_ab.hip: 641..960)
//   kMulti  2 / 4 / 8 wavefronts per pair: P form (pair_ld_kernel, 5..10 per lane, 961..5,120) or a/b form (pair_ld_abm_kernel,
//           9..15 per lane with the row slice in registers: most of 1,281..7,680 -- pair_config has the table)
//   kStream any n_ind: the candidate's vector (its first 10,240 individuals beyond that many) in registers, the row vector -- or,
//           with cfg.waves == 4 (NGSLD_PAIR_KERNEL=stream), both -- re-read every iteration
//   kHard   every likelihood triple of the matrix is a called genotype or "no data": the pairs' 16 genotype-combination
//           counts replace the individuals (any n_ind up to kHardMaxInd)
enum PairKernel { kGroup = 0, kMulti = 2, kStream = 4, kRun = 5, kHard = 6, kRunAB = 7 };
// NGSLD_PAIR_KERNEL=multi | ab | stream (tests, A/B): the multi-wavefront kernel from 513 individuals on / the a/b kernel for
// 513..1024 / the plain streaming kernel (nothing resident) beyond 5,120; abm | bres: several wavefronts per pair in the a/b
// form wherever it has a shape / never (P form up to 5,120, the streaming kernel with the candidate's vector resident beyond)
enum PairChoice { kChooseAuto = 0, kChooseMulti = 1, kChooseAB = 2, kChoosePlainStream = 3, kChooseABMulti = 4, kChooseResidentStream = 5 };
// kernels launched over runs of items (one workgroup per run, candidates addressed as 64 * item + offset)
inline bool uses_runs(int kernel) { return kernel == kRun || kernel == kGroup || kernel == kHard || kernel == kRunAB; }
constexpr uint32_t kHardMaxWords = 512;                 // row bit sets in LDS: 4 x 512 x 8 B = 16 KB
constexpr uint64_t kHardMaxInd = 64ull * kHardMaxWords;
struct PairConfig {
  int kernel;   // PairKernel
  int group;    // kGroup: lanes per pair (8, 16 or 32); 64 otherwise
  int slots;    // individuals per lane
  int waves;    // wavefronts per pair
  int form;     // kMulti: 0 = P form (pair_ld_kernel), 1 = a/b form (pair_ld_abm_kernel, ld_pair_ab.hip)
  uint32_t np;  // padded individuals per genotype plane
};
bool pair_config(uint64_t n_ind, PairConfig *cfg, int choice = kChooseAuto, bool masked = false);
// the kernel a launch really takes (a hook for families that differ with --ignore_miss_data; none does at present)
inline int effective_kernel(const PairConfig &cfg, bool /*masked*/) { return cfg.kernel; }
// ... and the shape of the multi-wavefront kernel: 2 x 10 slots (1,153..1,280 individuals) run as 4 x 5 under
// --ignore_miss_data (measured -2.6 % otherwise; both shapes read the same planes: np = 1,280)
inline void multi_shape(const PairConfig &cfg, bool masked, int *slots, int *waves) {
  *slots = cfg.slots;
  *waves = cfg.waves;
  if (masked && cfg.waves == 2 && cfg.slots == 10) {
    *slots = 5;
    *waves = 4;
  }
}
hipError_t launch_pair_kernel(const PairConfig &cfg, bool masked, const PairArgs &args, hipStream_t stream);
hipError_t launch_pair_hard(bool masked, const PairArgs &args, hipStream_t stream);  // ld_pair_hard.hip
hipError_t launch_pair_ab(int slots, bool masked, const PairArgs &args, hipStream_t stream);  // ld_pair_ab.hip
hipError_t launch_pair_abm(int slots, int waves, bool masked, const PairArgs &args, hipStream_t stream);  // ld_pair_ab.hip
hipError_t launch_pair_bres(int slots, bool masked, const PairArgs &args, hipStream_t stream);  // ld_pair_stream.hip
// Per-site classification behind kHard (ld_pair_hard.hip): masks / u as in PairArgs; *all_hard (device int, preset to 1) is
// cleared when any triple is neither a called genotype (1,0,0) / (0,1,0) / (0,0,1) nor three equal values
hipError_t launch_classify_hard(const double *planes, uint64_t site_stride, uint32_t np, uint32_t n_ind, uint64_t n_sites,
                                uint64_t *masks, double *u, int *all_hard, hipStream_t stream);
// candidate s2 sites per work item
inline uint32_t item_span(const PairConfig &cfg, uint32_t pairs_per_item) {
  if (uses_runs(cfg.kernel)) return 64u;  // run form: candidates are addressed as 64 * item + offset
  // (multi-wavefront kernel: one workgroup works through the item pair by pair; 64 candidates per item instead of 16 means a
  // quarter of the workgroups and of the per-item scalar loads: -1.3 % kernel time at n_ind 1000, -4.0 % at 2000)
  const uint32_t span = cfg.kernel == kMulti ? 4u * pairs_per_item : pairs_per_item;
  return span > 64u ? 64u : span;
}

}  // namespace ngsld
