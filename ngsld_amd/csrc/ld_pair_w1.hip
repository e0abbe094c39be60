// ld_pair_w1.hip -- kernel selection by cohort size, instantiations of the one-wavefront-per-pair and lockstep kernels, and
// the launcher.
#include <cstdlib>

#include <algorithm>
#include <cstring>

#include "ld_kernel_run.h"
#include "ld_kernel_group.h"
#include "ld_kernel_stream.h"
#include "ld_dispatch.h"
#include "knobs.h"

namespace ngsld {

// n_ind -> kernel family and shape, by measurement (profiles/r03/sweep_513_1024.txt; profiles/r02b/sweep_nind.txt).
// Lane groups of 8 / 16 / 32 lanes x 8 slots cover 64 / 128 / 256 individuals (group kernel); one wavefront holds up to
// 10 * 64 = 640 individuals as 18 * 10 = 180 VGPRs of P (nine and ten slots spill a few registers OUTSIDE the EM loop and
// still beat two wavefronts of five by 54 % / 29 %: the per-iteration bookkeeping is paid once, nothing meets behind a
// barrier); up to 960 the a/b form on one wavefront; above that 2..8 wavefronts share the pair -- in the P form with up to
// eight slots per lane (nine or ten just past a doubling), or in the a/b form with nine to fifteen where that measured
// ahead -- and beyond 7,680 (5,120 where the a/b form has no shape) the streaming kernel takes over.
bool pair_config(uint64_t n_ind, PairConfig *cfg, int choice, bool masked) {
  if (n_ind == 0 || n_ind >= 0xffffffc0ull) return false;
  cfg->group = 64;
  cfg->waves = 1;
  cfg->form = 0;
  // Several wavefronts per pair in the a/b form (pair_ld_abm_kernel, ld_pair_ab.hip: the whole slice of the row vector in
  // registers, 12 registers per individual instead of the P form's 18) where it measured ahead, same box
  // (profiles/r03/sweep_abm.txt, sweep_abm2.txt, sweep_abm3.txt; pairs/s against the P form on twice the wavefronts, or against the streaming
  // kernel beyond 5,120):
  //   every individual counts   2 x 11..13 +10 %, 2 x 14 / 15 +3..4 %; 4 x 9 +1..2 %, 4 x 10 +10..11 %, 4 x 11..13 +25..40 %, 4 x 14 / 15
  //                             +29..33 %; 8 x 9 / 10 +19 %, 8 x 11..13 +58..65 %, 8 x 14 / 15 +47..73 %   (2 x 9 -10 %, 2 x 10 +1 %: the P form's)
  //   --ignore_miss_data        2 x 9 +4 %, 2 x 10 +16 %, 2 x 11..13 +6..15 %; 4 x 9 +25 %, 4 x 10 +37 %, 4 x 11..13 +28..45 %; 8 x 9 +38 %,
  //                             8 x 10 +56 %, 8 x 11..13 +35..69 %; once the pads stopped being hoisted out of the EM loop (sweep_abm_maskfix.txt:
  //                             13 slots another +12..15 %) 4 x 14 +24 %, 8 x 14 +39 %   (2 x 14 +-0; fifteen slots spill inside the loop: -40 %)
  //   (sixteen per lane spill ~480 bytes inside the EM loop: -30..-70 %, sweep_abm16.txt)
  // `masked` is what the matrix was set with (ngsld_set_geno_*): both forms compute either way, the layout follows this one.
  // NGSLD_PAIR_KERNEL=abm: wherever it has a shape (9..15 slots); =multi / =bres: never.
  if (choice == kChooseABMulti || choice == kChooseAuto) {
    for (int w = 2; w <= 8; w *= 2) {
      const uint64_t slots = (n_ind + (uint64_t)w * 64 - 1) / ((uint64_t)w * 64);
      const uint64_t lo = choice == kChooseABMulti ? 9 : (!masked && w == 2 ? 11 : 9);
      const uint64_t hi = choice == kChooseABMulti ? 15 : (masked ? (w == 2 ? 13 : 14) : 15);
      if (slots >= lo && slots <= hi) {
        cfg->kernel = kMulti;
        cfg->form = 1;
        cfg->waves = w;
        cfg->slots = (int)slots;
        cfg->np = (uint32_t)(slots * (uint64_t)w * 64);
        return true;
      }
    }
  }
  if (n_ind > 5120u) {  // beyond 8 wavefronts x 10 slots x 64 lanes: streaming kernel, one workgroup per pair
    cfg->kernel = kStream;
    cfg->waves = choice == kChoosePlainStream ? 4 : 8;  // 8: the candidate's vector resident (ld_pair_stream.hip)
    cfg->slots = 0;
    cfg->np = (uint32_t)((n_ind + 63) / 64 * 64);
    return true;
  }
  // 32-lane groups pay only where they pad less than 64 lanes do (an odd number of 32-individual slots: measured
  // +6..8 % at n_ind 160 / 200, -3 % at 250 where both shapes hold 256 individuals and lockstep is a pure loss)
  const bool g32 = n_ind > 128 && n_ind <= 256 && (((n_ind + 31) / 32) & 1ull);
  if (n_ind <= 128 || g32) {
    cfg->kernel = kGroup;
    cfg->group = n_ind <= 64 ? 8 : (n_ind <= 128 ? 16 : 32);
    cfg->slots = (int)((n_ind + (uint64_t)cfg->group - 1) / (uint64_t)cfg->group);
    cfg->np = (uint32_t)(cfg->slots * cfg->group);
    return true;
  }
  // one wavefront per pair: P form up to ten slots, a/b form up to fifteen (round 3, same box, pairs/s against two
  // wavefronts: 513 +54 %, 576 +54 %, 640 +29 %; with the row vector in registers, ld_pair_ab.hip and
  // profiles/r03/sweep_ab_range.txt: 641 +22 %, 704 +20 %, 768 +12 %, 832 +15 %, 896 +3.6 %, 960 +2.3 %; sixteen slots --
  // 961..1,024 -- are the two-wavefront kernel's: -5 % for the a/b form at 1,000)
  const uint64_t slots1 = (n_ind + 63) / 64;
  const bool want_ab = choice == kChooseAB && n_ind > 512 && n_ind <= 1024;
  if (!want_ab && (n_ind <= 512 || (choice == kChooseAuto && slots1 <= 10))) {
    cfg->kernel = kRun;
    cfg->slots = (int)slots1;
    cfg->np = (uint32_t)(slots1 * 64);
    return true;
  }
  if (want_ab || (choice == kChooseAuto && slots1 <= 15)) {
    cfg->kernel = kRunAB;
    cfg->slots = (int)slots1;
    cfg->np = (uint32_t)(slots1 * 64);
    return true;
  }
  // Several wavefronts per pair.  Just past a doubling of the wavefronts (1,025..1,280, 2,049..2,560, 4,097..5,120
  // individuals) NINE or TEN slots on half as many wavefronts beat five on twice as many: half-empty lanes and twice the
  // per-iteration bookkeeping against a few spilled registers outside the EM loop (nine: +35..46 % / +48..63 %; ten, round
  // 3, once empty slots no longer cost registers: +20 % at 1,153..1,280, +37 % at 2,305..2,560, and 4,609..5,120 stay on the
  // register kernels at 1.35e7 pairs/s against 6.2e6 for the streaming kernel; profiles/r03/sweep_bycount.txt).  Under
  // --ignore_miss_data 2 x 10 loses 2.6 % to 4 x 5 -- same plane layout, so the launcher takes that shape there (multi_shape).
  int w = 2;
  while ((n_ind + 64ull * w - 1) / (64ull * w) > 8) w *= 2;
  const uint64_t half = (n_ind + 32ull * w - 1) / (32ull * w);
  if (w >= 4 && (half == 9 || half == 10)) w /= 2;
  cfg->waves = w;
  cfg->slots = (int)((n_ind + 64ull * w - 1) / (64ull * w));
  cfg->np = (uint32_t)(cfg->slots * w * 64);
  cfg->kernel = kMulti;
  return true;
}

template <int G, int SLOTS>
static hipError_t launch_g(bool masked, const PairArgs &a, hipStream_t stream) {
  if (a.n_runs == 0) return hipSuccess;
  if (a.n_runs > 0x7fffffffull) return hipErrorInvalidValue;
  const dim3 grid((unsigned)a.n_runs), block(256);
  if (masked)
    hipLaunchKernelGGL((pair_ld_group_kernel<G, SLOTS, true>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((pair_ld_group_kernel<G, SLOTS, false>), grid, block, 0, stream, a);
  return hipGetLastError();
}

template <int G>
static hipError_t launch_group(int slots, bool masked, const PairArgs &a, hipStream_t stream) {
  switch (slots) {
    case 1: return launch_g<G, 1>(masked, a, stream);
    case 2: return launch_g<G, 2>(masked, a, stream);
    case 3: return launch_g<G, 3>(masked, a, stream);
    case 4: return launch_g<G, 4>(masked, a, stream);
    case 5: return launch_g<G, 5>(masked, a, stream);
    case 6: return launch_g<G, 6>(masked, a, stream);
    case 7: return launch_g<G, 7>(masked, a, stream);
    case 8: return launch_g<G, 8>(masked, a, stream);
    default: return hipErrorInvalidValue;
  }
}

template <int SLOTS>
static hipError_t launch_run(bool masked, const PairArgs &a, hipStream_t stream) {
  if (a.n_runs == 0) return hipSuccess;
  if (a.n_runs > 0x7fffffffull) return hipErrorInvalidValue;
  const dim3 grid((unsigned)a.n_runs), block(256);
  if (a.skip_degenerate && a.flags != nullptr) {  // (a matrix with degenerate sites: their pairs' EM is left to the replay)
    if (masked)
      hipLaunchKernelGGL((pair_ld_run_kernel<SLOTS, true, true>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((pair_ld_run_kernel<SLOTS, false, true>), grid, block, 0, stream, a);
    return hipGetLastError();
  }
  if (masked) {
    hipLaunchKernelGGL((pair_ld_run_kernel<SLOTS, true>), grid, block, 0, stream, a);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((pair_ld_run_kernel<SLOTS, false>), grid, block, 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_pair_wn(int slots, int waves, bool masked, const PairArgs &a, hipStream_t stream);

static hipError_t launch_pair_chunk(const PairConfig &cfg, bool masked, const PairArgs &a, hipStream_t stream);

// HIP addresses the threads of a launch with 32 bits per dimension: gridDim.x * blockDim.x has to stay below 2^32, and a
// grid beyond that is not refused -- it silently wraps (50,000 x 1,000 all pairs on ONE device is 7.8e7 items of the
// multi-wavefront kernel x 128 threads = 1.0e10: only the first 14 % of the items ran).  Workgroups have at most 512
// threads, so one launch takes at most 2^22 of them; longer item / run lists go out as consecutive launches on the
// same stream.  NGSLD_TEST_MAX_BLOCKS lowers the cap (tests).
hipError_t launch_pair_kernel(const PairConfig &cfg, bool masked, const PairArgs &a, hipStream_t stream) {
  uint64_t max_blocks = 1ull << 22;
  if (const char *e = test_knob("MAX_BLOCKS")) {
    const uint64_t u = std::strtoull(e, nullptr, 10);
    if (u >= 1 && u < max_blocks) max_blocks = u;
  }
  // Multi-wavefront kernel over long rows: tiled workgroup order (see pair_ld_kernel), one launch per group of up to 512
  // rows so that the rows of a launch have nearly the same number of items (ids beyond a row's count are empty workgroups).
  // (tests: NGSLD_TEST_TILES=0 keeps the plain item order.)  So do rows of fewer than 32 items (windowed runs: neighbouring rows
  // share their candidates anyway; tiled, configs[4]'s rows of 7-9 items lost 9 % to empty workgroups) and matrices that
  // fit the 256 MB Infinity Cache (nothing to gain: 12,000 x 1,000 all pairs -0.4 %).
  if (cfg.kernel == kMulti && cfg.waves > 1 && a.h_item_off != nullptr && a.item_off != nullptr && a.row1 > a.row0 &&
      !test_knob_is("TILES", "0")) {
    uint64_t longest = 0;
    for (uint32_t r = a.row0; r < a.row1; ++r) longest = std::max<uint64_t>(longest, a.h_item_off[r + 1] - a.h_item_off[r]);
    const uint64_t min_items = 32;
    uint64_t min_bytes = 256ull << 20;
    if (const char *e = test_knob("TILE_MIN_MB")) min_bytes = std::strtoull(e, nullptr, 10) << 20;  // tests
    if (longest >= min_items && a.planes_bytes >= min_bytes) {
      const uint32_t kTileRows = 64, kGroupRows = 512;  // (rows per tile, per launch group: 8 tiles)
      for (uint32_t g0 = a.row0; g0 < a.row1; g0 += kGroupRows) {
        const uint32_t g1 = std::min<uint32_t>(a.row1, g0 + kGroupRows);
        uint64_t most = 0;
        for (uint32_t r = g0; r < g1; ++r) most = std::max<uint64_t>(most, a.h_item_off[r + 1] - a.h_item_off[r]);
        if (most == 0) continue;
        PairArgs b = a;
        b.row0 = g0;
        b.row1 = g1;
        b.tile_rows = kTileRows;
        b.tile_nk = (uint32_t)((most + 7) / 8);
        const uint64_t row_blocks = (g1 - g0 + kTileRows - 1) / kTileRows;
        const uint64_t blocks = row_blocks * b.tile_nk * kTileRows * 8ull;
        if (blocks > max_blocks) {  // rows of more than 2^22 / 512 * 8 items (or the tests' cap): this group in plain order
          b.tile_nk = 0;
          const uint64_t first = a.h_item_off[g0], count = a.h_item_off[g1] - first;
          for (uint64_t off = 0; off < count; off += max_blocks) {
            b.items = a.items_all + first + off;
            b.n_items = std::min<uint64_t>(max_blocks, count - off);
            const hipError_t e = launch_pair_chunk(cfg, masked, b, stream);
            if (e != hipSuccess) return e;
          }
          continue;
        }
        b.n_items = blocks;  // the grid (launch_pair_wn launches n_items workgroups)
        const hipError_t e = launch_pair_chunk(cfg, masked, b, stream);
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    }
  }
  const bool by_runs = uses_runs(cfg.kernel);
  const uint64_t total = by_runs ? a.n_runs : a.n_items;
  for (uint64_t off = 0; off < total; off += max_blocks) {
    PairArgs b = a;
    const uint64_t n = total - off < max_blocks ? total - off : max_blocks;
    if (by_runs) {
      b.runs = a.runs + off;
      b.n_runs = n;
    } else {
      b.items = a.items + off;
      b.n_items = n;
    }
    const hipError_t e = launch_pair_chunk(cfg, masked, b, stream);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

static hipError_t launch_pair_chunk(const PairConfig &cfg, bool masked, const PairArgs &a, hipStream_t stream) {
  if (cfg.kernel == kStream) {
    if (a.n_items == 0) return hipSuccess;
    if (a.n_items > 0x7fffffffull) return hipErrorInvalidValue;
    if (cfg.waves == 8) return launch_pair_bres((int)((a.np / 64u + 7u) / 8u), masked, a, stream);
    if (masked)
      hipLaunchKernelGGL((pair_ld_stream_kernel<true>), dim3((unsigned)a.n_items), dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL((pair_ld_stream_kernel<false>), dim3((unsigned)a.n_items), dim3(256), 0, stream, a);
    return hipGetLastError();
  }
  if (cfg.kernel == kHard) return launch_pair_hard(masked, a, stream);
  if (effective_kernel(cfg, masked) == kRunAB) return launch_pair_ab(cfg.slots, masked, a, stream);
  if (cfg.kernel == kGroup) {
    if (cfg.group == 8) return launch_group<8>(cfg.slots, masked, a, stream);
    if (cfg.group == 16) return launch_group<16>(cfg.slots, masked, a, stream);
    return launch_group<32>(cfg.slots, masked, a, stream);
  }
  if (cfg.kernel == kMulti && cfg.form == 1) return launch_pair_abm(cfg.slots, cfg.waves, masked, a, stream);
  if (cfg.kernel == kMulti) {
    int slots = cfg.slots, waves = cfg.waves;
    multi_shape(cfg, masked, &slots, &waves);
    return launch_pair_wn(slots, waves, masked, a, stream);
  }
  switch (cfg.slots) {  // kRun
    case 1: return launch_run<1>(masked, a, stream);
    case 2: return launch_run<2>(masked, a, stream);
    case 3: return launch_run<3>(masked, a, stream);
    case 4: return launch_run<4>(masked, a, stream);
    case 5: return launch_run<5>(masked, a, stream);
    case 6: return launch_run<6>(masked, a, stream);
    case 7: return launch_run<7>(masked, a, stream);
    case 8: return launch_run<8>(masked, a, stream);
    case 9: return launch_run<9>(masked, a, stream);
    case 10: return launch_run<10>(masked, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ngsld
