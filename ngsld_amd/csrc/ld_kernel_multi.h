// ld_kernel_multi.h -- pair_ld_kernel: 2, 4 or 8 wavefronts share one pair, P form (configs[3], configs[4]); instantiated in
// ld_pair_wn.hip.
#pragma once

#include "ld_em.h"

namespace ngsld {

// ---------------------------------------------------------------------------------------------
// Multi-wavefront kernel: WAVES = 2, 4 or 8 wavefronts share one pair (n_ind > 512, or whatever the one-wavefront kernels
// do not take).
//   SLOTS  individuals per lane (compile time, P lives in 18*SLOTS VGPRs)
//   MASKED --ignore_miss_data: individuals missing at either site are left out (gen_func.cpp:1089)
// Every wavefront only ever reads ITS slice of a site vector (individuals sub*SLOTS*64 ...), so the slice of the NEXT
// pair is copied global->LDS asynchronously into a wave-private 1536*SLOTS-byte buffer while the EM loop of the current
// pair runs; the row vector (same for the whole item, L2-hot) is read directly.  No extra barrier is needed for the
// prefetch.  (Without the prefetch -- every pair starting with an L2 / HBM round trip -- the kernel measured 12 % slower.)
// A cohort that does not fill all slots but the last (513 individuals on 2 x 5 slots: the second wavefront's fourth slot
// holds ONE individual, its fifth none) needs nothing special: empty slots are ghosts (stage_pair).
// SKIP: as in pair_ld_run_kernel -- a pair with a degenerate site (sc4[.][3]) is staged for its Pearson moment only, all
// wavefronts of the pair leave its EM out together, the NaN frequencies flag it and the exact-order replay is its one evaluation.
template <int SLOTS, int WAVES, bool MASKED, bool SKIP = false>
__global__ __launch_bounds__(WAVES * 64, 2) void pair_ld_kernel(PairArgs A) {
  static_assert(WAVES == 2 || WAVES == 4 || WAVES == 8, "pair_ld_kernel: 2, 4 or 8 wavefronts per pair");
  constexpr int kSliceBytes = SLOTS * 64 * 3 * 8;
  constexpr int kXchBase = WAVES * kSliceBytes;
  // kParked (every individual counts): the Pearson cross moment needs no meeting of the wavefronts before the EM loop --
  // each parks its partial sum per candidate, thread t adds them up when it writes the record -- and x is n_ind: one
  // barrier, one LDS round trip and one f64 division less per pair
  constexpr bool kParked = !MASKED;
  __shared__ __attribute__((aligned(16))) char smem[kXchBase + WAVES * 96 + 64 * sizeof(PairResult) +
                                                    (kParked ? 64 * WAVES * sizeof(double) : 0)];
  PairResult *res = reinterpret_cast<PairResult *>(smem + kXchBase + WAVES * 96);  // one per candidate
  double (*parked)[WAVES] = reinterpret_cast<double (*)[WAVES]>(smem + kXchBase + WAVES * 96 + 64 * sizeof(PairResult));
  double (*xch)[WAVES][4] = reinterpret_cast<double (*)[WAVES][4]>(smem + kXchBase);
  double (*xch0)[2] = reinterpret_cast<double (*)[2]>(smem + kXchBase + WAVES * 64);

  const int lane = threadIdx.x & 63;
  const int sub = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#ifdef NGSLD_PHASE_DELAY
  // Experiment (round 5, tools/ab_phase.sh; not in the product build): the review's idea for configs[4] -- a SIMD holds one
  // wavefront of each of the CU's two workgroups, and VALU sits idle when both are in the serial stretch of their iteration --
  // start one of the two half an iteration late.  Which of the two: the wavefront slot's parity (HW_ID bits 3:0).
  {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((4 - 1) << 11));
    if (__syncthreads_or((int)(hw & 1u))) __builtin_amdgcn_s_sleep(NGSLD_PHASE_DELAY);  // (x 64 cycles)
  }
#endif
  const Item *item_ptr;
  if (A.tile_nk != 0) {
    // Tiled order.  Workgroup ids go round the eight XCDs, so with tiles of tile_rows rows x 8 items, laid out row by row,
    // XCD x works on item column x of every row of the tile: the same ~64 + tile_rows candidate sites for tile_rows rows,
    // out of its own L2 -- in plain item order the workgroups in flight together are one row's whole candidate range, no
    // site is used twice while it is anywhere on the chip, and an all-pairs run streams the matrix from HBM once per row
    // (50,000 x 1,000: 2.1 TB/s, paid for in clock: the device is at its power limit).
    const uint32_t per = A.tile_rows * 8u;
    const uint32_t t = blockIdx.x / per, w = blockIdx.x % per;
    const uint32_t row = A.row0 + (t / A.tile_nk) * A.tile_rows + (w >> 3);
    // (the column an XCD takes rotates from tile to tile: the last tile of a row block is only partly filled, and with a fixed
    // assignment the XCDs of its first columns would carry all of it -- the dispatcher deals workgroup ids round robin, an
    // XCD cannot take over another's share: measured -18 % on rows of 7-9 items)
    const uint32_t k = (t % A.tile_nk) * 8u + ((w + t) & 7u);
    if (row >= A.row1) return;
    const uint64_t lo = A.item_off[row], hi = A.item_off[row + 1];
    if ((uint64_t)k >= hi - lo) return;
    item_ptr = A.items_all + lo + k;
  } else {
    if ((uint64_t)blockIdx.x >= A.n_items) return;
    item_ptr = A.items + blockIdx.x;
  }

  const Item it = *item_ptr;
  const uint32_t s1 = it.s1;
  const double m1 = A.maf[s1];
  const double mean1 = A.mean_e[s1];
  const double rsx1 = A.rsx[s1];
  const uint64_t rec0 = it.first_record - A.out_base;
  const double *pa = A.planes + (uint64_t)s1 * A.site_stride;
  const uint32_t i0 = (uint32_t)sub * (SLOTS * 64) + (uint32_t)lane;
  char *lds_b = smem + sub * kSliceBytes;
  // the scalars of the item's candidate sites come into LDS once, by one coalesced load per array, so the pair loop waits
  // for no ordinary global load (a ~2 us round trip per pair, and it would drain the slice copy in flight)
  __shared__ double site_sc[SKIP ? 4 : 3][64];
  const bool skip1 = SKIP && A.sc4[4 * (uint64_t)s1 + 3] != 0.0;
  if (threadIdx.x < it.count) {
    const uint32_t s2 = it.s2_begin + threadIdx.x;
    site_sc[0][threadIdx.x] = A.maf[s2];
    site_sc[1][threadIdx.x] = A.mean_e[s2];
    site_sc[2][threadIdx.x] = A.rsx[s2];
    if (SKIP) site_sc[SKIP ? 3 : 0][threadIdx.x] = A.sc4[4 * (uint64_t)s2 + 3];
  }
  __syncthreads();

  // copy this wavefront's slice of site s2 (three runs of SLOTS*512 B, one per genotype plane) into lds_b
  auto dma_slice = [&](uint32_t s2) {
    const char *g = reinterpret_cast<const char *>(A.planes + (uint64_t)s2 * A.site_stride + (uint32_t)sub * (SLOTS * 64)) +
                    lane * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int k = 0; k < (SLOTS * 512 + 1023) / 1024; ++k)
        if ((k + 1) * 1024 <= SLOTS * 512 || lane * 16 < SLOTS * 512 - k * 1024)
          __builtin_amdgcn_global_load_lds((glb_void_t *)(g + (size_t)pl * A.np * 8 + k * 1024),
                                           (lds_void_t *)(lds_b + pl * SLOTS * 512 + k * 1024), 16, 0, 0);
  };
  auto next_kept = [&](uint32_t c) -> uint32_t {  // first computed pair at or after c (ngsLD.cpp:270-282 filters)
    while (c < it.count && !((it.mask >> c) & 1ull)) ++c;
    return c;
  };

  // The wavefront's slice of the ROW vector is the same for all 64 candidates of the item: what fits beside P is loaded once,
  // relabelled (that depends on the row's frequency only) and kept in registers, and every pair is spared those loads from L2
  // -- and part of their round trip -- at its start.  Up to six slots per lane all of it fits: +6 % at 1,281..1,536 and
  // 2,561..3,072 individuals, +8..10 % under --ignore_miss_data; seven slots take three, eight slots two (three on two
  // wavefronts): configs[3] +1.2 % (+2.5 % masked), configs[4] +1.6 % (+3.9 %), eight wavefronts +1 %, same record bits
  // (profiles/r03/sweep_multi_aregs.txt, sweep_multi_aregs_8w.txt, ab_aregs78.txt, ab_aregs_final.txt).  Nine / ten slots:
  // none (they spill as it is).
  constexpr int kNA = SLOTS <= 6   ? SLOTS
                      : SLOTS == 7 ? (WAVES == 8 ? 2 : 3)
                      : SLOTS == 8 ? (WAVES == 2 ? 3 : (WAVES == 8 && MASKED ? 0 : 2))   // 8 x 8 masked would spill 72 B
                                   : 0;
  constexpr bool kARegs = kNA > 0;
  double a_regs[kARegs ? kNA : 1][3];
  if (kARegs) {
    const bool flip1 = m1 > 0.5;  // (relabel())
    const double *q0 = pa + (flip1 ? 2 * A.np : 0u), *q1 = pa + A.np, *q2 = pa + (flip1 ? 0u : 2 * A.np);
#pragma unroll
    for (int j = 0; j < kNA; ++j) {
      a_regs[j][0] = q0[i0 + 64u * (uint32_t)j]; a_regs[j][1] = q1[i0 + 64u * (uint32_t)j]; a_regs[j][2] = q2[i0 + 64u * (uint32_t)j];
    }
  }
  uint32_t c = next_kept(0);
  if (c < it.count) dma_slice(it.s2_begin + c);
  uint32_t xpar = 0;  // exchanges of this workgroup so far (see em_pair)
  while (c < it.count) {
    const uint32_t cn = next_kept(c + 1);
    const double m2 = site_sc[0][c], mean2 = site_sc[1][c], rsx2 = site_sc[2][c];
    double P[SLOTS][9];
    uint32_t vbits;
    double sxy;
    const Relabel rl = relabel(m1, m2, mean1, mean2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slice copied during the previous pair has landed
    stage_pair<SLOTS, MASKED, false, true, true>(pa, A.np, i0, reinterpret_cast<const double *>(lds_b),
                                                  (uint32_t)(SLOTS * 64), (uint32_t)lane, i0, A.n_ind, rl.mean1, rl.mean2, P,
                                                  vbits, sxy, rl.flip1, rl.flip2, kARegs ? a_regs : nullptr, kNA);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // a, b and the scalars are all consumed
    if (cn < it.count) dma_slice(it.s2_begin + cn);
    uint32_t x = count_valid<SLOTS>(vbits);
    // the cross moment comes uncentred (round 3: read off P, no bounds tests, a DPP-only reduction -- ~80 instructions less
    // per wavefront and pair than centring every element and folding with permlane swaps); n mean1 mean2 is taken off once
    sxy = wave_sum1_bcast(sxy);
    const double centre = (double)A.n_ind * rl.mean1 * rl.mean2;
    if (kParked) {
      if (lane == 0) lds_post(lds_addr(&parked[c][sub]), sxy);
      x = A.n_ind;  // (the ballots of the wavefronts add up to it: padding lanes are the only ones left out)
    } else {
      // (no barrier behind the reads: xch0 is written again a pair later, and every EM loop has a barrier of its own that
      // no wavefront passes before all have read these)
      const uint32_t base = lds_addr(&xch0[0][0]);
      if (lane == 0) lds_post2(base + (uint32_t)sub * 16u, sxy, (double)x);
      lds_barrier();
      dbl2 q[WAVES];
      lds_gather<WAVES>(base, q);
      double sx = 0.0, xs = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        sx += q[w][0];
        xs += q[w][1];
      }
      sxy = sx - centre;
      x = (uint32_t)xs;
    }
    double f0, f1, f2, f3;
    uint32_t n_iter = 0;
    if (SKIP && (skip1 || site_sc[SKIP ? 3 : 0][c] != 0.0)) {  // (the same in every wavefront of the pair)
      f0 = f1 = f2 = f3 = __builtin_nan("");
    } else {
      n_iter = em_pair<SLOTS, WAVES>(P, vbits, kParked ? A.inv_n : 1.0 / (double)x, rl.m1, rl.m2, f0, f1, f2, f3, xch, sub, lane,
                                     A.status, &xpar);
      unrelabel(rl.flip1, rl.flip2, f0, f1, f2, f3);
    }
    if (lane == 0 && sub == 0) {
      PairResult &r = res[c];
      r.f[0] = f0; r.f[1] = f1; r.f[2] = f2; r.f[3] = f3;
      r.sxy = kParked ? centre : sxy;  // (parked partial sums: the centring term travels in their place)
      r.rsx2 = rsx2;
      r.x = x;
      r.n_iter = n_iter;
    }
    c = cn;
  }
  // the whole workgroup shares the item: thread t derives and writes the record of candidate t
  if (kParked) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the parked partial sums are stores the compiler does not see
  __syncthreads();
  const uint32_t t = threadIdx.x;
  if (t < it.count && ((it.mask >> t) & 1ull)) {
    const PairResult r = res[t];
    double sxy = r.sxy;
    if (kParked) {
      sxy = 0.0;  // (the order the exchange added them in)
      for (int w = 0; w < WAVES; ++w) sxy += parked[t][w];
      sxy -= r.sxy;  // centred: sum e1 e2 - n mean1 mean2
    }
    write_pair(A, rec0 + (uint64_t)__popcll(it.mask & ((1ull << t) - 1ull)), r.f[0], r.f[1], r.f[2], r.f[3], sxy, rsx1,
               r.rsx2, r.x, r.n_iter);
  }
}

}  // namespace ngsld
