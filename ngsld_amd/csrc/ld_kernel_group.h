// ld_kernel_group.h -- pair_ld_group_kernel: 8 / 16 / 32 lanes per pair, several pairs of one row per wavefront in lockstep
// (configs[1]); instantiated in ld_pair_w1.hip.
#pragma once

#include "ld_group_reduce.h"
#include "ld_run_pipeline.h"

namespace ngsld {

// ---------------------------------------------------------------------------------------------
// Group kernel (n_ind <= 256): a group of G = 8, 16 or 32 lanes owns one pair, so a wavefront runs 8, 4 or 2 pairs in
// lockstep, all of one row s1.  With few individuals the per-iteration bookkeeping (f products, contraction,
// reduction, convergence test) outweighs the per-individual work; sharing each of those instructions between the
// pairs of a wavefront is worth more than the lanes lost to lockstep (a group that has converged idles until the
// slowest group of its wavefront has).
//   lane = G*grp + r;  individual of (lane, slot j) = G*j + r;  SLOTS = ceil(n_ind / G) <= 8;  np = G*SLOTS
//   LDS: [row vector a, linear][per wavefront: next-site buffers of its 64/G groups, interleaved in pieces of 16*G
//        bytes because global_load_lds writes wave-base + 16*lane: piece q of group gg sits at q*1024 + gg*16*G]
// Values that are wavefront-uniform in the 64-lane kernels (f, the f products, eps) are group-uniform VGPR values
// here; reductions are DPP steps inside the group, in a fixed order.
// ---------------------------------------------------------------------------------------------
template <int G, int SLOTS, bool MASKED>
__global__ __launch_bounds__(256, 2) void pair_ld_group_kernel(PairArgs A) {
  constexpr uint32_t kNp = SLOTS * G;
  constexpr int kSiteBytes = (int)kNp * 24;
  constexpr int kPiece = G * 16;                             // bytes one group moves per copy instruction
  constexpr int kPieces = (kSiteBytes + kPiece - 1) / kPiece;
  constexpr int kABytes = ((kSiteBytes + 1023) / 1024) * 1024;
  constexpr int kWaveBuf = (kPieces + 1) * 1024;             // the 64/G groups of a wavefront, interleaved, + their scalars
  constexpr int kGroups = 64 / G;
  constexpr uint32_t kRing = 32;
  constexpr int kRingOff = kABytes + 4 * kWaveBuf;
  constexpr int kListOff = kRingOff + 4 * (int)(kRing * sizeof(RunResult));
  constexpr unsigned long long kGroupMask = G == 32 ? 0xffffffffull : ((1ull << G) - 1ull);
  __shared__ __attribute__((aligned(16))) char smem[kListOff + sizeof(RunList)];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = lane / G, gl = lane % G;
  // Run form (see pair_ld_run_kernel): the workgroup works through up to kRunItems consecutive items of one row; the row
  // vector is loaded once, groups claim candidates from one counter for the whole run, item headers sit in LDS, a site's
  // scalars travel with its planes, results go through a wave-private ring.  With a pair costing a few microseconds at
  // these cohort sizes, one workgroup per 64-candidate item meant a workgroup turnover every ~10 us.
  const Run run = A.runs[blockIdx.x];
  const Item *g_items = A.items_all + run.first_item;
  const uint32_t s1 = g_items[0].s1;
  const double m1_in = A.sc4[4 * (uint64_t)s1], mean1_in = A.sc4[4 * (uint64_t)s1 + 1], rsx1 = A.sc4[4 * (uint64_t)s1 + 2];
  char *lds_a = smem;
  char *lds_w = smem + kABytes + wave * kWaveBuf;
  RunResult *ring = reinterpret_cast<RunResult *>(smem + kRingOff) + wave * kRing;
  RunList *L = reinterpret_cast<RunList *>(smem + kListOff);

  // one computed pair of the run per group: site and record index (group-uniform values)
  struct Cand {
    uint32_t s2;
    uint64_t rec;
    bool ok;
  };
  uint32_t n_kept = 0, s2_base = 0;   // set once the run's list is built
  uint64_t rec_base = 0;
  // a group claims the next computed pair of the run (the maf[s2] / sub-sampling filters, ngsLD.cpp:270-282, shaped the list)
  auto claim_group = [&]() -> Cand {
    uint32_t j = 0;
    if (gl == 0) j = atomicAdd(&L->claim, 1u);
    j = (uint32_t)__shfl((int)j, lane & ~(G - 1));
    if (j >= n_kept) return Cand{0u, 0ull, false};
    return Cand{s2_base + (uint32_t)L->cand[j], rec_base + j, true};
  };
  // byte offset of individual-slot j, genotype plane g of this lane's group inside the interleaved wave buffer
  auto b_off = [&](int g, int j) -> uint32_t {
    const uint32_t o = ((uint32_t)g * kNp + (uint32_t)j * (uint32_t)G + (uint32_t)gl) * 8u;  // offset inside the site
    return (o / (uint32_t)kPiece) * 1024u + (uint32_t)grp * (uint32_t)kPiece + (o % (uint32_t)kPiece);
  };
  // start the copy of every group's next site: lane (grp, r) moves the 16 bytes [q*kPiece + r*16, +16) of its group's
  // site for q = 0 .. kPieces-1; lanes r = 0, 1 then move the site's 32 bytes of scalars {maf, mean_e, rsx, 0}
  auto dma_groups = [&](const Cand &cd) {
    const char *g = reinterpret_cast<const char *>(A.planes + (uint64_t)(cd.ok ? cd.s2 : 0u) * A.site_stride) + gl * 16;
#pragma unroll
    for (int q = 0; q < kPieces; ++q)
      if (cd.ok && q * kPiece + gl * 16 < kSiteBytes)
        __builtin_amdgcn_global_load_lds((glb_void_t *)(g + q * kPiece), (lds_void_t *)(lds_w + q * 1024), 16, 0, 0);
    if (cd.ok && gl < 2)
      __builtin_amdgcn_global_load_lds((glb_void_t *)(reinterpret_cast<const char *>(A.sc4 + 4 * (uint64_t)cd.s2) + gl * 16),
                                       (lds_void_t *)(lds_w + kPieces * 1024), 16, 0, 0);
  };

  // the row vector: linear copy, 1 KiB per wave-instruction, chunks dealt round-robin to the four wavefronts
  {
    const char *g = reinterpret_cast<const char *>(A.planes + (uint64_t)s1 * A.site_stride) + lane * 16;
#pragma unroll
    for (int k = 0; k < kABytes / 1024; ++k)
      if ((k & 3) == wave && k * 1024 + lane * 16 < kSiteBytes)
        __builtin_amdgcn_global_load_lds((glb_void_t *)(g + k * 1024), (lds_void_t *)(lds_a + k * 1024), 16, 0, 0);
  }
  build_run_list(L, g_items, run.n_items);  // its last barrier: row vector and list in place
  n_kept = L->base[run.n_items];
  s2_base = L->items[0].s2_begin;
  rec_base = g_items[0].first_record - A.out_base;
  Cand cur = claim_group();
  dma_groups(cur);
  uint32_t held = 0;
  auto flush = [&](uint32_t n) {  // lane t derives and writes the record of ring entry t (holes: groups without a pair)
    if ((uint32_t)lane < n) {
      const RunResult r = ring[lane];
      if (r.rec != ~0ull) write_pair(A, r.rec, r.f[0], r.f[1], r.f[2], r.f[3], r.sxy, rsx1, r.rsx2, r.x, r.n_iter);
    }
  };

  while (__any(cur.ok)) {
    const bool active = cur.ok;
    Cand nxt = cur;
    if (active) nxt = claim_group();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the site copies issued a generation ago have landed
    const double *sc = reinterpret_cast<const double *>(lds_w + kPieces * 1024 + grp * kPiece);
    const double m2_in = active ? sc[0] : 0.5, mean2_in = active ? sc[1] : 0.0, rsx2 = active ? sc[2] : 0.0;
    // allele relabelling (see Relabel): site 1's is the same for the whole run, site 2's differs from group to group
    const Relabel rl = relabel(m1_in, m2_in, mean1_in, mean2_in);
    const double m1 = rl.m1, m2 = rl.m2, mean1 = rl.mean1, mean2 = rl.mean2;
    const int gb0 = rl.flip2 ? 2 : 0, gb2 = rl.flip2 ? 0 : 2;

    // ---- stage: P = a (x) b per lane, validity, Pearson cross moment (group sums) ----
    double P[SLOTS][9];
    uint32_t vbits = 0;
    double sxy = 0.0;
    const double *la = reinterpret_cast<const double *>(lds_a);
    const double *la0 = la + (rl.flip1 ? 2 * kNp : 0u), *la2 = la + (rl.flip1 ? 0u : 2 * kNp);
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      const uint32_t i = (uint32_t)j * (uint32_t)G + (uint32_t)gl;
      const double a0 = la0[i], a1 = la[kNp + i], a2 = la2[i];
      const double b0 = *reinterpret_cast<const double *>(lds_w + b_off(gb0, j));
      const double b1 = *reinterpret_cast<const double *>(lds_w + b_off(1, j));
      const double b2 = *reinterpret_cast<const double *>(lds_w + b_off(gb2, j));
      const bool inb = i < A.n_ind;
      bool ok = inb && active;
      if (MASKED) ok = ok && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);  // gen_func.cpp:1089
      vbits |= (ok ? 1u : 0u) << j;
      double z0 = a0, z1 = a1, z2 = a2;
      if (MASKED) {  // an individual without data is a ghost, P = (1, 0, ..., 0): see stage_pair
        const double keep = ok ? 1.0 : 0.0;
        z0 = a0 * keep; z1 = a1 * keep; z2 = a2 * keep;
        P[j][0] = fma(z0, b0, 1.0 - keep);
      } else if (j == SLOTS - 1) {  // padding lanes (zeros in the planes) of the last slot
        P[j][0] = fma(a0, b0, inb ? 0.0 : 1.0);
      } else {
        P[j][0] = a0 * b0;
      }
      P[j][1] = z0 * b1; P[j][2] = z0 * b2;
      P[j][3] = z1 * b0; P[j][4] = z1 * b1; P[j][5] = z1 * b2;
      P[j][6] = z2 * b0; P[j][7] = z2 * b1; P[j][8] = z2 * b2;
      // expected genotypes p1 + 2 p2 (ngsLD.cpp:113); pearson_r runs over ALL individuals (ngsLD.cpp:290); the planes
      // hold zeros beyond n_ind, so the uncentred cross moment needs no bounds test
      if (!MASKED)  // (a1 + 2 a2)(b1 + 2 b2) = P4 + 2 P5 + 2 P7 + 4 P8
        sxy += fma(4.0, P[j][8], fma(2.0, P[j][5] + P[j][7], P[j][4]));
      else          // P of individuals without data is zeroed: take the moment from a and b
        sxy = fma(fma(2.0, a2, a1), fma(2.0, b2, b1), sxy);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // buffers consumed: start the next generation's copies
    dma_groups(nxt);
    // individuals with data in this group's pair (gen_func.cpp:1091), integer exact: everybody without --ignore_miss_data
    // (then 1/x comes precomputed -- the same IEEE quotient -- instead of a ~35-instruction f64 division per generation)
    uint32_t x = A.n_ind;
    if (MASKED) {
      x = 0;
#pragma unroll
      for (int j = 0; j < SLOTS; ++j)
        x += (uint32_t)__popcll((__ballot((vbits >> j) & 1u) >> (grp * G)) & kGroupMask);
    }
    // centred once per pair: sum e1 e2 - n mean1 mean2 (as the run kernel)
    sxy = fma(-(double)A.n_ind * mean1, mean2, group_sum<G>(sxy));

    // ---- haplo_freq (gen_func.cpp:1027-1059), 64/G pairs in lockstep ----
    double f0 = (1 - m1) * (1 - m2), f1 = (1 - m1) * m2, f2 = m1 * (1 - m2), f3 = m1 * m2;
    if (active && (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1)) {
      if (gl == 0) atomicExch(A.status, (int)NGSLD_ERR_MAF_RANGE);
      f0 = f1 = f2 = f3 = __builtin_nan("");
    }
    const double inv_x = MASKED ? 1.0 / (double)x : A.inv_n;
    // one reciprocal per lane and iteration (RcpTree), empty slots are ghosts: see em_pair
    constexpr bool kTree = SLOTS > 1;
    auto em_step = [&](auto tree_tag, double &n0, double &n1, double &n2, double &n3) {
      constexpr bool kT = decltype(tree_tag)::value;
      constexpr bool kDrop = kT;  // shared-reciprocal step: three-value form; the other one: full (see em_pair)
      const double p00 = f0 * f0, p01 = f0 * f1, p02 = f0 * f2, p03 = f0 * f3, p11 = f1 * f1;
      const double p12 = f1 * f2, p13 = f1 * f3, p22 = f2 * f2, p23 = f2 * f3, p33 = f3 * f3;
      const double w1 = p01 + p01, w3 = p02 + p02, w4 = 2.0 * (p03 + p12), w5 = p13 + p13, w7 = p23 + p23;
      double R0 = 0, R1 = 0, R2 = 0, R3 = 0, R4 = 0, R5 = 0, R6 = 0, R7 = 0, R8 = 0;
      auto slot_s = [&](int j) -> double {
        double s = p00 * P[j][0];
        s = fma(w1, P[j][1], s); s = fma(p11, P[j][2], s);
        s = fma(w3, P[j][3], s); s = fma(w4, P[j][4], s); s = fma(w5, P[j][5], s);
        s = fma(p22, P[j][6], s); s = fma(w7, P[j][7], s); s = fma(p33, P[j][8], s);
        return s;
      };
      auto slot_acc = [&](int j, double r) {
        if (!kDrop) R0 = fma(P[j][0], r, R0);
        R1 = fma(P[j][1], r, R1); R2 = fma(P[j][2], r, R2);
        R3 = fma(P[j][3], r, R3); R4 = fma(P[j][4], r, R4); R5 = fma(P[j][5], r, R5);
        R6 = fma(P[j][6], r, R6); R7 = fma(P[j][7], r, R7); R8 = fma(P[j][8], r, R8);
      };
      if constexpr (kT) {
        double sv[SLOTS], rv[SLOTS];
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(0);  // dense stretch: see em_pair
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) sv[j] = slot_s(j);
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(3);
        // 1/x rides on the root inverse (as in em_pair): every R, and with them the three t_k, come out divided by x
        RcpTree<SLOTS>::down(sv, rcp_refined(RcpTree<SLOTS>::prod(sv)) * inv_x, rv);
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) slot_acc(j, rv[j]);
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(3);
      } else {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
          // without --ignore_miss_data only the last slot can hold padding lanes; a group without a pair computes
          // on stale buffers there, which is harmless (it is `done` from the start and never written)
          if ((!MASKED && j < SLOTS - 1) || ((vbits >> j) & 1u)) slot_acc(j, rcp_refined(slot_s(j)));
        }
      }
      double t1 = fma(p13, R5, fma(p12, R4, fma(p11, R2, p01 * R1)));
      double t2 = fma(p23, R7, fma(p22, R6, fma(p12, R4, p02 * R3)));
      double t3 = fma(p33, R8, fma(p23, R7, fma(p13, R5, p03 * R4)));
      group_sum3<G>(t1, t2, t3, (lane & 1) != 0, (lane & 2) != 0);
      if (kT) {
        n1 = t1; n2 = t2; n3 = t3;  // already divided by x
      } else {
        n1 = t1 * inv_x; n2 = t2 * inv_x; n3 = t3 * inv_x;
      }
      if (kDrop) {  // the first frequency is what the other three leave (see em_pair)
        n0 = 1.0 - ((n1 + n2) + n3);
      } else {
        const double t0 = group_sum<G>(fma(p03, R4, fma(p02, R3, fma(p01, R1, p00 * R0))));
        n0 = kT ? t0 : t0 * inv_x;
      }
    };
    bool done = !active, tie = false;
    uint32_t n_iter = (uint32_t)kIterMax;
    constexpr bool kMaskDone = SLOTS >= NGSLD_MASK_SLOTS;
    // As in em_pair: the hot loop holds the shared-reciprocal step in its three-value form only; a step that is not sane
    // in some live group leaves it for one iteration with a reciprocal per individual, and as soon as hap 0 of any live
    // group falls below kFullBelow the wavefront leaves it for good and finishes in the full four-value form.
    constexpr double kFullBelow = 0x1p-10;
    bool full = __any(!done && f0 < kFullBelow);
    uint32_t itn = 0;
    while (itn < (uint32_t)kIterMax) {
      if (kTree && !full) {
        bool all_done = false;
        for (; itn < (uint32_t)kIterMax; ++itn) {
          double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
          // A group that has converged idles until the slowest group of its wavefront has.  It idles with its lanes
          // SWITCHED OFF (EXEC), not computing on stale values: the device runs these kernels at its power limit
          // (1.35 kW, 2.05-2.1 GHz of 2.4: profiles/r02/clocks_power_r02.txt), so what idle lanes do not burn comes
          // back as clock.  (Every cross-lane step of the EM stays inside a group, all of whose lanes are on or off.)
          // Same-box A/B (tools/ab_mask.sh): +5.5 % at n_ind 100 (configs[1]), +6.9 % at 64, +2 % at 48, +1 % at 200 (two
          // groups: little to idle); -2.5 % at 24 / 32 / 40 and in the genotype-combination kernel, whose iterations are
          // too short for the mask's own bookkeeping -- hence only from NGSLD_MASK_SLOTS individuals per lane on.
          if (!kMaskDone || !done) em_step(PairedTag(), n0, n1, n2, n3);
          if (__any(!done && !(n1 < 2.0))) break;  // an odd step (one NaN reciprocal poisons every R, see em_pair)
          // as in em_pair: eps is at least the change of hap 1, and while that alone is above EPSILON in every live
          // group the other three differences are not formed
          if (__any(!done && fabs(n1 - f1) < kEpsilonTie)) {
            const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
            if (!done && fabs(eps - kEpsilon) < kTieMargin) tie = true;  // too close to call: replayed
            if (!done && eps < kEpsilon) {  // gen_func.cpp:1054-1055
              done = true;
              n_iter = itn;
              f0 = n0; f1 = n1; f2 = n2; f3 = n3;
            }
          }
          if (!done) {
            f0 = n0; f1 = n1; f2 = n2; f3 = n3;
          }
          if (__all(done)) {
            all_done = true;
            break;
          }
          if (__any(!done && f0 < kFullBelow)) {
            full = true;
            ++itn;  // this iteration is complete
            break;
          }
        }
        if (all_done || itn >= (uint32_t)kIterMax) break;
        if (full) continue;
      }
      double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
      if (!kMaskDone || !done) em_step(SingleTag(), n0, n1, n2, n3);
      const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
      if (!done) {
        if (!(n1 < 2.0)) {  // the reference's all-NaN step: "converges" at this iteration (see em_pair)
          f0 = f1 = f2 = f3 = __builtin_nan("");
          done = true;
          n_iter = itn;
        } else {
          f0 = n0; f1 = n1; f2 = n2; f3 = n3;
          if (fabs(eps - kEpsilon) < kTieMargin) tie = true;
          if (eps < kEpsilon) {
            done = true;
            n_iter = itn;
          }
        }
      }
      if (__all(done)) break;
      ++itn;
    }

    unrelabel(rl.flip1, rl.flip2, f0, f1, f2, f3);
    if (gl == 0) {  // one ring entry per group and generation; a group without a pair leaves a hole
      RunResult &r = ring[held + (uint32_t)grp];
      r.f[0] = f0; r.f[1] = f1; r.f[2] = f2; r.f[3] = f3;
      r.sxy = sxy;
      r.rsx2 = rsx2;
      r.x = x;
      r.n_iter = n_iter | (tie ? kTieBit : 0u);
      r.rec = active ? cur.rec : ~0ull;
    }
    held += (uint32_t)kGroups;
    if (held + (uint32_t)kGroups > kRing) {
      flush(held);
      held = 0;
    }
    cur = nxt;
  }
  flush(held);
}

}  // namespace ngsld
