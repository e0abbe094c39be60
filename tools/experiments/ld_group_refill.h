// ld_group_refill.h -- EXPERIMENT THAT LOST (round 4; not part of the product build): the group kernel
// (pair_ld_group_kernel, ld_device.h) with idle groups REFILLED before the slowest group of the wavefront has converged.
// Measured on one box, configs[1] (profiles/r04/ab_group_refill.txt): lockstep generations 19.3 ms (6.42e8 pairs/s); a refill
// as soon as 1 / 2 / 3 groups are idle 60.4 / 39.8 / 33.2 ms (-68 / -51 / -42 %), records equal on all 12,497,500 pairs.  The
// instruction-count simulation had said -9 / -4 / -10 % (round 2's: +3 %): the staging pass, run with a quarter to three
// quarters of the lanes on and everything a live group carries held across it, spills 368 bytes per lane.
// To reproduce: in ld_pair_w1.hip include this file and launch pair_ld_group_refill_kernel<G, SLOTS, masked, THRESH> from
// launch_g instead of pair_ld_group_kernel<G, SLOTS, masked> (same arguments, same grid).
//
// In the shipping kernel a wavefront works in generations: its 64 / G groups stage a pair each, iterate in lockstep until the
// last of them has converged (converged groups sit out with their lanes off), write their results, stage the next four.  The
// lockstep tail is 1.16 x the pairs' own iterations at n_ind 100 (configs[1]).  Here a generation ends as soon as
// NGSLD_GROUP_REFILL groups that have another pair waiting are idle: those stage their next pair (the staging code runs with
// only their lanes on), the others carry P, f and their own iteration counter across.  Same arithmetic per pair, same records.
// What it costs: the staging pass (~1.2 iterations' worth of instructions) runs once per refill instead of once per 64 / G
// pairs, an iteration counter per group, one more ballot per iteration.
#ifndef NGSLD_LD_GROUP_REFILL_H
#define NGSLD_LD_GROUP_REFILL_H

#include "../../ngsld_amd/csrc/ld_device.h"

namespace ngsld {

template <int G, int SLOTS, bool MASKED, int THRESH>
__global__ __launch_bounds__(256, 2) void pair_ld_group_refill_kernel(PairArgs A) {
  constexpr uint32_t kNp = SLOTS * G;
  constexpr int kSiteBytes = (int)kNp * 24;
  constexpr int kPiece = G * 16;
  constexpr int kPieces = (kSiteBytes + kPiece - 1) / kPiece;
  constexpr int kABytes = ((kSiteBytes + 1023) / 1024) * 1024;
  constexpr int kWaveBuf = (kPieces + 1) * 1024;
  constexpr int kGroups = 64 / G;
  constexpr uint32_t kRing = 32;
  constexpr uint32_t kPer = kRing / kGroups;  // ring entries per group
  constexpr int kRingOff = kABytes + 4 * kWaveBuf;
  constexpr int kListOff = kRingOff + 4 * (int)(kRing * sizeof(RunResult));
  constexpr unsigned long long kGroupMask = G == 32 ? 0xffffffffull : ((1ull << G) - 1ull);
  __shared__ __attribute__((aligned(16))) char smem[kListOff + sizeof(RunList)];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = lane / G, gl = lane % G;
  const Run run = A.runs[blockIdx.x];
  const Item *g_items = A.items_all + run.first_item;
  const uint32_t s1 = g_items[0].s1;
  const double m1_in = A.sc4[4 * (uint64_t)s1], mean1_in = A.sc4[4 * (uint64_t)s1 + 1], rsx1 = A.sc4[4 * (uint64_t)s1 + 2];
  char *lds_a = smem;
  char *lds_w = smem + kABytes + wave * kWaveBuf;
  RunResult *ring = reinterpret_cast<RunResult *>(smem + kRingOff) + wave * kRing;
  RunList *L = reinterpret_cast<RunList *>(smem + kListOff);

  struct Cand {
    uint32_t s2;
    uint64_t rec;
    bool ok;
  };
  uint32_t n_kept = 0, s2_base = 0;
  uint64_t rec_base = 0;
  auto claim_group = [&]() -> Cand {
    uint32_t j = 0;
    if (gl == 0) j = atomicAdd(&L->claim, 1u);
    j = (uint32_t)__shfl((int)j, lane & ~(G - 1));
    if (j >= n_kept) return Cand{0u, 0ull, false};
    return Cand{s2_base + (uint32_t)L->cand[j], rec_base + j, true};
  };
  auto b_off = [&](int g, int j) -> uint32_t {
    const uint32_t o = ((uint32_t)g * kNp + (uint32_t)j * (uint32_t)G + (uint32_t)gl) * 8u;
    return (o / (uint32_t)kPiece) * 1024u + (uint32_t)grp * (uint32_t)kPiece + (o % (uint32_t)kPiece);
  };
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
  {
    const char *g = reinterpret_cast<const char *>(A.planes + (uint64_t)s1 * A.site_stride) + lane * 16;
#pragma unroll
    for (int k = 0; k < kABytes / 1024; ++k)
      if ((k & 3) == wave && k * 1024 + lane * 16 < kSiteBytes)
        __builtin_amdgcn_global_load_lds((glb_void_t *)(g + k * 1024), (lds_void_t *)(lds_a + k * 1024), 16, 0, 0);
  }
  if ((uint32_t)lane < kRing) ring[lane].rec = ~0ull;  // (wave-private ring: an entry is a hole until a group fills it)
  build_run_list(L, g_items, run.n_items);
  n_kept = L->base[run.n_items];
  s2_base = L->items[0].s2_begin;
  rec_base = g_items[0].first_record - A.out_base;
  auto flush = [&]() {
    if ((uint32_t)lane < kRing) {
      const RunResult r = ring[lane];
      if (r.rec != ~0ull && r.n_iter != 0xffffffffu) {
        write_pair(A, r.rec, r.f[0], r.f[1], r.f[2], r.f[3], r.sxy, rsx1, r.rsx2, r.x, r.n_iter);
        ring[lane].rec = ~0ull;
      }
    }
  };

  // per-group state, carried across refills
  // (register budget: the shipping kernel sits at 252 VGPRs.  What a pair needs only when its record is written -- record
  // index, cross moment, rsx, sample size -- goes into the group's ring entry when the pair is STAGED, not carried; the pair
  // waiting in the group's buffer is carried as its index in the run's list alone.)
  uint32_t pend_j;  // list index of the pair whose site is in flight to (or waiting in) the group's buffer; >= n_kept: none
  {
    const Cand c0 = claim_group();
    pend_j = c0.ok ? (uint32_t)(c0.rec - rec_base) : 0xffffffffu;
    dma_groups(c0);
  }
  bool live = false, done = true, tie = false, flip2 = false;
  const bool flip1 = m1_in > 0.5;
  double P[SLOTS][9];
  uint32_t vbits = 0, it_g = 0, n_iter = (uint32_t)kIterMax, held = 0;
  double inv_x = A.inv_n;
  double f0 = 0.25, f1 = 0.25, f2 = 0.25, f3 = 0.25;
#pragma unroll
  for (int j = 0; j < SLOTS; ++j)
#pragma unroll
    for (int q = 0; q < 9; ++q) P[j][q] = q == 0 ? 1.0 : 0.0;  // (a defined value for groups that never get a pair)

  constexpr bool kTree = SLOTS > 1;
  constexpr bool kMaskDone = SLOTS >= NGSLD_MASK_SLOTS;
  constexpr double kFullBelow = 0x1p-10;

  for (;;) {
    const bool need = !live && pend_j < n_kept;
    const bool any_need = __any(need);
    if (!any_need && !__any(live)) break;
    if (any_need) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the site copies in flight have landed
      if (need) {
        const double *sc = reinterpret_cast<const double *>(lds_w + kPieces * 1024 + grp * kPiece);
        const double m2_in = sc[0], mean2_in = sc[1], rsx2 = sc[2];
        const Relabel rl = relabel(m1_in, m2_in, mean1_in, mean2_in);
        flip2 = rl.flip2;
        const int gb0 = rl.flip2 ? 2 : 0, gb2 = rl.flip2 ? 0 : 2;
        vbits = 0;
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
          bool ok = inb;
          if (MASKED) ok = ok && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);
          vbits |= (ok ? 1u : 0u) << j;
          double z0 = a0, z1 = a1, z2 = a2;
          if (MASKED) {
            const double keep = ok ? 1.0 : 0.0;
            z0 = a0 * keep; z1 = a1 * keep; z2 = a2 * keep;
            P[j][0] = fma(z0, b0, 1.0 - keep);
          } else if (j == SLOTS - 1) {
            P[j][0] = fma(a0, b0, inb ? 0.0 : 1.0);
          } else {
            P[j][0] = a0 * b0;
          }
          P[j][1] = z0 * b1; P[j][2] = z0 * b2;
          P[j][3] = z1 * b0; P[j][4] = z1 * b1; P[j][5] = z1 * b2;
          P[j][6] = z2 * b0; P[j][7] = z2 * b1; P[j][8] = z2 * b2;
          if (!MASKED)
            sxy += fma(4.0, P[j][8], fma(2.0, P[j][5] + P[j][7], P[j][4]));
          else
            sxy = fma(fma(2.0, a2, a1), fma(2.0, b2, b1), sxy);
        }
        uint32_t x = A.n_ind;
        if (MASKED) {
          x = 0;
#pragma unroll
          for (int j = 0; j < SLOTS; ++j)
            x += (uint32_t)__popcll((__ballot((vbits >> j) & 1u) >> (grp * G)) & kGroupMask);
        }
        sxy = fma(-(double)A.n_ind * rl.mean1, rl.mean2, group_sum<G>(sxy));
        const double m1 = rl.m1, m2 = rl.m2;
        f0 = (1 - m1) * (1 - m2); f1 = (1 - m1) * m2; f2 = m1 * (1 - m2); f3 = m1 * m2;
        if (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1) {
          if (gl == 0) atomicExch(A.status, (int)NGSLD_ERR_MAF_RANGE);
          f0 = f1 = f2 = f3 = __builtin_nan("");
        }
        if (MASKED) inv_x = 1.0 / (double)x;
        if (gl == 0) {  // what the record needs beyond the frequencies, parked in the group's ring entry
          RunResult &r = ring[(uint32_t)grp * kPer + held];
          r.sxy = sxy;
          r.rsx2 = rsx2;
          r.x = x;
          r.rec = rec_base + pend_j;
          r.n_iter = 0xffffffffu;  // (not finished: flush leaves it alone)
        }
        it_g = 0;
        n_iter = (uint32_t)kIterMax;
        tie = false;
        done = false;
        live = true;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // buffers consumed: the groups that staged claim and copy their next
      Cand nx{0u, 0ull, false};
      if (need) nx = claim_group();
      dma_groups(nx);
      if (need) pend_j = nx.ok ? (uint32_t)(nx.rec - rec_base) : 0xffffffffu;
    }

    // ---- EM iterations of the live groups until enough of them are idle with a pair waiting (or none is live) ----
    auto em_step = [&](auto tree_tag, double &n0, double &n1, double &n2, double &n3) {
      constexpr bool kT = decltype(tree_tag)::value;
      constexpr bool kDrop = kT;
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
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) sv[j] = slot_s(j);
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(3);
        RcpTree<SLOTS>::down(sv, rcp_refined(RcpTree<SLOTS>::prod(sv)) * inv_x, rv);
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) slot_acc(j, rv[j]);
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(3);
      } else {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
          if ((!MASKED && j < SLOTS - 1) || ((vbits >> j) & 1u)) slot_acc(j, rcp_refined(slot_s(j)));
        }
      }
      double t1 = fma(p13, R5, fma(p12, R4, fma(p11, R2, p01 * R1)));
      double t2 = fma(p23, R7, fma(p22, R6, fma(p12, R4, p02 * R3)));
      double t3 = fma(p33, R8, fma(p23, R7, fma(p13, R5, p03 * R4)));
      group_sum3<G>(t1, t2, t3, (lane & 1) != 0, (lane & 2) != 0);
      if (kT) {
        n1 = t1; n2 = t2; n3 = t3;
      } else {
        n1 = t1 * inv_x; n2 = t2 * inv_x; n3 = t3 * inv_x;
      }
      if (kDrop) {
        n0 = 1.0 - ((n1 + n2) + n3);
      } else {
        const double t0 = group_sum<G>(fma(p03, R4, fma(p02, R3, fma(p01, R1, p00 * R0))));
        n0 = kT ? t0 : t0 * inv_x;
      }
    };
    // enough idle groups with a pair waiting, or nobody left iterating: time to refill
    auto refill_now = [&]() -> bool {
      if (__all(done)) return true;
      return __popcll(__ballot(done && pend_j < n_kept)) >= THRESH * G;
    };
    bool full = __any(!done && f0 < kFullBelow);
    for (;;) {
      if (kTree && !full) {
        bool leave = false;
        for (;;) {
          double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
          if (!kMaskDone || !done) em_step(PairedTag(), n0, n1, n2, n3);
          if (__any(!done && !(n1 < 2.0))) break;  // an odd step: second opinion below, same iteration
          if (__any(!done && fabs(n1 - f1) < kEpsilonTie)) {
            const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
            if (!done && fabs(eps - kEpsilon) < kTieMargin) tie = true;
            if (!done && eps < kEpsilon) {
              done = true;
              n_iter = it_g;
              f0 = n0; f1 = n1; f2 = n2; f3 = n3;
            }
          }
          if (!done) {
            f0 = n0; f1 = n1; f2 = n2; f3 = n3;
            if (++it_g >= (uint32_t)kIterMax) done = true;  // (n_iter stays ITER_MAX: gen_func.cpp:1041)
          }
          if (refill_now()) {
            leave = true;
            break;
          }
          if (__any(!done && f0 < kFullBelow)) {
            full = true;
            break;
          }
        }
        if (leave) break;
        if (full) continue;
      }
      double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
      if (!kMaskDone || !done) em_step(SingleTag(), n0, n1, n2, n3);
      const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
      if (!done) {
        if (!(n1 < 2.0)) {
          f0 = f1 = f2 = f3 = __builtin_nan("");
          done = true;
          n_iter = it_g;
        } else {
          f0 = n0; f1 = n1; f2 = n2; f3 = n3;
          if (fabs(eps - kEpsilon) < kTieMargin) tie = true;
          if (eps < kEpsilon) {
            done = true;
            n_iter = it_g;
          } else if (++it_g >= (uint32_t)kIterMax) {
            done = true;
          }
        }
      }
      if (refill_now()) break;
      if (!full) full = __any(!done && f0 < kFullBelow);
      // (once in the full form the wavefront stays in it until the refill, as the shipping kernel does within a generation)
    }

    // ---- the groups that have finished their pair hand the result over ----
    const bool fin = live && done;
    if (fin) {
      double g0 = f0, g1 = f1, g2 = f2, g3 = f3;
      unrelabel(flip1, flip2, g0, g1, g2, g3);
      if (gl == 0) {
        RunResult &r = ring[(uint32_t)grp * kPer + held];
        r.f[0] = g0; r.f[1] = g1; r.f[2] = g2; r.f[3] = g3;
        r.n_iter = n_iter | (tie ? kTieBit : 0u);
      }
      ++held;
      live = false;
    }
    // a group's entries fill in order; when any group has used all of its own, every FINISHED entry is written out and the
    // groups start over -- an entry of a pair still iterating moves to the front of its group's part
    if (__any(held >= kPer)) {
      flush();
      if (live && gl == 0 && held > 0) {
        ring[(uint32_t)grp * kPer] = ring[(uint32_t)grp * kPer + held];
        ring[(uint32_t)grp * kPer + held].rec = ~0ull;
      }
      held = 0;
    }
  }
  flush();
}

}  // namespace ngsld
#endif
