// ld_kernel_run.h -- pair_ld_run_kernel: one wavefront per pair, the row vector shared in LDS, runs of items (the headline's
// kernel at eight slots; the pipeline is described in ld_run_pipeline.h); instantiated in ld_pair_w1.hip.
#pragma once

#include "ld_run_pipeline.h"

namespace ngsld {

// SKIP (a launch over a matrix with degenerate sites, PairArgs::skip_degenerate): a pair with such a site -- marked in
// sc4[.][3] by site_skip_kernel, ld_prep.hip -- is staged for its Pearson moment only; its frequencies are left NaN, which
// write_pair flags, and the exact-order replay is the pair's one evaluation (it was going to start the pair over anyway).  A
// variant of its own so that launches over SNP-called matrices keep the instruction stream they had.
template <int SLOTS, bool MASKED, bool SKIP = false>
__global__ __launch_bounds__(256, 2) void pair_ld_run_kernel(PairArgs A) {
  constexpr int kSiteBytes = SLOTS * 64 * 3 * 8;
  constexpr int kBuf = kSiteBytes + 32;
  constexpr uint32_t kNp = SLOTS * 64;
  // (ten slots: five site buffers of 15 KB leave 5 KB for rings and list under the 80 KB that let two workgroups share a CU)
  constexpr uint32_t kRing = SLOTS <= 9 ? 32 : 8;
  constexpr int kRingOff = kSiteBytes + 4 * kBuf;
  constexpr int kListOff = kRingOff + 4 * (int)(kRing * sizeof(RunResult));
  __shared__ __attribute__((aligned(16))) char smem[kListOff + sizeof(RunList)];
  static_assert(sizeof(smem) <= 81920, "run kernel: two workgroups per CU need <= 80 KB of LDS each");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // (An XCD-aware order -- each XCD taking 64 consecutive runs of every 512 -- was measured and dropped: neighbouring
  // rows drift apart by more pairs than the 4 MB L2 bridges, L2-miss traffic rose 35 % and the kernel lost 0.7 %.)
  const Run run = A.runs[blockIdx.x];
  const Item *g_items = A.items_all + run.first_item;
  const uint32_t s1 = g_items[0].s1;
  const double m1 = A.sc4[4 * (uint64_t)s1], mean1 = A.sc4[4 * (uint64_t)s1 + 1], rsx1 = A.sc4[4 * (uint64_t)s1 + 2];
  const bool skip1 = SKIP && A.sc4[4 * (uint64_t)s1 + 3] != 0.0;
  char *lds_a = smem;
  char *lds_b = smem + kSiteBytes + wave * kBuf;
  RunResult *ring = reinterpret_cast<RunResult *>(smem + kRingOff) + wave * kRing;
  RunList *L = reinterpret_cast<RunList *>(smem + kListOff);

  dma_site_to_lds<SLOTS>(A.planes + (uint64_t)s1 * A.site_stride, lds_a, lane, wave, 4);  // a quarter per wavefront
  build_run_list(L, g_items, run.n_items);  // its last barrier: row vector and list in place
  const uint32_t n_kept = (uint32_t)__builtin_amdgcn_readfirstlane((int)L->base[run.n_items]);
  const uint32_t s2_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)L->items[0].s2_begin);
  const uint64_t rec_base = g_items[0].first_record - A.out_base;

  // one computed pair of the run: site and record index
  struct Cand {
    uint32_t s2;
    uint64_t rec;
    bool ok;
  };
  auto claim_next = [&]() -> Cand {  // (the maf[s2] / sub-sampling filters, ngsLD.cpp:270-282, already shaped the list)
    uint32_t j = 0;
    if (lane == 0) j = atomicAdd(&L->claim, 1u);
    j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
    if (j >= n_kept) return Cand{0u, 0ull, false};
    const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)L->cand[j]);
    return Cand{s2_base + off, rec_base + j, true};
  };
  auto dma_site = [&](uint32_t s2) {
    dma_site_to_lds<SLOTS>(A.planes + (uint64_t)s2 * A.site_stride, lds_b, lane, 0, 1);
    if (lane < 2)
      __builtin_amdgcn_global_load_lds((glb_void_t *)(reinterpret_cast<const char *>(A.sc4 + 4 * (uint64_t)s2) + lane * 16),
                                       (lds_void_t *)(lds_b + kSiteBytes), 16, 0, 0);
  };
  auto flush = [&](uint32_t n) {  // lane t derives and writes the record of ring entry t
    if ((uint32_t)lane < n) {
      const RunResult r = ring[lane];
      write_pair(A, r.rec, r.f[0], r.f[1], r.f[2], r.f[3], r.sxy, rsx1, r.rsx2, r.x, r.n_iter);
    }
  };

  Cand cur = claim_next();
  if (cur.ok) dma_site(cur.s2);
  uint32_t held = 0;
  while (cur.ok) {
    const Cand nxt = claim_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's site copy (issued a pair ago) has landed
    const double *sc = reinterpret_cast<const double *>(lds_b + kSiteBytes);
    const double m2 = uniform(sc[0]), mean2 = uniform(sc[1]), rsx2 = uniform(sc[2]);
    const bool skip = SKIP && (skip1 || uniform(sc[3]) != 0.0);  // (read here: the next site's copy is about to land on sc)
    double P[SLOTS][9];
    uint32_t vbits;
    double sxy;
    const Relabel rl = relabel(m1, m2, mean1, mean2);
    stage_pair<SLOTS, MASKED, !MASKED, true>(reinterpret_cast<const double *>(lds_a), kNp, (uint32_t)lane,
                                             reinterpret_cast<const double *>(lds_b), kNp, (uint32_t)lane, (uint32_t)lane,
                                             A.n_ind, rl.mean1, rl.mean2, P, vbits, sxy, rl.flip1, rl.flip2);
    // all ds_reads of the buffer are consumed (P is computed): start the copy of the next site over it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (nxt.ok) dma_site(nxt.s2);
    const uint32_t x = MASKED ? count_valid<SLOTS>(vbits) : A.n_ind;
    const double inv_x = MASKED ? 1.0 / (double)x : A.inv_n;
    sxy = fma(-(double)A.n_ind * rl.mean1, rl.mean2, wave_sum1_bcast(sxy));  // centred: sum e1 e2 - n mean1 mean2
    double f0, f1, f2, f3;
    uint32_t n_iter = 0;
    if (skip) {  // (wavefront-uniform)
      f0 = f1 = f2 = f3 = __builtin_nan("");
    } else {
      n_iter = em_pair<SLOTS, 1>(P, vbits, inv_x, rl.m1, rl.m2, f0, f1, f2, f3, (double (*)[1][4]) nullptr, 0, lane, A.status);
      unrelabel(rl.flip1, rl.flip2, f0, f1, f2, f3);
    }
    if (lane == 0) {
      RunResult &r = ring[held];
      r.f[0] = f0; r.f[1] = f1; r.f[2] = f2; r.f[3] = f3;
      r.sxy = sxy;
      r.rsx2 = rsx2;
      r.x = x;
      r.n_iter = n_iter;
      r.rec = cur.rec;
    }
    if (++held == kRing) {
      flush(held);
      held = 0;
    }
    cur = nxt;
  }
  flush(held);
}

}  // namespace ngsld
