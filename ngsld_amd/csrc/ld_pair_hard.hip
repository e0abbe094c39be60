// ld_pair_hard.hip -- the pair kernel of HARD-CALLED matrices (gfx950).
//
// Called genotypes -- a text genotype file of {-1,0,1,2} (shared/read_data.cpp:83-99), or likelihoods put through
// --call_geno (ngsLD.cpp:92-98, gen_func.cpp:886-914) -- reach haplo_freq as likelihood triples that are exactly
// (1,0,0), (0,1,0), (0,0,1) or three equal values (no data).  Every individual of a pair then belongs to one of
// 4 x 4 genotype combinations, and all individuals of a combination contribute the same term to the reference's sums
// (gen_func.cpp:1086-1105): the EM over n_ind individuals is the EM over 16 weighted combinations,
//     ff_k = sum_G count[G] * tmp_k(G) / sum(G),
// with count[G] = popcount(set1[g1] & set2[g2]) from per-site bit sets.  The work per pair no longer grows with the
// cohort beyond n_ind / 64 popcounts per combination.  Same results as the per-individual kernels up to summation order
// (count * term instead of count additions of it): nIter and sample_size exact, everything else inside 1e-9
// (tests/test_gpu_hardcalls.py checks both paths pair by pair).
//
// Mapping: a group of 16 lanes owns a pair (lane = 4 * g1 + g2), four pairs per wavefront in lockstep, run form
// (one workgroup per run of <= 16 items of one row, the row's bit sets in LDS, claims from the run's list, result rings)
// like pair_ld_group_kernel.  The step is the full four-value form with one reciprocal per combination; no allele
// relabelling is needed for it.
#include "ld_group_reduce.h"
#include "ld_run_pipeline.h"
#include "ld_dispatch.h"

namespace ngsld {

__global__ __launch_bounds__(256) void classify_hard_kernel(const double *planes, uint64_t site_stride, uint32_t np,
                                                             uint32_t n_ind, uint64_t n_sites, uint64_t *masks, double *u,
                                                             int *all_hard) {
  // one wavefront per site: lane l looks at individuals 64 w + l, the ballots are the bit sets' words
  const int lane = threadIdx.x & 63;
  const uint64_t site = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (site >= n_sites) return;
  // a matrix of likelihoods is found out by the first wavefronts: the rest of the grid leaves at once (the bit sets of a
  // matrix that is not all called genotypes are never used) -- the kernel cost 1.2 ms per 1.2 GB, more than the prep kernel
  if (__builtin_amdgcn_readfirstlane(*(volatile int *)all_hard) == 0) return;
  const double *pl = planes + site * site_stride;
  const uint32_t words = (n_ind + 63) / 64;
  uint64_t *m = masks + site * 4ull * words;
  double uu = 0.0;
  bool hard = true, have_u = false;
  for (uint32_t w = 0; w < words; ++w) {
    const uint32_t i = w * 64 + (uint32_t)lane;
    const bool in = i < n_ind;
    const double a0 = in ? pl[i] : 0.0, a1 = in ? pl[np + i] : 0.0, a2 = in ? pl[2 * np + i] : 0.0;
    const bool c0 = in && a0 == 1.0 && a1 == 0.0 && a2 == 0.0;
    const bool c1 = in && a0 == 0.0 && a1 == 1.0 && a2 == 0.0;
    const bool c2 = in && a0 == 0.0 && a1 == 0.0 && a2 == 1.0;
    const bool c3 = in && a0 == a1 && a1 == a2;
    if (in && !(c0 || c1 || c2 || c3)) hard = false;
    if (__any(!hard)) break;
    if (c3) {  // every individual without data must carry the same value (it does: one arithmetic, read_data.cpp:94-99)
      if (have_u && uu != a0) hard = false;
      uu = a0;
      have_u = true;
    }
    const uint64_t b0 = __ballot(c0), b1 = __ballot(c1), b2 = __ballot(c2), b3 = __ballot(c3);
    if (lane == 0) {
      m[w] = b0;
      m[words + w] = b1;
      m[2 * words + w] = b2;
      m[3 * words + w] = b3;
    }
  }
  // one u per site: lanes may each have seen one
  const uint64_t with_u = __ballot(have_u);
  double site_u = 1.0 / 3.0;
  if (with_u) {
    const int src = __ffsll((unsigned long long)with_u) - 1;
    site_u = read_lane(uu, src);
    if (have_u && uu != site_u) hard = false;
  }
  if (lane == 0) u[site] = site_u;
  if (__any(!hard) && lane == 0) atomicExch(all_hard, 0);
}

hipError_t launch_classify_hard(const double *planes, uint64_t site_stride, uint32_t np, uint32_t n_ind, uint64_t n_sites,
                                uint64_t *masks, double *u, int *all_hard, hipStream_t stream) {
  if (n_sites == 0) return hipSuccess;
  hipLaunchKernelGGL(classify_hard_kernel, dim3((unsigned)((n_sites + 3) / 4)), dim3(256), 0, stream, planes, site_stride, np,
                     n_ind, n_sites, masks, u, all_hard);
  return hipGetLastError();
}

template <bool MASKED>
__global__ __launch_bounds__(256, 4) void pair_ld_hard_kernel(PairArgs A) {
  constexpr int G = 16;
  constexpr uint32_t kRing = 32;
  __shared__ uint64_t row_sets[4 * kHardMaxWords];
  __shared__ RunResult rings[4][kRing];
  __shared__ RunList list;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = lane >> 4, gl = lane & 15;
  const int g1 = gl >> 2, g2 = gl & 3;  // this lane's genotype combination: 0, 1, 2 = called genotype, 3 = no data
  const Run run = A.runs[blockIdx.x];
  const Item *g_items = A.items_all + run.first_item;
  const uint32_t s1 = g_items[0].s1;
  const uint32_t W = A.mask_words;
  RunList *L = &list;
  RunResult *ring = rings[wave];

  for (uint32_t i = threadIdx.x; i < 4 * W; i += 256) row_sets[i] = A.hard_masks[(uint64_t)s1 * 4 * W + i];
  build_run_list(L, g_items, run.n_items);  // ends with a barrier: bit sets and list in place
  const uint32_t n_kept = L->base[run.n_items];
  const uint32_t s2_base = L->items[0].s2_begin;
  const uint64_t rec_base = g_items[0].first_record - A.out_base;
  const double m1 = A.sc4[4 * (uint64_t)s1], mean1 = A.sc4[4 * (uint64_t)s1 + 1], rsx1 = A.sc4[4 * (uint64_t)s1 + 2];
  const double u1 = A.hard_u[s1];
  const double a0 = g1 == 0 ? 1.0 : (g1 == 3 ? u1 : 0.0), a1 = g1 == 1 ? 1.0 : (g1 == 3 ? u1 : 0.0),
               a2 = g1 == 2 ? 1.0 : (g1 == 3 ? u1 : 0.0);
  const double c1 = fma(2.0, a2, a1) - mean1;  // expected genotype p1 + 2 p2, centred (ngsLD.cpp:113, :290)

  struct Cand {
    uint32_t s2;
    uint64_t rec;
    bool ok;
  };
  auto claim_group = [&]() -> Cand {
    uint32_t j = 0;
    if (gl == 0) j = atomicAdd(&L->claim, 1u);
    j = (uint32_t)__shfl((int)j, lane & ~(G - 1));
    if (j >= n_kept) return Cand{0u, 0ull, false};
    return Cand{s2_base + (uint32_t)L->cand[j], rec_base + j, true};
  };
  uint32_t held = 0;
  auto flush = [&](uint32_t n) {
    if ((uint32_t)lane < n) {
      const RunResult r = ring[lane];
      if (r.rec != ~0ull) write_pair(A, r.rec, r.f[0], r.f[1], r.f[2], r.f[3], r.sxy, rsx1, r.rsx2, r.x, r.n_iter);
    }
  };

  Cand cur = claim_group();
  while (__any(cur.ok)) {
    const bool active = cur.ok;
    const uint32_t s2 = active ? cur.s2 : s1;
    // ---- the pair's 16 counts: popcount(set1[g1] & set2[g2]) ----
    const uint64_t *set2 = A.hard_masks + ((uint64_t)s2 * 4 + (uint32_t)g2) * W;
    const uint64_t *set1 = row_sets + (uint32_t)g1 * W;
    uint32_t cnt = 0;
    for (uint32_t w = 0; w < W; ++w) cnt += (uint32_t)__popcll(set1[w] & set2[w]);
    if (!active) cnt = 0;
    const double m2 = A.sc4[4 * (uint64_t)s2], mean2 = A.sc4[4 * (uint64_t)s2 + 1], rsx2 = A.sc4[4 * (uint64_t)s2 + 2];
    const double u2 = A.hard_u[s2];
    const double b0 = g2 == 0 ? 1.0 : (g2 == 3 ? u2 : 0.0), b1 = g2 == 1 ? 1.0 : (g2 == 3 ? u2 : 0.0),
                 b2 = g2 == 2 ? 1.0 : (g2 == 3 ? u2 : 0.0);
    const double wgt = (double)cnt;
    // individuals of this combination that take part in the EM (gen_func.cpp:1089: --ignore_miss_data leaves out whoever
    // lacks data at either site)
    const bool valid = cnt != 0 && (!MASKED || (g1 != 3 && g2 != 3));
    const double P0 = a0 * b0, P1 = a0 * b1, P2 = a0 * b2, P3 = a1 * b0, P4 = a1 * b1, P5 = a1 * b2, P6 = a2 * b0,
                 P7 = a2 * b1, P8 = a2 * b2;
    const uint32_t x = (uint32_t)group_sum<G>(valid ? wgt : 0.0);                 // exact: integers far below 2^53
    const double sxy = group_sum<G>(wgt * c1 * (fma(2.0, b2, b1) - mean2));       // over ALL individuals (ngsLD.cpp:290)
    const double inv_x = 1.0 / (double)x;  // x == 0: 0 * inf = NaN like the reference's 0 / 0

    // ---- haplo_freq (gen_func.cpp:1027-1059), four pairs in lockstep ----
    double f0 = (1 - m1) * (1 - m2), f1 = (1 - m1) * m2, f2 = m1 * (1 - m2), f3 = m1 * m2;
    if (active && (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1)) {  // error() in the reference (:1030); reported through status
      if (gl == 0) atomicExch(A.status, (int)NGSLD_ERR_MAF_RANGE);
      f0 = f1 = f2 = f3 = __builtin_nan("");
    }
    bool done = !active, tie = false;
    uint32_t n_iter = (uint32_t)kIterMax;
    for (uint32_t itn = 0; itn < (uint32_t)kIterMax; ++itn) {
      // (kHardMask: groups that have converged would sit out with their lanes switched off as in pair_ld_group_kernel --
      // measured 2.5 % SLOWER here, the iteration is too short for the mask's own bookkeeping: off)
      constexpr bool kHardMask = false;
      if (!kHardMask || !done) {
      const double p00 = f0 * f0, p01 = f0 * f1, p02 = f0 * f2, p03 = f0 * f3, p11 = f1 * f1;
      const double p12 = f1 * f2, p13 = f1 * f3, p22 = f2 * f2, p23 = f2 * f3, p33 = f3 * f3;
      const double w1 = p01 + p01, w3 = p02 + p02, w4 = 2.0 * (p03 + p12), w5 = p13 + p13, w7 = p23 + p23;
      double s = p00 * P0;  // the reference's 16-term `sum` (gen_func.cpp:1093-1096) as a bilinear form, see ld_device.h
      s = fma(w1, P1, s); s = fma(p11, P2, s); s = fma(w3, P3, s); s = fma(w4, P4, s); s = fma(w5, P5, s);
      s = fma(p22, P6, s); s = fma(w7, P7, s); s = fma(p33, P8, s);
      // an empty combination adds nothing; a populated one whose s is 0 poisons every R with NaN, which is the
      // reference's 0/0 for each of those individuals (gen_func.cpp:1103)
      const double r = valid ? rcp_refined(s) * wgt : 0.0;
      const double R0 = P0 * r, R1 = P1 * r, R2 = P2 * r, R3 = P3 * r, R4 = P4 * r, R5 = P5 * r, R6 = P6 * r, R7 = P7 * r,
                   R8 = P8 * r;
      double t1 = fma(p13, R5, fma(p12, R4, fma(p11, R2, p01 * R1)));
      double t2 = fma(p23, R7, fma(p22, R6, fma(p12, R4, p02 * R3)));
      double t3 = fma(p33, R8, fma(p23, R7, fma(p13, R5, p03 * R4)));
      const double t0 = group_sum<G>(fma(p03, R4, fma(p02, R3, fma(p01, R1, p00 * R0))));
      group_sum3<G>(t1, t2, t3, (lane & 1) != 0, (lane & 2) != 0);
      const double n0 = t0 * inv_x, n1 = t1 * inv_x, n2 = t2 * inv_x, n3 = t3 * inv_x;  // f = ff / (2x), gen_func.cpp:1108-1109
      if (!done) {
        if (!(n1 < 2.0)) {  // the reference's all-NaN step: "converges" at this iteration (gen_func.cpp:1049-1055)
          f0 = f1 = f2 = f3 = __builtin_nan("");
          done = true;
          n_iter = itn;
        } else {
          const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
          f0 = n0; f1 = n1; f2 = n2; f3 = n3;
          if (fabs(eps - kEpsilon) < kTieMargin) tie = true;  // too close to call: replayed in the reference's order
          if (eps < kEpsilon) {  // gen_func.cpp:1054-1055
            done = true;
            n_iter = itn;
          }
        }
      }
      }
      if (__all(done)) break;
    }
    if (gl == 0) {  // one ring entry per group and generation; a group without a pair leaves a hole
      RunResult &r = ring[held + (uint32_t)grp];
      r.f[0] = f0; r.f[1] = f1; r.f[2] = f2; r.f[3] = f3;
      r.sxy = sxy;
      r.rsx2 = rsx2;
      r.x = x;
      r.n_iter = n_iter | (tie ? kTieBit : 0u);
      r.rec = active ? cur.rec : ~0ull;
    }
    held += 4;
    if (held + 4 > kRing) {
      flush(held);
      held = 0;
    }
    cur = claim_group();
  }
  flush(held);
}

hipError_t launch_pair_hard(bool masked, const PairArgs &a, hipStream_t stream) {
  if (a.n_runs == 0) return hipSuccess;
  if (a.n_runs > 0x7fffffffull || a.mask_words == 0 || a.mask_words > kHardMaxWords) return hipErrorInvalidValue;
  const dim3 grid((unsigned)a.n_runs), block(256);
  if (masked)
    hipLaunchKernelGGL((pair_ld_hard_kernel<true>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((pair_ld_hard_kernel<false>), grid, block, 0, stream, a);
  return hipGetLastError();
}

}  // namespace ngsld
