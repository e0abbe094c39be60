// ld_pair_ab.hip -- one wavefront per pair beyond what the P form holds in registers: the EM step in its a/b form.
// The default for 641..960 individuals (pair_config, by measurement: profiles/r03/sweep_513_1024.txt, sweep_areg.txt);
// NGSLD_PAIR_KERNEL=ab selects it for 513..1024.
//
// The idea.  The other kernels keep P = a (x) b (9 products per individual, 18 VGPRs) for the whole pair, which caps a
// lane at 8 individuals; above 512 individuals a pair is spread over 2..8 wavefronts that meet behind a barrier in EVERY
// EM iteration (pair_ld_kernel<SLOTS, WAVES>) -- at n_ind 1000 the SIMDs issue VALU 85 % of the time against 97 % in the
// one-wavefront kernel, and each of the two wavefronts pays the per-iteration bookkeeping (f products, contraction,
// reduction, convergence test) for 8 slots only: 2 x 280 = 560 VALU instructions per pair and iteration.
// Here a lane holds up to 16 individuals of ONE factor: b (site 2) in 6 VGPRs per slot.  The row's vector a sits in LDS once
// per workgroup (shared by the workgroup's eight wavefronts) and -- round 3 -- a lane copies its own slots of it into
// REGISTERS once per run wherever they fit beside b (AbRegs: all of them up to 13 slots); what does not fit is re-read from
// LDS in every iteration (24 B per individual-iteration).  Per individual and iteration
//     v = W(f) b              9 mul / FMA        (W: the 3x3 two-locus genotype weights)
//     s = a . v               3 FMA              (the reference's 16-term `sum`, gen_func.cpp:1093-1096)
//     r = 1 / s               shared: one v_rcp_f64 per TREE individuals of a lane (RcpTree)
//     R[g1][g2] += (r a[g1]) b[g2]     3 mul + 8 FMA (R[0][0] is not needed: hap 0 is recovered from the sum)
// = 23 f64 VALU + the tree's share against 17 + share in the P form, no barrier, no LDS exchange, the bookkeeping once per
// 16 slots: 500 VALU instructions per pair and iteration (-11 %), 237-254 VGPRs, no scratch.
//
// The measurement at n_ind 1,000 (same box, 12,000 x 1,000 all pairs, profiles/r02_multi/pmc_multi_vs_ab_n1000.txt), where
// it does NOT pay: 8.83e7 pairs/s against 8.98e7 for the two-wavefront kernel.  The counters say why: this kernel needs 8 % FEWER SIMD cycles
// (1.65e12 against 1.79e12, VALU busy 87.6 % against 85.1 %) -- and runs at a 9 % LOWER shader clock (1.99 GHz against
// 2.20 GHz, GRBM_GUI_ACTIVE / duration): the chip is power-limited under these kernels, 5x the LDS traffic
// (SQ_LDS_IDX_ACTIVE 1.5e11 against 5.8e10) and 35 % more f64 FMAs per individual cost more energy than the barrier
// stalls they remove -- stalled cycles are cheap, busy ones are not.  What decides pairs/s here is energy per pair, i.e.
// instructions and bytes moved per pair, not how full the issue slots are.
//
// Everything else is the run kernel's pipeline (pair_ld_run_kernel): a workgroup works through a run of <= 16 items of
// one row, wavefronts claim pairs from an LDS list, a site's scalars ride behind its copy, results go through a ring and
// become records one LANE per pair.  LDS holds the row vector (24 KB), and per wavefront a landing buffer for the FIRST
// HALF (8 slots) of the next site, copied global -> LDS while the current EM loop runs; the second half is loaded straight
// into registers when the pair starts (a full-site buffer per wavefront would leave room for 4 wavefronts per CU).
#include <type_traits>

#include "ld_run_pipeline.h"
#include "ld_dispatch.h"

namespace ngsld {

constexpr int kAbLand = 8;  // slots of the next site that are prefetched into LDS

// Compile-time loop: f(integral_constant<int, B>), f(integral_constant<int, B + S>), ... below E.  (A `#pragma unroll`
// over the trees was left rolled by the compiler, which put the register array b into scratch memory.)
template <int B, int E, int S, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>());
    static_for<B + S, E, S>(f);
  }
}

// haplo_freq (gen_func.cpp:1027-1059) for one pair, a/b form.  la0 / la1 / la2: this lane's view of the row vector's
// genotype planes in LDS (already relabelled), element j at [64 j].  Same control flow as em_pair: the hot loop holds the
// shared-reciprocal step in its three-value form; a step that does not look sane is redone with one reciprocal per
// individual; a pair whose hap 0 falls below kFullBelow finishes in the full four-value form.
//   AREG: the row vector's first AREG slots of this lane sit in REGISTERS (Ar, loaded once per run: the row is the same for
//   every pair of the run), only the slots beyond are re-read from LDS in every iteration
//   WAVES > 1 (pair_ld_abm_kernel): several wavefronts share the pair, the whole slice of a sits in registers, and the partial
//   sums of an iteration meet in LDS exactly as in em_pair.  Every individual counts: empty slots are GHOSTS (a = b =
//   (1, 0, 0): s = f0^2, nothing added to R[1..8]; the steps that accumulate R[0] skip them by their validity bit).
//   --ignore_miss_data: b = 0 and a pad of 1 read off the validity bit, so that s is exactly 1 and nothing is added to R
template <int SLOTS, bool MASKED, int TREE, int AREG, int WAVES = 1>
__device__ __forceinline__ uint32_t em_pair_ab(const double (&B)[SLOTS][3], const double (&Ar)[AREG > 0 ? AREG : 1][3],
                                               const double *la0, const double *la1, const double *la2, uint32_t vbits,
                                               const double *pads, double inv_x, double m1, double m2, double &f0, double &f1,
                                               double &f2, double &f3, int lane, int *status,
                                               double (*xch)[WAVES][4] = nullptr, int sub = 0, uint32_t *xpar = nullptr) {
  static_assert(WAVES == 1 || AREG == SLOTS, "several wavefronts per pair: the whole slice of a in registers");
  auto a_of = [&](int j, int g) -> double {  // (j, g compile-time after unrolling)
    if (j < AREG) return Ar[j < AREG ? j : 0][g];
    return g == 0 ? la0[64 * j] : (g == 1 ? la1[64 * j] : la2[64 * j]);
  };
  f0 = (1 - m1) * (1 - m2); f1 = (1 - m1) * m2; f2 = m1 * (1 - m2); f3 = m1 * m2;  // gen_func.cpp:1034-1037
  if (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1) {  // error() in the reference (:1030); reported through status
    if (lane == 0 && sub == 0) atomicExch(status, (int)NGSLD_ERR_MAF_RANGE);
    f0 = f1 = f2 = f3 = __builtin_nan("");
  }
  asm("" : "+v"(inv_x));
  const double pad_last = ((vbits >> (SLOTS - 1)) & 1u) ? 0.0 : 1.0;  // padding lanes of the last slot hold a == b == 0
  bool bad = false, tie = false;
  uint32_t n_iter = 0;

  auto em_step = [&](auto tree_tag, double &n0, double &n1, double &n2, double &n3) {
    constexpr bool kTree = decltype(tree_tag)::value;  // shared reciprocals, three-value form; otherwise one per individual, full form
    // (several wavefronts, --ignore_miss_data: the pads are read off the validity bits in every iteration -- were the bits not
    // hidden here, loop-invariant code motion would form all SLOTS pads in front of the EM loop and keep them in registers the
    // kernel does not have)
    if (WAVES > 1 && MASKED) asm volatile("" : "+v"(vbits));
    const double p00 = f0 * f0, p01 = f0 * f1, p02 = f0 * f2, p03 = f0 * f3, p11 = f1 * f1;
    const double p12 = f1 * f2, p13 = f1 * f3, p22 = f2 * f2, p23 = f2 * f3, p33 = f3 * f3;
    const double w1 = p01 + p01, w3 = p02 + p02, w4 = 2.0 * (p03 + p12), w5 = p13 + p13, w7 = p23 + p23;
    double R0 = 0, R1 = 0, R2 = 0, R3 = 0, R4 = 0, R5 = 0, R6 = 0, R7 = 0, R8 = 0;
    auto slot_s = [&](int j, double a0, double a1, double a2, bool padded) -> double {
      const double v0 = fma(p11, B[j][2], fma(w1, B[j][1], p00 * B[j][0]));
      const double v1 = fma(w5, B[j][2], fma(w4, B[j][1], w3 * B[j][0]));
      const double v2 = fma(p33, B[j][2], fma(w7, B[j][1], p22 * B[j][0]));
      const double pad = WAVES > 1 ? (((vbits >> j) & 1u) ? 0.0 : 1.0) : (MASKED ? pads[j] : pad_last);
      double s = padded ? fma(a0, v0, pad) : a0 * v0;
      s = fma(a1, v1, s);
      return fma(a2, v2, s);
    };
    auto slot_acc = [&](int j, double a0, double a1, double a2, double r) {
      const double r0 = r * a0, r1 = r * a1, r2 = r * a2;
      if (!kTree) R0 = fma(r0, B[j][0], R0);
      R1 = fma(r0, B[j][1], R1); R2 = fma(r0, B[j][2], R2);
      R3 = fma(r1, B[j][0], R3); R4 = fma(r1, B[j][1], R4); R5 = fma(r1, B[j][2], R5);
      R6 = fma(r2, B[j][0], R6); R7 = fma(r2, B[j][1], R7); R8 = fma(r2, B[j][2], R8);
    };
    if constexpr (kTree) {
      static_for<0, SLOTS, TREE>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        constexpr int kT = TREE;
        double av[kT][3], sv[kT], rv[kT];
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(NGSLD_PRIO_S);
#pragma unroll
        for (int t = 0; t < kT; ++t) {
          const int j = h + t;
          if (j < SLOTS) {
            av[t][0] = a_of(j, 0); av[t][1] = a_of(j, 1); av[t][2] = a_of(j, 2);
            sv[t] = slot_s(j, av[t][0], av[t][1], av[t][2], WAVES == 1 ? (MASKED || j == SLOTS - 1) : MASKED);
          } else {  // the last tree of a slot count that is no multiple of TREE: a neutral factor
            av[t][0] = av[t][1] = av[t][2] = 0.0;
            sv[t] = 1.0;
          }
        }
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(NGSLD_PRIO_TREE);
        // 1/x rides on the root inverse: every R -- and with them the three t_k -- come out divided by x
        RcpTree<kT>::down(sv, rcp_refined(RcpTree<kT>::prod(sv)) * inv_x, rv);
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(NGSLD_PRIO_R);
#pragma unroll
        for (int t = 0; t < kT; ++t)
          if (h + t < SLOTS) slot_acc(h + t, av[t][0], av[t][1], av[t][2], rv[t]);
        // one tree at a time: left alone, the scheduler hoists the next trees' LDS reads and s sums above this tree's
        // R sums for latency, and the live a / s values of several trees no longer fit the register file beside b
        __builtin_amdgcn_sched_barrier(0);
      });
      if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(NGSLD_PRIO_SERIAL);
    } else {
#pragma unroll
      for (int j = 0; j < SLOTS; ++j) {
        if ((vbits >> j) & 1u) {
          const double a0 = a_of(j, 0), a1 = a_of(j, 1), a2 = a_of(j, 2);
          slot_acc(j, a0, a1, a2, rcp_refined(slot_s(j, a0, a1, a2, false)));
        }
      }
    }
    // t_k = sum_h f_k f_h R[G(k,h)]  (= this lane's share of ff_k / 2, gen_func.cpp:1098-1104)
    double t0 = kTree ? 0.0 : fma(p03, R4, fma(p02, R3, fma(p01, R1, p00 * R0)));
    double t1 = fma(p13, R5, fma(p12, R4, fma(p11, R2, p01 * R1)));
    double t2 = fma(p23, R7, fma(p22, R6, fma(p12, R4, p02 * R3)));
    double t3 = fma(p33, R8, fma(p23, R7, fma(p13, R5, p03 * R4)));
    if (kTree) {
      if constexpr (WAVES > 1) {
        // as em_pair: the row totals go to the exchange buffer from the lanes that hold them, every wavefront adds the
        // partials up in the same order (the new frequencies must be the same bits in all of them)
        const double w = wave_sum3_rows(t1, t2, t3);
        const int par = (int)((*xpar)++ & 1u);
        const int row = lane >> 4;
        const uint32_t base = lds_addr(&xch[par][0][0]);  // [value k = 0..2][wavefront]
        if ((lane & 15) == 0 && row != 1) lds_post(base + (uint32_t)((row == 0 ? 0 : row - 1) * WAVES + sub) * 8u, w);
        lds_barrier();
        if constexpr (WAVES == 8) {
          double v;
          asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(base + (uint32_t)(lane & 31) * 8u) : "memory");
          v += dpp_mov<0xB1>(v);
          v += dpp_mov<0x4E>(v);
          v += dpp_mov<0x141>(v);
          t1 = read_lane(v, 0); t2 = read_lane(v, 8); t3 = read_lane(v, 16);
        } else {
          constexpr int kHalf = WAVES / 2;
          dbl2 q[3 * kHalf];
          lds_gather<3 * kHalf>(base, q);
          t1 = q[0][0] + q[0][1]; t2 = q[kHalf][0] + q[kHalf][1]; t3 = q[2 * kHalf][0] + q[2 * kHalf][1];
#pragma unroll
          for (int v = 1; v < kHalf; ++v) {
            t1 += q[v][0]; t2 += q[kHalf + v][0]; t3 += q[2 * kHalf + v][0];
            t1 += q[v][1]; t2 += q[kHalf + v][1]; t3 += q[2 * kHalf + v][1];
          }
        }
      } else {
        wave_sum3(t1, t2, t3);
      }
      n1 = t1; n2 = t2; n3 = t3;  // already divided by x (every wavefront scales its own partial sums)
      n0 = 1.0 - ((n1 + n2) + n3);
    } else {
      wave_sum4(t0, t1, t2, t3);
      if constexpr (WAVES > 1) {  // rare: plain LDS accesses
        const int par = (int)((*xpar)++ & 1u);
        if (lane == 0) {
          xch[par][sub][0] = t0; xch[par][sub][1] = t1; xch[par][sub][2] = t2; xch[par][sub][3] = t3;
        }
        lds_barrier();
        t0 = t1 = t2 = t3 = 0.0;
        for (int w = 0; w < WAVES; ++w) {
          t0 += xch[par][w][0]; t1 += xch[par][w][1]; t2 += xch[par][w][2]; t3 += xch[par][w][3];
        }
      }
      n0 = t0 * inv_x; n1 = t1 * inv_x; n2 = t2 * inv_x; n3 = t3 * inv_x;
    }
  };

  constexpr double kFullBelow = 0x1p-10;
  bool full = __builtin_amdgcn_ballot_w64(f0 < kFullBelow) != 0;
  bool done = false;
  while (!done && n_iter < (uint32_t)kIterMax) {
    if (!full) {
      for (; n_iter < (uint32_t)kIterMax; ++n_iter) {
        double n0, n1, n2, n3;
        em_step(PairedTag(), n0, n1, n2, n3);
        if (__builtin_amdgcn_ballot_w64(!(n1 < 2.0))) break;  // an odd step: second opinion below
        bool conv = false;
        if (__builtin_amdgcn_ballot_w64(fabs(n1 - f1) < kEpsilonTie)) {
          const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
          conv = __builtin_amdgcn_ballot_w64(eps < kEpsilon) != 0;  // gen_func.cpp:1054-1055
          tie |= __builtin_amdgcn_ballot_w64(fabs(eps - kEpsilon) < kTieMargin) != 0;
        }
        f0 = n0; f1 = n1; f2 = n2; f3 = n3;
        if (conv) {
          done = true;
          break;
        }
        if (__builtin_amdgcn_ballot_w64(n0 < kFullBelow)) {
          full = true;
          ++n_iter;  // this iteration is complete
          break;
        }
      }
      if (done || n_iter >= (uint32_t)kIterMax) break;
      if (full) continue;
      if (WAVES > 1) lds_barrier();  // an odd step: all redo it (n1 is the same everywhere), none still reads the exchange buffer
    }
    double n0, n1, n2, n3;
    em_step(SingleTag(), n0, n1, n2, n3);
    if (__builtin_amdgcn_ballot_w64(!(n1 < 2.0))) {  // the reference's all-NaN step (see em_pair)
      bad = true;
      break;
    }
    const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
    f0 = n0; f1 = n1; f2 = n2; f3 = n3;
    tie |= __builtin_amdgcn_ballot_w64(fabs(eps - kEpsilon) < kTieMargin) != 0;
    if (__builtin_amdgcn_ballot_w64(eps < kEpsilon)) break;
    ++n_iter;
  }
  if (bad) f0 = f1 = f2 = f3 = __builtin_nan("");
  return n_iter | (tie ? kTieBit : 0u);
}

template <int SLOTS, bool MASKED, int TREE, int AREG = 0>
__global__ __launch_bounds__(512, 2) void pair_ld_ab_kernel(PairArgs A) {
  constexpr int kAReg = AREG < SLOTS ? AREG : SLOTS;
  constexpr uint32_t kNp = SLOTS * 64;
  constexpr int kSiteBytes = (int)kNp * 24;
  constexpr int kLand = SLOTS < kAbLand ? SLOTS : kAbLand;
  constexpr int kLandPlane = kLand * 512;            // bytes of one genotype plane's landed slots
  constexpr int kBuf = 3 * kLandPlane + 32;          // + the site's scalars
  constexpr int kWaves = 8;
  constexpr uint32_t kRing = 16;
  constexpr int kRingOff = kSiteBytes + kWaves * kBuf;
  constexpr int kListOff = kRingOff + kWaves * (int)(kRing * sizeof(RunResult));
  __shared__ __attribute__((aligned(16))) char smem[kListOff + sizeof(RunList)];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const Run run = A.runs[blockIdx.x];
  const Item *g_items = A.items_all + run.first_item;
  const uint32_t s1 = g_items[0].s1;
  const double m1 = A.sc4[4 * (uint64_t)s1], mean1 = A.sc4[4 * (uint64_t)s1 + 1], rsx1 = A.sc4[4 * (uint64_t)s1 + 2];
  char *lds_a = smem;
  char *lds_b = smem + kSiteBytes + wave * kBuf;
  RunResult *ring = reinterpret_cast<RunResult *>(smem + kRingOff) + wave * kRing;
  RunList *L = reinterpret_cast<RunList *>(smem + kListOff);

  dma_site_to_lds<SLOTS>(A.planes + (uint64_t)s1 * A.site_stride, lds_a, lane, wave, kWaves);  // an eighth per wavefront
  // (build_run_list strides by 256 threads: the upper half of the workgroup repeats the lower half's writes)
  build_run_list(L, g_items, run.n_items);  // its last barrier: row vector and list in place
  const uint32_t n_kept = (uint32_t)__builtin_amdgcn_readfirstlane((int)L->base[run.n_items]);
  const uint32_t s2_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)L->items[0].s2_begin);
  const uint64_t rec_base = g_items[0].first_record - A.out_base;

  struct Cand {
    uint32_t s2;
    uint64_t rec;
    bool ok;
  };
  auto claim_next = [&]() -> Cand {
    uint32_t j = 0;
    if (lane == 0) j = atomicAdd(&L->claim, 1u);
    j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
    if (j >= n_kept) return Cand{0u, 0ull, false};
    const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)L->cand[j]);
    return Cand{s2_base + off, rec_base + j, true};
  };
  // the first kLand slots of every genotype plane of site s2 (3 runs of kLandPlane bytes) and its scalars -> this
  // wavefront's landing buffer, 1 KiB per wave-instruction
  auto dma_land = [&](uint32_t s2) {
    const char *g = reinterpret_cast<const char *>(A.planes + (uint64_t)s2 * A.site_stride) + lane * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      glb_void_t *gb = (glb_void_t *)(g + (size_t)pl * kNp * 8);
      lds_void_t *lb = (lds_void_t *)(lds_b + pl * kLandPlane);
#pragma unroll
      for (int k = 0; k < kLandPlane / 1024; ++k) {
        switch (k) {
          case 0: __builtin_amdgcn_global_load_lds(gb, lb, 16, 0, 0); break;
          case 1: __builtin_amdgcn_global_load_lds(gb, lb, 16, 1024, 0); break;
          case 2: __builtin_amdgcn_global_load_lds(gb, lb, 16, 2048, 0); break;
          default: __builtin_amdgcn_global_load_lds(gb, lb, 16, 3072, 0); break;
        }
      }
    }
    if (lane < 2)
      __builtin_amdgcn_global_load_lds((glb_void_t *)(reinterpret_cast<const char *>(A.sc4 + 4 * (uint64_t)s2) + lane * 16),
                                       (lds_void_t *)(lds_b + 3 * kLandPlane), 16, 0, 0);
  };
  auto flush = [&](uint32_t n) {  // lane t derives and writes the record of ring entry t
    if ((uint32_t)lane < n) {
      const RunResult r = ring[lane];
      write_pair(A, r.rec, r.f[0], r.f[1], r.f[2], r.f[3], r.sxy, rsx1, r.rsx2, r.x, r.n_iter);
    }
  };

  Cand cur = claim_next();
  if (cur.ok) dma_land(cur.s2);
  uint32_t held = 0;
  const double *la = reinterpret_cast<const double *>(lds_a) + lane;
  // the row vector's first kAReg slots into registers, once per run, already relabelled (flip1 depends on the row's maf only)
  double Ar[kAReg > 0 ? kAReg : 1][3];
  {
    const bool flip1_run = m1 > 0.5;  // (relabel())
    const double *r0 = la + (flip1_run ? 2 * kNp : 0u), *r1 = la + kNp, *r2 = la + (flip1_run ? 0u : 2 * kNp);
#pragma unroll
    for (int j = 0; j < kAReg; ++j) {
      Ar[j][0] = r0[64 * j]; Ar[j][1] = r1[64 * j]; Ar[j][2] = r2[64 * j];
    }
  }
  while (cur.ok) {
    const Cand nxt = claim_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's landing copy (issued a pair ago) is complete
    const double *sc = reinterpret_cast<const double *>(lds_b + 3 * kLandPlane);
    const double m2 = uniform(sc[0]), mean2 = uniform(sc[1]), rsx2 = uniform(sc[2]);
    const Relabel rl = relabel(m1, m2, mean1, mean2);
    const double *la0 = la + (rl.flip1 ? 2 * kNp : 0u), *la1 = la + kNp, *la2 = la + (rl.flip1 ? 0u : 2 * kNp);
    const int gb0 = rl.flip2 ? 2 : 0, gb2 = rl.flip2 ? 0 : 2;

    // ---- stage: b into registers (first kLand slots from the landing buffer, the rest straight from global), validity,
    // the Pearson cross moment ----
    double B[SLOTS][3];
    double pads[MASKED ? SLOTS : 1];
    const double *gsite = A.planes + (uint64_t)cur.s2 * A.site_stride + lane;
#pragma unroll
    for (int j = kLand; j < SLOTS; ++j) {  // issued first: they fly while the landed half is consumed
      B[j][0] = gsite[(uint32_t)gb0 * kNp + 64 * j];
      B[j][1] = gsite[kNp + 64 * j];
      B[j][2] = gsite[(uint32_t)gb2 * kNp + 64 * j];
    }
    const double *lb = reinterpret_cast<const double *>(lds_b) + lane;
#pragma unroll
    for (int j = 0; j < kLand; ++j) {
      B[j][0] = lb[gb0 * (kLandPlane / 8) + 64 * j];
      B[j][1] = lb[(kLandPlane / 8) + 64 * j];
      B[j][2] = lb[gb2 * (kLandPlane / 8) + 64 * j];
    }
    uint32_t vbits = 0;
    double sxy = 0.0;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      const double a0 = j < kAReg ? Ar[j < kAReg ? j : 0][0] : la0[64 * j], a1 = j < kAReg ? Ar[j < kAReg ? j : 0][1] : la1[64 * j],
                   a2 = j < kAReg ? Ar[j < kAReg ? j : 0][2] : la2[64 * j];
      const bool inb = (uint32_t)lane + 64u * (uint32_t)j < A.n_ind;
      bool ok = inb;
      if (MASKED) ok = inb && !miss_data(a0, a1, a2) && !miss_data(B[j][0], B[j][1], B[j][2]);  // gen_func.cpp:1089
      vbits |= (ok ? 1u : 0u) << j;
      // expected genotypes p1 + 2 p2 (ngsLD.cpp:113); pearson_r runs over ALL individuals (ngsLD.cpp:290); padding lanes
      // hold zeros
      sxy = fma(fma(2.0, a2, a1), fma(2.0, B[j][2], B[j][1]), sxy);
      if (MASKED) {  // an individual without data: b = 0 and pad 1, so its s is exactly 1 and it adds nothing to R
        const double keep = ok ? 1.0 : 0.0;
        B[j][0] *= keep; B[j][1] *= keep; B[j][2] *= keep;
        pads[j] = 1.0 - keep;
      }
    }
    // the landing buffer is consumed and every direct load has arrived: start the copy of the next site over it
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (nxt.ok) dma_land(nxt.s2);
    const uint32_t x = MASKED ? count_valid<SLOTS>(vbits) : A.n_ind;
    const double inv_x = MASKED ? 1.0 / (double)x : A.inv_n;
    sxy = fma(-(double)A.n_ind * rl.mean1, rl.mean2, wave_sum1_bcast(sxy));  // centred: sum e1 e2 - n mean1 mean2
    double f0, f1, f2, f3;
    const uint32_t n_iter = em_pair_ab<SLOTS, MASKED, TREE, kAReg>(B, Ar, la0, la1, la2, vbits, pads, inv_x, rl.m1, rl.m2, f0,
                                                                   f1, f2, f3, lane, A.status);
    unrelabel(rl.flip1, rl.flip2, f0, f1, f2, f3);
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

#ifndef NGSLD_AB_TREE
#define NGSLD_AB_TREE 8  // build-time A/B knob: individuals of a lane that share one reciprocal (4 or 8)
#endif

// ---------------------------------------------------------------------------------------------
// Several wavefronts per pair, a/b form (every individual counts; 9 .. 15 slots per lane, as pair_config picks them): where the P form would need
// twice the wavefronts -- 1,281..1,664 individuals on two, 2,561..3,328 on four, 5,121..6,656 on eight instead of four /
// eight / the streaming kernel.  Item form as pair_ld_kernel: one workgroup per item of 64 candidates; the wavefront's slice
// of the ROW vector sits in registers for the whole item (already relabelled), the first kAbLand slots of the next
// candidate's slice land in LDS while the EM loop runs, the rest is read when the pair starts; empty slots are ghosts
// (em_pair_ab); the Pearson partial sums are parked per candidate and added up when the records are written.
// ---------------------------------------------------------------------------------------------
template <int SLOTS, int WAVES, bool MASKED>
__global__ __launch_bounds__(WAVES * 64, 2) void pair_ld_abm_kernel(PairArgs A) {
  static_assert(WAVES == 2 || WAVES == 4 || WAVES == 8, "pair_ld_abm_kernel: 2, 4 or 8 wavefronts per pair");
  constexpr int kLand = SLOTS < kAbLand ? SLOTS : kAbLand;
  constexpr int kLandPlane = kLand * 512;  // bytes of one genotype plane's landed slots
  constexpr int kBuf = 3 * kLandPlane;
  constexpr int kXchBase = WAVES * kBuf;
  __shared__ __attribute__((aligned(16))) char smem[kXchBase + WAVES * 96 + 64 * sizeof(PairResult) + 64 * WAVES * sizeof(double)];
  PairResult *res = reinterpret_cast<PairResult *>(smem + kXchBase + WAVES * 96);  // one per candidate
  double (*parked)[WAVES] = reinterpret_cast<double (*)[WAVES]>(smem + kXchBase + WAVES * 96 + 64 * sizeof(PairResult));
  double (*xch)[WAVES][4] = reinterpret_cast<double (*)[WAVES][4]>(smem + kXchBase);
  double (*xch0)[2] = reinterpret_cast<double (*)[2]>(smem + kXchBase + WAVES * 64);

  const int lane = threadIdx.x & 63;
  const int sub = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const Item *item_ptr;
  if (A.tile_nk != 0) {  // tiled order: see pair_ld_kernel
    const uint32_t per = A.tile_rows * 8u;
    const uint32_t t = blockIdx.x / per, w = blockIdx.x % per;
    const uint32_t row = A.row0 + (t / A.tile_nk) * A.tile_rows + (w >> 3);
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
  const uint32_t np = A.np;
  const uint32_t i0 = (uint32_t)sub * (SLOTS * 64) + (uint32_t)lane;
  char *lds_b = smem + sub * kBuf;
  __shared__ double site_sc[3][64];
  if (threadIdx.x < it.count) {
    const uint32_t s2 = it.s2_begin + threadIdx.x;
    site_sc[0][threadIdx.x] = A.maf[s2];
    site_sc[1][threadIdx.x] = A.mean_e[s2];
    site_sc[2][threadIdx.x] = A.rsx[s2];
  }
  __syncthreads();

  // the first kLand slots of this wavefront's slice of site s2 (three runs of kLandPlane bytes) -> its landing buffer
  auto dma_land = [&](uint32_t s2) {
    const char *g = reinterpret_cast<const char *>(A.planes + (uint64_t)s2 * A.site_stride + (uint32_t)sub * (SLOTS * 64)) + lane * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int k = 0; k < kLandPlane / 1024; ++k)
        __builtin_amdgcn_global_load_lds((glb_void_t *)(g + (size_t)pl * np * 8 + k * 1024),
                                         (lds_void_t *)(lds_b + pl * kLandPlane + k * 1024), 16, 0, 0);
  };
  auto next_kept = [&](uint32_t c) -> uint32_t {  // first computed pair at or after c (ngsLD.cpp:270-282 filters)
    while (c < it.count && !((it.mask >> c) & 1ull)) ++c;
    return c;
  };

  // the cohort's members among this lane's slots and the row slice, relabelled -- with every individual counting, ghosts in
  uint32_t inbits = 0;
  double Ar[SLOTS][3];
  {
    const bool flip1 = m1 > 0.5;  // (relabel())
    const double *pa = A.planes + (uint64_t)s1 * A.site_stride;
    typedef const __attribute__((address_space(1))) double gdouble_t;
    gdouble_t *q0 = (gdouble_t *)uniform_ptr(pa + (flip1 ? 2 * np : 0u)), *q1 = (gdouble_t *)uniform_ptr(pa + np),
              *q2 = (gdouble_t *)uniform_ptr(pa + (flip1 ? 0u : 2 * np));
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      const uint32_t i = i0 + 64u * (uint32_t)j;
      const bool inb = i < A.n_ind;
      inbits |= (inb ? 1u : 0u) << j;
      const uint32_t ic = i < np ? i : i0;  // (a slice may reach beyond the planes: the wavefronts times the slots round up)
      const double a0 = q0[ic], a1 = q1[ic], a2 = q2[ic];
      Ar[j][0] = (MASKED || inb) ? a0 : 1.0; Ar[j][1] = (MASKED || inb) ? a1 : 0.0; Ar[j][2] = (MASKED || inb) ? a2 : 0.0;
    }
  }
  uint32_t c = next_kept(0);
  if (c < it.count) dma_land(it.s2_begin + c);
  uint32_t xpar = 0;  // exchanges of this workgroup so far (see em_pair)
  while (c < it.count) {
    const uint32_t cn = next_kept(c + 1);
    const double m2 = site_sc[0][c], mean2 = site_sc[1][c], rsx2 = site_sc[2][c];
    const Relabel rl = relabel(m1, m2, mean1, mean2);
    const int gb0 = rl.flip2 ? 2 : 0, gb2 = rl.flip2 ? 0 : 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the landing copy issued during the previous pair is complete
    double B[SLOTS][3];
    const double *gsite = A.planes + (uint64_t)(it.s2_begin + c) * A.site_stride + i0;
#pragma unroll
    for (int j = kLand; j < SLOTS; ++j) {  // issued first: they fly while the landed slots are consumed
      B[j][0] = gsite[(uint32_t)gb0 * np + 64 * j];
      B[j][1] = gsite[np + 64 * j];
      B[j][2] = gsite[(uint32_t)gb2 * np + 64 * j];
    }
    const double *lb = reinterpret_cast<const double *>(lds_b) + lane;
#pragma unroll
    for (int j = 0; j < kLand; ++j) {
      B[j][0] = lb[gb0 * (kLandPlane / 8) + 64 * j];
      B[j][1] = lb[(kLandPlane / 8) + 64 * j];
      B[j][2] = lb[gb2 * (kLandPlane / 8) + 64 * j];
    }
    double sxy = 0.0;
    uint32_t vbits = inbits;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      // expected genotypes p1 + 2 p2 (ngsLD.cpp:113); pearson_r runs over ALL individuals (ngsLD.cpp:290); the planes hold
      // zeros beyond the cohort
      if (MASKED) sxy = fma(fma(2.0, Ar[j][2], Ar[j][1]), fma(2.0, B[j][2], B[j][1]), sxy);
      if (MASKED) {  // an individual without data at either site (gen_func.cpp:1089): b = 0, and em_pair_ab pads its s to 1
        const bool ok = ((inbits >> j) & 1u) && !miss_data(Ar[j][0], Ar[j][1], Ar[j][2]) && !miss_data(B[j][0], B[j][1], B[j][2]);
        if (!ok) {
          vbits &= ~(1u << j);
          B[j][0] = 0.0; B[j][1] = 0.0; B[j][2] = 0.0;
        }
      } else {
        if (!((inbits >> j) & 1u)) {  // ghost: (1, 0, 0) at both sites -- expected genotype 0, s = f0^2, nothing added to R[1..8]
          B[j][0] = 1.0; B[j][1] = 0.0; B[j][2] = 0.0;
        }
        sxy = fma(fma(2.0, Ar[j][2], Ar[j][1]), fma(2.0, B[j][2], B[j][1]), sxy);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the landing buffer is consumed, the direct loads are in
    if (cn < it.count) dma_land(it.s2_begin + cn);
    sxy = wave_sum1_bcast(sxy);
    const double centre = (double)A.n_ind * rl.mean1 * rl.mean2;
    uint32_t x = A.n_ind;
    if (!MASKED) {
      if (lane == 0) lds_post(lds_addr(&parked[c][sub]), sxy);
    } else {  // the individuals with data at both sites (gen_func.cpp:1091) and the moment meet before the EM (as pair_ld_kernel)
      x = count_valid<SLOTS>(vbits);
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
    const uint32_t n_iter = em_pair_ab<SLOTS, MASKED, NGSLD_AB_TREE, SLOTS, WAVES>(B, Ar, nullptr, nullptr, nullptr, vbits, nullptr,
                                                                                    MASKED ? 1.0 / (double)x : A.inv_n, rl.m1,
                                                                                    rl.m2, f0, f1, f2, f3, lane, A.status, xch, sub,
                                                                                    &xpar);
    unrelabel(rl.flip1, rl.flip2, f0, f1, f2, f3);
    if (lane == 0 && sub == 0) {
      PairResult &r = res[c];
      r.f[0] = f0; r.f[1] = f1; r.f[2] = f2; r.f[3] = f3;
      r.sxy = MASKED ? sxy : centre;  // (parked partial sums: the centring term travels in their place)
      r.rsx2 = rsx2;
      r.x = x;
      r.n_iter = n_iter;
    }
    c = cn;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the parked partial sums are stores the compiler does not see
  __syncthreads();
  const uint32_t t = threadIdx.x;
  if (t < it.count && ((it.mask >> t) & 1ull)) {
    const PairResult r = res[t];
    double sxy = r.sxy;
    if (!MASKED) {
      sxy = 0.0;  // (in the order of the wavefronts)
      for (int w = 0; w < WAVES; ++w) sxy += parked[t][w];
      sxy -= r.sxy;  // centred: sum e1 e2 - n mean1 mean2
    }
    write_pair(A, rec0 + (uint64_t)__popcll(it.mask & ((1ull << t) - 1ull)), r.f[0], r.f[1], r.f[2], r.f[3], sxy, rsx1,
               r.rsx2, r.x, r.n_iter);
  }
}


// Slots of the row vector a lane keeps in REGISTERS for the whole run (AREG), by slot count, as measured (same box,
// profiles/r03/sweep_areg.txt; the records are the same bits whatever the split): every individual counts -- all of them up to
// 13 slots (641..832: +9 % over re-reading them from LDS in every iteration: 1.40 / 1.38 / 1.29 / 1.21e8 pairs/s at 641 / 704 /
// 768 / 832), six at 14 / 15 slots (b alone is 84 / 90 registers there; at 896 that is +4.6 % over two wavefronts per pair),
// none at 16 (any split spills inside the EM loop: -5..-11 % at 1,000); --ignore_miss_data, whose per-slot pads take two
// registers each: eight up to 12 slots (+5..7.5 %), four at 13 / 14 (+3.5 %), none beyond.
template <int SLOTS>
struct AbRegs {
  static constexpr int kAll = SLOTS <= 13 ? SLOTS : (SLOTS <= 15 ? 6 : 0);
  static constexpr int kMasked = SLOTS <= 12 ? 8 : (SLOTS <= 14 ? 4 : 0);
};

template <int SLOTS>
static hipError_t launch_ab_s(bool masked, const PairArgs &a, hipStream_t stream) {
  const dim3 grid((unsigned)a.n_runs), block(512);
  constexpr int kTree = NGSLD_AB_TREE;
  if (masked)
    hipLaunchKernelGGL((pair_ld_ab_kernel<SLOTS, true, kTree, AbRegs<SLOTS>::kMasked>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((pair_ld_ab_kernel<SLOTS, false, kTree, AbRegs<SLOTS>::kAll>), grid, block, 0, stream, a);
  return hipGetLastError();
}

template <int SLOTS, int WAVES>
static hipError_t launch_abm_sw(bool masked, const PairArgs &a, hipStream_t stream) {
  const dim3 grid((unsigned)a.n_items), block(WAVES * 64);
  if (masked)
    hipLaunchKernelGGL((pair_ld_abm_kernel<SLOTS, WAVES, true>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((pair_ld_abm_kernel<SLOTS, WAVES, false>), grid, block, 0, stream, a);
  return hipGetLastError();
}
template <int SLOTS>
static hipError_t launch_abm_s(int waves, bool masked, const PairArgs &a, hipStream_t stream) {
  switch (waves) {
    case 2: return launch_abm_sw<SLOTS, 2>(masked, a, stream);
    case 4: return launch_abm_sw<SLOTS, 4>(masked, a, stream);
    case 8: return launch_abm_sw<SLOTS, 8>(masked, a, stream);
    default: return hipErrorInvalidValue;
  }
}

// several wavefronts per pair, a/b form (item form: n_items workgroups, as launch_pair_wn)
hipError_t launch_pair_abm(int slots, int waves, bool masked, const PairArgs &a, hipStream_t stream) {
  if (a.n_items == 0) return hipSuccess;
  if (a.n_items > 0x7fffffffull || a.np < (uint32_t)(slots * waves * 64)) return hipErrorInvalidValue;
  switch (slots) {
    case 9: return launch_abm_s<9>(waves, masked, a, stream);
    case 10: return launch_abm_s<10>(waves, masked, a, stream);
    case 11: return launch_abm_s<11>(waves, masked, a, stream);
    case 12: return launch_abm_s<12>(waves, masked, a, stream);
    case 13: return launch_abm_s<13>(waves, masked, a, stream);
    case 14: return launch_abm_s<14>(waves, masked, a, stream);
    case 15: return launch_abm_s<15>(waves, masked, a, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_pair_ab(int slots, bool masked, const PairArgs &a, hipStream_t stream) {
  if (a.n_runs == 0) return hipSuccess;
  if (a.n_runs > 0x7fffffffull) return hipErrorInvalidValue;
  switch (slots) {
    case 9: return launch_ab_s<9>(masked, a, stream);
    case 10: return launch_ab_s<10>(masked, a, stream);
    case 11: return launch_ab_s<11>(masked, a, stream);
    case 12: return launch_ab_s<12>(masked, a, stream);
    case 13: return launch_ab_s<13>(masked, a, stream);
    case 14: return launch_ab_s<14>(masked, a, stream);
    case 15: return launch_ab_s<15>(masked, a, stream);
    case 16: return launch_ab_s<16>(masked, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ngsld
