// ld_em.h -- the building blocks of the pair kernels: the shared reciprocal tree, allele relabelling, staging a pair into
// registers (stage_pair), the EM loop of one pair (em_pair: haplo_freq / pair_freq_iter, gen_func.cpp:1027-1119), and the
// derived statistics with the replay flag (write_pair: ngsLD.cpp:296-306).
#pragma once

#include "ld_common.h"

namespace ngsld {

// ---------------------------------------------------------------------------------------------
// Building blocks of the pair kernels
// ---------------------------------------------------------------------------------------------
// Reciprocals of N positive numbers from ONE reciprocal: products up a binary tree (N - 1 multiplies), 1/root, then
// down again -- the inverse of a node is the parent's inverse times the sibling's product (2 multiplies per inner
// node).  N = 8: 21 multiplies + one refined v_rcp_f64 instead of 8 (or 4, taken in pairs) of the 16-cycle kind.
template <int N>
struct RcpTree {
  static constexpr int L = N / 2;
  static __device__ __forceinline__ double prod(const double *s) {
    return RcpTree<L>::prod(s) * RcpTree<N - L>::prod(s + L);
  }
  static __device__ __forceinline__ void down(const double *s, double inv, double *r) {
    const double pl = RcpTree<L>::prod(s), pr = RcpTree<N - L>::prod(s + L);  // same expressions as in prod(): CSE'd
    RcpTree<L>::down(s, inv * pr, r);
    RcpTree<N - L>::down(s + L, inv * pl, r + L);
  }
};
template <>
struct RcpTree<1> {
  static __device__ __forceinline__ double prod(const double *s) { return s[0]; }
  static __device__ __forceinline__ void down(const double *, double inv, double *r) { r[0] = inv; }
};

struct PairedTag { static constexpr bool value = true; };   // compile-time selectors of em_pair's reciprocal scheme
struct SingleTag { static constexpr bool value = false; };
typedef __attribute__((address_space(3))) void lds_void_t;        // operands of __builtin_amdgcn_global_load_lds
typedef const __attribute__((address_space(1))) void glb_void_t;

// Allele relabelling.  The frequency recovered from the other three carries an ABSOLUTE error of ~1e-16.
// That is harmless for the largest of the four and ruinous for a tiny one: with both sites nearly monomorphic the
// denominators of D' and r2 are products of two small margins (1e-14, say), and 1e-16 in a hap00 of 1e-15 moved D' in
// the third decimal.  So each site's alleles are labelled such that its estimated frequency is <= 1/2 -- a site with
// maf > 1/2 has its genotype planes 0 and 2 read in each other's place -- which makes hap 0 (initially (1-m1)(1-m2) >=
// 1/4) the common-common haplotype, the one that tends to 1 exactly where the conditioning is bad.  The EM is equivariant
// under the relabelling; the frequencies are put back in the caller's order afterwards (k = 2 * allele1 + allele2).
struct Relabel {
  bool flip1, flip2;
  double m1, m2, mean1, mean2;  // frequencies and mean expected genotypes under the new labels
};
__device__ __forceinline__ Relabel relabel(double m1, double m2, double mean1, double mean2) {
  Relabel r;
  r.flip1 = m1 > 0.5;
  r.flip2 = m2 > 0.5;
  r.m1 = r.flip1 ? 1.0 - m1 : m1;
  r.m2 = r.flip2 ? 1.0 - m2 : m2;
  r.mean1 = r.flip1 ? 2.0 - mean1 : mean1;  // expected genotype p1 + 2 p2 of a normalised triple becomes 2 - e
  r.mean2 = r.flip2 ? 2.0 - mean2 : mean2;
  return r;
}
__device__ __forceinline__ void unrelabel(bool flip1, bool flip2, double &f0, double &f1, double &f2, double &f3) {
  if (flip1) {  // allele at site 1: haplotypes k <-> k ^ 2
    double t = f0; f0 = f2; f2 = t;
    t = f1; f1 = f3; f3 = t;
  }
  if (flip2) {  // allele at site 2: k <-> k ^ 1
    double t = f0; f0 = f1; f1 = t;
    t = f2; f2 = f3; f3 = t;
  }
}

// Stage both sites of one pair: P = a (x) b for this lane's SLOTS individuals, their validity bits and the
// Pearson cross moment.  pa / pb point at a site's three planes [3][np] -- in HBM/L2 (direct kernel) or in
// LDS (prefetch kernel); after inlining the compiler knows which and emits global_load or ds_read.
//   UNCENTRED: sxy comes back as the uncentred cross moment sum e1 e2 -- padding lanes hold a == b == 0, so no bounds
//   test -- and the caller subtracts n * mean1 * mean2 once per pair
//   GHOSTS.  A slot that holds no individual -- a padding lane, or under --ignore_miss_data an individual without data at
//   either site -- is staged as P = (1, 0, ..., 0).  In the hot EM step (shared reciprocal, three-value form) such a slot has
//   s = f0^2 -- positive, at least 2^-20 while the pair is in that loop -- so it neither zeroes the lane's product tree nor
//   overflows its reciprocal, and it adds nothing to R[1..8]: r * 0.  R[0] is never accumulated in that form.  The steps
//   that do accumulate R[0] take one reciprocal per individual and skip the slot by its validity bit.  (Rounds 1-2 kept
//   P = 0 and added a per-slot `pad` of 0 / 1 to s: two registers per slot in every kernel that may hold empty slots
//   anywhere -- all of --ignore_miss_data -- which is what spilled there.)
__device__ __forceinline__ const double *uniform_ptr(const double *p) {  // a wavefront-uniform pointer, said so: SGPRs
  const uint64_t v = (uint64_t)(uintptr_t)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return reinterpret_cast<const double *>((uintptr_t)(((uint64_t)hi << 32) | lo));
}

//   A_GLOBAL: pa points into global memory (the multi-wavefront kernels read their slice of the row vector from L2 for
//   every pair): the three plane bases are handed to the loads as SGPR pairs + one 32-bit lane offset.  Left to itself the
//   compiler kept ~10 64-bit VGPR addresses for them, and under --ignore_miss_data, where registers are tightest, SPILLED
//   them -- seven scratch reloads per pair, one after the other into the same register pair, each an L2 round trip in
//   front of the load it feeds: 4.7 us of a 15 us pair at n_ind 2,000
template <int SLOTS, bool MASKED, bool ONLY_LAST = false, bool UNCENTRED = false, bool A_GLOBAL = false>  // ONLY_LAST: only the last slot can hold padding lanes
__device__ __forceinline__ void stage_pair(const double *pa, uint32_t npa, uint32_t ia0, const double *pb, uint32_t npb,
                                           uint32_t ib0, uint32_t ind0, uint32_t n_ind, double mean1, double mean2,
                                           double (&P)[SLOTS][9], uint32_t &vbits, double &sxy, bool flip_a = false,
                                           bool flip_b = false, const double (*a_regs)[3] = nullptr, int n_a_regs = 0) {
  // a_regs (may be null): this lane's first n_a_regs triples of site 1, already relabelled, held in registers by the caller
  // for all the pairs of an item (the row vector is the same for every one of them) -- pa is not read for those slots
  // pa[g * npa + ia0 + 64 j] / pb[g * npb + ib0 + 64 j] hold genotype g of individual ind0 + 64 j (this lane, slot j);
  // flip_a / flip_b (wavefront-uniform) relabel the alleles of a site: genotype planes 0 and 2 trade places (see Relabel)
  vbits = 0;
  sxy = 0.0;
  const double *pa0 = pa + (flip_a ? 2 * npa : 0u), *pa1 = pa + npa, *pa2 = pa + (flip_a ? 0u : 2 * npa);
  const double *pb0 = pb + (flip_b ? 2 * npb : 0u), *pb2 = pb + (flip_b ? 0u : 2 * npb);
  if (A_GLOBAL) {
    pa0 = uniform_ptr(pa0); pa1 = uniform_ptr(pa1); pa2 = uniform_ptr(pa2);
  }
  // kByCount (several wavefronts per pair, every individual counts): a wavefront's slots are full up to a wavefront-
  // uniform slot n_full, at most ONE slot is partly filled (lanes below rem), the rest are empty -- validity bits and ghosts
  // come from those two numbers instead of a compare and a select per slot: inside the loop below these held nine compare
  // masks and select temporaries beside the loads in flight and cost 64 bytes of scratch per lane (2 x 9 slots: -14 % pairs/s)
  constexpr bool kByCount = !MASKED && !ONLY_LAST;
  const uint32_t lane_in_wave = ind0 & 63u;
  uint32_t n_full = 0, rem = 0;
  if (kByCount) {
    const uint32_t first = ind0 - lane_in_wave;  // this wavefront's first individual
    const uint32_t have = n_ind > first ? n_ind - first : 0u;
    n_full = (uint32_t)__builtin_amdgcn_readfirstlane((int)(have >> 6 < (uint32_t)SLOTS ? have >> 6 : (uint32_t)SLOTS));
    rem = (uint32_t)__builtin_amdgcn_readfirstlane((int)(have >> 6 < (uint32_t)SLOTS ? have & 63u : 0u));
    vbits = ((1u << n_full) - 1u) | ((lane_in_wave < rem ? 1u : 0u) << n_full);
  }
#pragma unroll
  for (int j = 0; j < SLOTS; ++j) {
    const uint32_t ia = ia0 + (uint32_t)j * 64, ib = ib0 + (uint32_t)j * 64;
    typedef const __attribute__((address_space(1))) double gdouble_t;  // (said to be global memory: global_load, not flat_load)
    const bool in_regs = a_regs != nullptr && j < n_a_regs;
    const double a0 = in_regs ? a_regs[j < n_a_regs ? j : 0][0] : (A_GLOBAL ? ((gdouble_t *)pa0)[ia] : pa0[ia]),
                 a1 = in_regs ? a_regs[j < n_a_regs ? j : 0][1] : (A_GLOBAL ? ((gdouble_t *)pa1)[ia] : pa1[ia]),
                 a2 = in_regs ? a_regs[j < n_a_regs ? j : 0][2] : (A_GLOBAL ? ((gdouble_t *)pa2)[ia] : pa2[ia]);
    const double b0 = pb0[ib], b1 = pb[npb + ib], b2 = pb2[ib];
    const bool inb = kByCount ? ((vbits >> j) & 1u) != 0
                              : ((ONLY_LAST && j < SLOTS - 1) ? true : ind0 + (uint32_t)j * 64 < n_ind);
    if (!kByCount) {
      bool ok = inb;
      if (MASKED) ok = inb && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);  // gen_func.cpp:1089
      vbits |= (ok ? 1u : 0u) << j;
    }
    double z0 = a0, z1 = a1, z2 = a2;
    if (MASKED) {  // an individual without data: P = (1, 0, ..., 0)
      const double keep = ((vbits >> j) & 1u) ? 1.0 : 0.0;
      z0 = a0 * keep; z1 = a1 * keep; z2 = a2 * keep;
      P[j][0] = fma(z0, b0, 1.0 - keep);
    } else if (ONLY_LAST && j == SLOTS - 1) {  // padding lanes hold zeros in the planes already
      P[j][0] = fma(a0, b0, inb ? 0.0 : 1.0);
    } else {                                   // (kByCount: the ghosts are put in after the loop)
      P[j][0] = a0 * b0;
    }
    P[j][1] = z0 * b1; P[j][2] = z0 * b2;
    P[j][3] = z1 * b0; P[j][4] = z1 * b1; P[j][5] = z1 * b2;
    P[j][6] = z2 * b0; P[j][7] = z2 * b1; P[j][8] = z2 * b2;
    // expected genotypes p1 + 2*p2 (ngsLD.cpp:113); pearson_r runs over ALL individuals (ngsLD.cpp:290)
    if (UNCENTRED && !MASKED) {  // (a1 + 2 a2)(b1 + 2 b2) = P4 + 2 P5 + 2 P7 + 4 P8 (measured 0.3 % faster than from a, b)
      sxy += fma(4.0, P[j][8], fma(2.0, P[j][5] + P[j][7], P[j][4]));
    } else if (UNCENTRED) {      // P of individuals without data is zeroed: take the moment from a and b
      sxy = fma(fma(2.0, a2, a1), fma(2.0, b2, b1), sxy);
    } else {
      const double c1 = inb ? fma(2.0, a2, a1) - mean1 : 0.0;
      const double c2 = inb ? fma(2.0, b2, b1) - mean2 : 0.0;
      sxy = fma(c1, c2, sxy);
    }
  }
  if (kByCount) {  // ghosts behind scalar branches: full wavefronts -- all but a pair's last -- skip every one of them
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      if (n_full <= (uint32_t)j) {
        const bool keep = n_full == (uint32_t)j && lane_in_wave < rem;
        P[j][0] = keep ? P[j][0] : 1.0;
      }
    }
  }
}

// x = individuals with data (gen_func.cpp:1091): popcount of ballots, integer exact
template <int SLOTS>
__device__ __forceinline__ uint32_t count_valid(uint32_t vbits) {
  uint32_t x = 0;
#pragma unroll
  for (int j = 0; j < SLOTS; ++j) x += (uint32_t)__popcll(__ballot((vbits >> j) & 1u));
  return x;
}

// haplo_freq (gen_func.cpp:1027-1059) on the staged pair.  Returns n_iter; f0..f3 hold hap_freq on exit.
//   vbits:     bit j = slot j of this lane holds an individual that counts (not a ghost, see stage_pair)
//   WAVES > 1: the pair is spread over WAVES wavefronts, partial sums meet in xch (LDS, double buffered)
//   xpar:      (WAVES > 1) the caller's count of exchanges so far: its parity picks the half of xch an exchange uses.  Carried
//              from pair to pair, consecutive exchanges alternate whatever the iteration counts were -- no barrier is needed
//              between the last exchange of one pair and the first of the next
// Reciprocals.  ALL slots of a lane share one v_rcp_f64 (RcpTree); ghost slots take part with s = f0^2.  s lies in (0, 1];
// the product of SLOTS values can underflow (all below ~1e-38 for eight slots), and that -- like any other non-finite
// outcome -- is caught by the sanity test on the new frequencies, after which the iteration is redone with one reciprocal
// per individual before anything is concluded from it.  (One reciprocal per individual, and one per two individuals, were
// the earlier forms: -9 % and -4 % against the tree at eight slots.)
template <int SLOTS, int WAVES>
__device__ __forceinline__ uint32_t em_pair(const double (&P)[SLOTS][9], uint32_t vbits, double inv_x, double m1,
                                            double m2, double &f0, double &f1, double &f2, double &f3,
                                            double (*xch)[WAVES][4], int sub, int lane, int *status,
                                            uint32_t *xpar = nullptr) {
  static_assert(WAVES == 1 || WAVES == 2 || WAVES == 4 || WAVES == 8, "em_pair: 1, 2, 4 or 8 wavefronts per pair");
  f0 = (1 - m1) * (1 - m2); f1 = (1 - m1) * m2; f2 = m1 * (1 - m2); f3 = m1 * m2;  // gen_func.cpp:1034-1037
  if (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1) {  // error() in the reference (:1030); reported through status
    if (lane == 0 && sub == 0) atomicExch(status, (int)NGSLD_ERR_MAF_RANGE);
    f0 = f1 = f2 = f3 = __builtin_nan("");
  }
  // f = ff/(2x) (gen_func.cpp:1108-1109).  The renormalisation that follows there (:1112-1113) divides
  // by sum_k ff_k/(2x) = (1/x) sum_i s_i/s_i = 1 up to rounding, and the EM map does not depend on the
  // scale of f, so it is not repeated per iteration.  inv_x = 1/x; x == 0 gives 0 * inf = NaN like the reference's 0/0.
  // (held in a VGPR: the four products t_k * inv_x below take t_k from SGPRs, and a VALU op reads one SGPR at most)
  asm("" : "+v"(inv_x));
  bool bad = false, tie = false;
  uint32_t n_iter = 0;
  constexpr bool kTree = SLOTS > 1;
  constexpr bool kScaled = WAVES == 1;  // (several wavefronts per pair: partial sums are scaled after they met)
  // tree_tag: the step with the shared reciprocal, or with one reciprocal per individual.  drop_tag: the step in its
  // three-value form (hap 0 recovered from the sum) or in the full four-value form.  The shared-reciprocal step only exists
  // in the three-value form; the step with one reciprocal per individual, which only ever runs outside the hot loop, in
  // the full form -- and, where several wavefronts share a pair, in the three-value form too (all of them have to
  // exchange the same values, and some take this step in every iteration).
  auto em_step = [&](auto tree_tag, auto drop_tag, double &n0, double &n1, double &n2, double &n3) {
    constexpr bool kShared = decltype(tree_tag)::value;
    constexpr bool kDrop = decltype(drop_tag)::value;
    static_assert(!kShared || (kTree && kDrop), "the shared-reciprocal step: several slots, three-value form");
    // products f_k f_h: they build the two-locus genotype weights W (s = sum_G W[G] P[G] is the
    // reference's 16-term `sum`, gen_func.cpp:1093-1096) and are reused by the t_k contraction below
    const double p00 = f0 * f0, p01 = f0 * f1, p02 = f0 * f2, p03 = f0 * f3, p11 = f1 * f1;
    const double p12 = f1 * f2, p13 = f1 * f3, p22 = f2 * f2, p23 = f2 * f3, p33 = f3 * f3;
    const double w1 = p01 + p01, w3 = p02 + p02, w4 = 2.0 * (p03 + p12), w5 = p13 + p13, w7 = p23 + p23;
    double R0 = 0, R1 = 0, R2 = 0, R3 = 0, R4 = 0, R5 = 0, R6 = 0, R7 = 0, R8 = 0;
    if (NGSLD_SETPRIO && kShared) __builtin_amdgcn_s_setprio(NGSLD_PRIO_S);  // the dense s sums start here
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
    if constexpr (kShared) {
      double sv[SLOTS], rv[SLOTS];
#pragma unroll
      for (int j = 0; j < SLOTS; ++j) sv[j] = slot_s(j);
      // Two wavefronts share a SIMD.  The one inside a serial stretch of its iteration (reciprocal tree; contraction,
      // reduction, convergence test and the next f products) has one instruction ready at a time and every cycle it
      // waits for the issue slot lengthens its critical path; the one inside a dense stretch (the s and R sums) has
      // dozens ready.  Priority goes to the former.
      if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(NGSLD_PRIO_TREE);
      // kScaled: 1/x rides on the root inverse, so every R -- and with them the three t_k -- come out divided by x
      double inv = rcp_refined(RcpTree<SLOTS>::prod(sv));
      if (kScaled) inv *= inv_x;
      RcpTree<SLOTS>::down(sv, inv, rv);
      if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(NGSLD_PRIO_R);
#pragma unroll
      for (int j = 0; j < SLOTS; ++j) slot_acc(j, rv[j]);
      if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(NGSLD_PRIO_SERIAL);
    } else {
#pragma unroll
      for (int j = 0; j < SLOTS; ++j) {
        if ((vbits >> j) & 1u) slot_acc(j, rcp_refined(slot_s(j)));  // (ghost slots are skipped: this form accumulates R[0])
      }
    }
    // t_k = sum_h f_k f_h R[G(k,h)]  (= this lane's share of ff_k / 2, gen_func.cpp:1098-1104)
    double t0 = kDrop ? 0.0 : fma(p03, R4, fma(p02, R3, fma(p01, R1, p00 * R0)));
    double t1 = fma(p13, R5, fma(p12, R4, fma(p11, R2, p01 * R1)));
    double t2 = fma(p23, R7, fma(p22, R6, fma(p12, R4, p02 * R3)));
    double t3 = fma(p33, R8, fma(p23, R7, fma(p13, R5, p03 * R4)));
    if (WAVES > 1 && kDrop) {
      // several wavefronts per pair, three-value form: the row totals go to the exchange buffer from the lanes that hold
      // them (no v_readlane, no copies back to VGPRs), and every wavefront adds the partials up in the same order -- the
      // new frequencies must be the same bit pattern in all of them, they decide together when to leave the loop.
      // The LDS accesses are assembly (lds_post / lds_gather): the compiler must not order them behind the slice copy in flight.
      const double w = wave_sum3_rows(t1, t2, t3);
      const int par = (int)((*xpar)++ & 1u);
      const int row = lane >> 4;
      // layout of one parity's buffer (WAVES * 32 bytes): [value k = 0..2][wavefront] -- a value's partials side
      // by side, WAVES / 2 reads of 16 bytes each; added up in the order of the wavefronts
      const uint32_t base = lds_addr(&xch[par][0][0]);
      if ((lane & 15) == 0 && row != 1)  // rows 0 / 2 / 3 hold t1 / t2 / t3 (wave_sum3_rows)
        lds_post(base + (uint32_t)((row == 0 ? 0 : row - 1) * WAVES + sub) * 8u, w);
      lds_barrier();
      if constexpr (WAVES == 8) {
        // 24 partials: lane l < 24 reads partial l (value l / 8 of wavefront l % 8), three DPP steps add the eight of a value
        // inside their eight lanes -- a fixed tree, the same in every wavefront -- and lanes 0 / 8 / 16 hand the totals out
        // (12 reads of 16 bytes per lane -- 48 registers of partials in flight -- lost 4.5 % at n_ind 4000)
        double v;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(base + (uint32_t)(lane & 31) * 8u) : "memory");
        v += dpp_mov<0xB1>(v);   // quad_perm:[1,0,3,2]
        v += dpp_mov<0x4E>(v);   // quad_perm:[2,3,0,1]
        v += dpp_mov<0x141>(v);  // row_half_mirror: lane l <-> 7 - l inside each 8 lanes
        t1 = read_lane(v, 0); t2 = read_lane(v, 8); t3 = read_lane(v, 16);
      } else {
        constexpr int kHalf = WAVES > 1 ? WAVES / 2 : 1;  // (one wavefront per pair: instantiated, never run)
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
      if (kDrop)
        wave_sum3(t1, t2, t3);
      else
        wave_sum4(t0, t1, t2, t3);
      if (WAVES > 1) {  // (the full four-value form of a pair spread over several wavefronts: rare, plain LDS accesses)
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
    }
    const bool scaled = kShared && kScaled;
    n1 = scaled ? t1 : t1 * inv_x; n2 = scaled ? t2 : t2 * inv_x; n3 = scaled ? t3 : t3 * inv_x;
    // sum_k ff_k / (2x) = 1 (every individual's four posterior weights add up to one): in the three-value form the first
    // frequency is what the other three leave, R[0] is never accumulated and three values go through the reduction
    // instead of four
    n0 = kDrop ? 1.0 - ((n1 + n2) + n3) : t0 * inv_x;
  };
  // Any individual with s == 0 makes every tmp/sum NaN in the reference, hence all four f NaN and, as a
  // NaN difference never raises eps (gen_func.cpp:1049-1053), "convergence" at this iteration.  Here
  // s == 0 poisons every R with inf/NaN -- fma(P, NaN, R) -- so any one accumulated frequency that is not a sane
  // value below 2 <=> the reference is all NaN.
  // The hot loops hold the one-reciprocal step and nothing else: a step that does not look sane leaves its loop, is
  // redone with one reciprocal per individual (an underflowed product has to be ruled out before anything is
  // concluded), and the loop is entered again -- with a single definition of the new frequencies per trip the compiler
  // carries them from one iteration to the next without register copies.
  // Two forms of the step.  The three-value form leaves hap 0 with an ABSOLUTE error of ~1e-16, which is nothing while
  // hap 0 is a sizeable frequency (the allele relabelling makes it the common-common haplotype) and too much once it is
  // tiny: its update is multiplicative, so a relative error stays for good, and the denominators of D' and r2 can be
  // products of two small margins.  Below kFullBelow the pair therefore leaves the hot loop for good and finishes in
  // the full four-value form (one reciprocal per individual: slower, and rare): hap 0 then keeps the relative accuracy
  // it had at the switch (1e-16 / kFullBelow ~ 1e-13).
  constexpr double kFullBelow = 0x1p-10;
  bool full = __builtin_amdgcn_ballot_w64(f0 < kFullBelow) != 0;  // wave-uniform (f is)
  bool done = false;
  while (!done && n_iter < (uint32_t)kIterMax) {
    if constexpr (kTree) {
      if (!full) {
        for (; n_iter < (uint32_t)kIterMax; ++n_iter) {
          double n0, n1, n2, n3;
          em_step(PairedTag(), PairedTag(), n0, n1, n2, n3);  // (the tags double as true / false)
          if (__builtin_amdgcn_ballot_w64(!(n1 < 2.0))) break;  // an odd step (wave-uniform values: all-or-nothing)
          // eps = the largest of the four changes (gen_func.cpp:1049-1053) is at least the change of hap 1: while that one
          // alone is above EPSILON -- nine iterations in ten -- the other three differences are not formed (+1.2 %)
          bool conv = false;
          if (__builtin_amdgcn_ballot_w64(fabs(n1 - f1) < kEpsilonTie)) {
            const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
            conv = __builtin_amdgcn_ballot_w64(eps < kEpsilon) != 0;  // gen_func.cpp:1054-1055
            tie |= __builtin_amdgcn_ballot_w64(fabs(eps - kEpsilon) < kTieMargin) != 0;  // too close to call: replayed
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
        // an odd step: on to the second opinion
        if (WAVES > 1) lds_barrier();  // n1 is the same in every wavefront: all redo, none still reads the exchange buffer
      }
    }
    // one iteration with one reciprocal per individual, four-value form: the kernels' only path where any slot may be
    // empty, the second opinion on an odd step, and how a pair with a tiny hap 0 finishes
    double n0, n1, n2, n3;
    if (WAVES == 1 || full)
      em_step(SingleTag(), SingleTag(), n0, n1, n2, n3);
    else
      em_step(SingleTag(), PairedTag(), n0, n1, n2, n3);
    if (__builtin_amdgcn_ballot_w64(!(n1 < 2.0))) {
      bad = true;
      break;
    }
    const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
    f0 = n0; f1 = n1; f2 = n2; f3 = n3;
    tie |= __builtin_amdgcn_ballot_w64(fabs(eps - kEpsilon) < kTieMargin) != 0;
    if (__builtin_amdgcn_ballot_w64(eps < kEpsilon)) break;
    ++n_iter;
    if (!full && __builtin_amdgcn_ballot_w64(f0 < kFullBelow)) full = true;  // (wavefronts without a hot loop)
  }
  if (bad) f0 = f1 = f2 = f3 = __builtin_nan("");
  return n_iter | (tie ? kTieBit : 0u);
}

// Is v closer than d to a point where "%f" (six decimals) rounds the other way?  NaN / inf: no.
__device__ __forceinline__ bool near_rounding(double v, double d) {
  const double t = fabs(v) * 1e6;
  return fabs((t - floor(t)) - 0.5) < d * 1e6;
}

// ngsLD.cpp:296-306 (hap-derived maf, D, D', r2) + pearson_r, one record per pair; pairs whose outcome the reference's
// rounding decides are flagged for the exact-order replay (see kHapNoise / kReplayFloor above).
__device__ __forceinline__ void write_pair(const PairArgs &A, uint64_t slot, double f0, double f1, double f2,
                                           double f3, double sxy, double rsx1, double rsx2, uint32_t x,
                                           uint32_t n_iter) {
  const bool tie = (n_iter & kTieBit) != 0;
  n_iter &= ~kTieBit;
  const double hm0 = 1 - (f0 + f1);
  const double hm1 = 1 - (f0 + f2);
  const double D = f0 * f3 - f1 * f2;
  const double q00 = hm0 * hm1, q11 = (1 - hm0) * (1 - hm1);
  const double q01 = hm0 * (1 - hm1), q10 = (1 - hm0) * hm1;
  const double den = D < 0 ? -(q00 <= q11 ? q00 : q11) : (q01 <= q10 ? q01 : q10);
  const double Dp = D / den;
  const double rr = D / sqrt(hm0 * hm1 * (1 - hm0) * (1 - hm1));
  // a constant site (rsx = 1/sqrt(0) = inf) is 0/0 = NaN in gsl_stats_correlation; said explicitly because a cross
  // moment centred after the fact (run kernel) is only ~0 there, not exactly 0.
  // (a negative rsx: a site with a triple that does not sum to 1 whose alleles the kernels relabelled -- their Pearson moment
  // assumes e' = 2 - e there: ld_prep.hip, signed_rsx; such pairs are replayed)
  const double a1 = fabs(rsx1), a2 = fabs(rsx2);
  const bool odd_site = rsx1 < 0 || rsx2 < 0;
  const bool constant_site = a1 == __builtin_inf() || a2 == __builtin_inf();
  const double r = constant_site ? __builtin_nan("") : sxy * a1 * a2;
  ngsld_rec_std o;
  o.r2_ExpG = ref_nan(r * r);
  o.D = ref_nan(D);
  o.Dp = ref_nan(Dp);
  o.r2 = ref_nan(rr * rr);
  A.out_std[slot] = o;
  if (A.out_ext != nullptr) {
    ngsld_rec_ext e;
    e.hap[0] = ref_nan(f0); e.hap[1] = ref_nan(f1); e.hap[2] = ref_nan(f2); e.hap[3] = ref_nan(f3);
    e.n_ind_data = x;
    e.n_iter = n_iter;
    A.out_ext[slot] = e;
  }
  if (A.flags != nullptr) {
    const double q0 = fabs(hm0) <= fabs(1 - hm0) ? fabs(hm0) : fabs(1 - hm0);
    const double q1 = fabs(hm1) <= fabs(1 - hm1) ? fabs(hm1) : fabs(1 - hm1);
    // (NaN frequencies fail both comparisons)
    // host_only: reasons that concern r2_ExpG -- what the device-side replay of called genotypes (ld_replay.hip) leaves alone
    const bool pearson_bad = !constant_site && (double)A.n_ind * a1 * a2 > kPearsonCond;  // (a constant site: NaN on every path)
    bool host_only = odd_site || (pearson_bad && !A.pearson_on_device);
    // (written so that a NaN anywhere -- frequencies, D', r2 -- flags the pair)
    const double amp_q = 1.0 / q0 + 1.0 / q1, big = fabs(Dp) >= o.r2 ? fabs(Dp) : o.r2;
    bool flag = tie || host_only || pearson_bad || !(q0 >= kReplayFloor) || !(q1 >= kReplayFloor) || !(kHapNoise * amp_q * big <= kRecordTol);
    // The TSV prints six decimals (ngsLD.cpp:314-349).  A value that sits on a rounding point of the sixth decimal --
    // closer to it than this kernel and the reference can differ -- would print a different last digit, and a D within
    // rounding noise of zero a different sign ("-0.000000"): those pairs are replayed too, so that the text is the
    // reference's byte for byte.  Error bounds: hap, hap_maf and D are absolute (a few ulp of 1); D' and r2 divide by
    // products of the margins q (relative error ~ulp / q); r2_ExpG carries ~ulp * n * rsx1 * rsx2 of cancellation.
    // (flag_text: only where the records may become text -- ngsld_run; ngsld_run_device leaves them on the device)
    constexpr double kUlp = 0x1p-52;
    const double d_abs = A.flag_text ? 32 * kUlp : -1.0;  // (negative: near_rounding is never true)
    const double amp = A.flag_text ? amp_q : 0.0;
    // (a pair whose moment is ill conditioned: this r2_ExpG is not the number to look at -- whoever replays the pair does)
    host_only = host_only || (!pearson_bad && near_rounding(o.r2_ExpG, 0.5 * d_abs * (1.0 + 4.0 * (double)A.n_ind * a1 * a2)));
    flag = flag || host_only || fabs(D) < 2 * d_abs || near_rounding(D, d_abs) ||
           near_rounding(Dp, 2 * d_abs * (1.0 + fabs(Dp) * amp)) || near_rounding(o.r2, 2 * d_abs * (1.0 + o.r2 * amp));
    if (A.out_ext != nullptr)
      flag = flag || near_rounding(f0, d_abs) || near_rounding(f1, d_abs) || near_rounding(f2, d_abs) ||
             near_rounding(f3, d_abs) || near_rounding(hm0, d_abs) || near_rounding(hm1, d_abs);
    if (flag) {
      atomicOr(&A.flags[flag_head_words(A.flag_cap) + (slot >> 5)], 1u << (slot & 31u));
      if (host_only && A.flags_host != nullptr) {
        atomicOr(&A.flags_host[slot >> 5], 1u << (slot & 31u));
        const uint32_t kh = atomicAdd(&A.flags[1], 1u);
        if (kh < kFlagHostCap) reinterpret_cast<uint64_t *>(A.flags + kFlagListAt + 2u * A.flag_cap)[kh] = slot;
      }
      const uint32_t k = atomicAdd(&A.flags[0], 1u);
      if (k < A.flag_cap) reinterpret_cast<uint64_t *>(A.flags + kFlagListAt)[k] = slot | (host_only ? kFlagHostOnly : 0ull);
    }
  }
}

// What the EM leaves behind for one pair.  The derived statistics (write_pair: ~100 wavefront-uniform f64
// instructions with two divisions and a square root) are not computed by the wavefront that ran the EM -- there they
// would cost a full instruction issue each for ONE pair -- but once per work item, one LANE per pair.
struct PairResult {
  double f[4], sxy, rsx2;
  uint32_t x, n_iter;
};

}  // namespace ngsld
