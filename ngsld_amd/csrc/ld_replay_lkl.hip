// ld_replay_lkl.hip -- exact-order replay ON THE DEVICE for genotype-LIKELIHOOD matrices (gfx950).
//
// Why.  The pair kernels flag the pairs whose outcome the reference's own rounding decides (ld_device.h, write_pair).  On
// SNP-called input that is a few dozen pairs per 10^8 and the host replays them.  On matrices that are NOT SNP-called -- the
// input the reference's README.md:73 warns about and its own examples/test.sh feeds -- every pair with a (nearly)
// monomorphic site is such a pair: 35-40 % of the pairs at 20 % monomorphic sites.  Replayed on host threads at the
// reference's own speed that was 2.2e6 pairs/s for a pass whose kernel runs 2.2e8 (profiles/r05/before).
//
// What.  The reference's evaluation of one pair (gen_func.cpp:1027-1119, ngsLD.cpp:296-306) is sequential over the
// individuals only in ONE respect: the four running sums ff[k] += tmp_k / sum.  Everything an individual contributes to them
// -- its 16-term `sum`, its four `tmp_k`, the four IEEE divisions -- depends on that individual and on f alone.  So a
// WAVEFRONT owns a flagged pair: lane l holds individuals l, l + 64, ... (both sites' triples in registers for the whole pair,
// loaded once, coalesced), computes their four quotients in the reference's own operation order -- products left to right,
// no fused multiply-add (-ffp-contract=off), IEEE division -- and parks them in LDS in individual order; four lanes then
// add them up one by one, in the reference's order (the chain: 4 x n_ind dependent additions per iteration, the one part
// that cannot be spread over the lanes); then ff / (2x), the SEQUENTIAL renormalisation, haplo_freq's convergence test,
// and after the loop D, D', r2 with a correctly rounded square root.  Same bits as the reference's loop.
//
// The inputs must be the reference's bits too: normal-space likelihoods as the HOST's libm leaves them (log -> post_prob ->
// exp, read_data.cpp:37-45, ngsLD.cpp:110) and est_maf from its sequential loop (gen_func.cpp:974-1009).  The engine keeps
// those in an "exact store" on the device (engine_replay.hip: built by host threads from the caller's raw values the first
// time a run flags more pairs than the host should replay; ngsld_set_geno_lkl input IS such a store already), laid out
// like the planes of the pair kernels: [site][genotype][np].
//
// r2_ExpG is left as the pair kernel wrote it, with one exception.  GSL's long double recurrence has no device twin, so bit
// equality with it is the host's business -- but a pair flagged because the pair kernel's cross moment is ILL CONDITIONED
// (both sites nearly constant: 1 / (std1 std2) > 2^13, every pair of two monomorphic sites of deep data) needs no bit
// equality, only a sound evaluation: two passes over the exact expected genotypes (mean, then centred sums), which agree
// with the recurrence to ~n 2^-64 (2 + mean1 / std1 + mean2 / std2) -- inside 1e-9 as long as neither site's mean / std
// exceeds 2^20.  Those are settled here; the rest (mean / std beyond that, or a value that lands on a sixth-decimal rounding
// point of a text run) are ADDED to the host-only bitmap and list for the host.  Pairs that were host-only from the start
// (PairArgs::flags_host) are never touched.
//
// Work distribution.  No list: the launch's flag BITMAP is the work queue.  A team (one wavefront per pair up to 512
// individuals, 2 / 4 / 8 beyond) claims chunks of kChunkWords words with one atomic, walks their set bits and maps a record
// index to its pair through a cursor over the plan's items (one binary search per chunk, then steps) -- records of one chunk
// are neighbours in (s1, s2) order, so the row's vector stays in registers from pair to pair.
#include "ld_common.h"
#include "ld_replay.h"

#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

#include <hipcub/hipcub.hpp>

namespace ngsld {
namespace {

constexpr double kEps = 1e-5;  // EPSILON, gen_func.hpp:16
constexpr int kMaxIter = 100;  // ITER_MAX, gen_func.hpp:18
constexpr int kMaxSlots = 8;   // individuals per lane, at most (both sites' triples: 6 doubles each, in registers)
constexpr double kMeanOverStd = 0x1p20;  // r2_ExpG on the device only where GSL's own recurrence is good to 1e-9
constexpr uint32_t kChunkWords = 4;  // 128 records per claim at most (launch_replay_lkl: fewer for small launches)

__device__ __forceinline__ double ref_abs(double x) { return x >= 0 ? x : -x; }           // gen_func.hpp:21-23: macros
__device__ __forceinline__ double ref_min(double a, double b) { return a <= b ? a : b; }

// genotype at site 1 / site 2 of the haplotype pair (h, k): bit 1 = allele at site 1, bit 0 = allele at site 2
__device__ __forceinline__ constexpr int geno1(int h, int k) { return ((h >> 1) & 1) + ((k >> 1) & 1); }
__device__ __forceinline__ constexpr int geno2(int h, int k) { return (h & 1) + (k & 1); }

// gen_func.cpp:862-868 on a normal-space triple (pair_freq_iter's use of it, :1089)
__device__ __forceinline__ bool no_data(const double (&g)[3]) { return ref_abs(g[0] - g[1]) < kEps && ref_abs(g[1] - g[2]) < kEps; }

// One individual's four quotients tmp_k / sum exactly as pair_freq_iter forms them (gen_func.cpp:1092-1104) -- the reference's
// expression trees, with what they share taken out.  Every rewrite below is EXACT (the same correctly rounded operations on the
// same operands, or an operation whose result IEEE arithmetic fixes without rounding):
//   * f[k] * f[h] is the leftmost product of every term (C associates left to right) and does not depend on the individual:
//     formed once per iteration (FreqProducts), f[k] * f[h] == f[h] * f[k];
//   * inside the parentheses of `tmp`, p[G1(h,k)] * q[G2(h,k)] and p[G1(k,h)] * q[G2(k,h)] are the same product (G1 and G2 are
//     symmetric), and x + x == 2 x without rounding; (f[k] f[h]) * (2 J) and (2 f[k] f[h]) * J are the same real number rounded
//     once (doubling is exact): the doubled products are formed once per iteration too;
//   * `sum = 0; sum += t0` and `tmp = 0; tmp += t0`: 0 + t0 == t0 for every t0 >= +0 or NaN, and the terms are products of
//     likelihoods and frequencies -- never negative, never -0.
// 9 + 20 + 10 multiplies, 15 + 12 additions, 4 divisions per individual and iteration (the literal loops: 9 + 32 + 16
// multiplies, 16 + 16 + 16 additions); the records are the same bits (tests/test_gpu_replay_lkl.py: device against host replay).
struct FreqProducts {
  double g[10];  // f[k] * f[h] for k <= h (twice that where the reference doubles: 2 * g[.], exact, formed where it is used)
  __device__ __forceinline__ static constexpr int at(int k, int h) {  // index of the unordered pair {k, h}
    return (k <= h) ? (k * 4 - k * (k - 1) / 2 + (h - k)) : (h * 4 - h * (h - 1) / 2 + (k - h));
  }
  __device__ __forceinline__ void set(const double (&f)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int h = k; h < 4; ++h) {
        g[at(k, h)] = f[k] * f[h];
      }
  }
};

// An operand of an f64 division that V_DIV_SCALE_F64 hands through unchanged and without raising VCC, whatever the other
// operand, provided that one passes this test too: a positive number in [2^-600, 2^100) -- the denominator is then neither
// denormal nor is its reciprocal, the exponents differ by less than 768, the quotient is no denormal, and the numerator's
// biased exponent is above 53 (the list of cases in the instruction set manual's description of V_DIV_SCALE_F64) -- or, for a numerator
// (ZERO_OK), +0 exactly: n * r, fma(-d, +0, +0) and fma(+0, r, +0) are +0, which is what V_DIV_FIXUP_F64 makes of 0 / d.
// Negative numbers, -0, NaN, infinities, denormals and anything tiny or huge fail: such an individual's quotients are the
// compiler's own divisions.
template <bool ZERO_OK>
__device__ __forceinline__ int div_operand_plain(double v) {
  constexpr uint32_t kHiMin = (1023u - 600u) << 20, kHiSpan = 700u << 20;
  // (ints and | on purpose: as bools with || the compiler made a branch of every test)
  const int in_range = (uint32_t)__double2hiint(v) - kHiMin < kHiSpan ? 1 : 0;
  return ZERO_OK ? (in_range | (__double_as_longlong(v) == 0 ? 1 : 0)) : in_range;
}

// What quotients<LIGHT> asks of its operands (see there), as one number per value: how far below 1 a likelihood / a frequency
// lies -- E = -floor(log2 v) for an ordinary number in (0, 2), 0 for +0 (a zero factor makes a zero term, which every test allows)
// -- and kNotPlain for everything else (-0, negative numbers, NaN, infinities, denormals, 2 and above).
constexpr uint32_t kNotPlain = 4096;
__device__ __forceinline__ uint32_t plain_depth(double v) {
  const uint32_t u = (uint32_t)__double2hiint(v) >> 20;  // sign and biased exponent
  if (__double_as_longlong(v) == 0) return 0;
  return (u - 1u < 1023u) ? 1023u - u : kNotPlain;          // (u == 0: a denormal)
}

// (counts: the caller uses this lane's quotients -- a padding lane, or an individual left out under --ignore_miss_data, whose
// result is thrown away, must not send its wavefront down the slow way)
// LIGHT: the caller vouches for the operands -- with Ea / Eb the largest plain_depth of a likelihood of either site (looked at
// once per site, where the store is transposed: ReplayLklArgs::xdepth) and Ef that of the step's four frequencies, 2 Ef + Ea +
// Eb <= 596: every product f f p q is then +0 or at least 2^-600 (and below 16), so are `sum` (16 such terms) and every `tmp`
// (4 terms of 2 f f p q): all div_operand_plain but for a `sum` of exactly zero (0 / 0 in the reference), the one test left per
// individual -- 1 instruction instead of 16.
template <bool LIGHT = false>
__device__ __forceinline__ void quotients(const FreqProducts &F, const double (&p)[3], const double (&q)[3], double (&out)[4],
                                          bool counts = true) {
  double J[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) J[a][b] = p[a] * q[b];
  // gen_func.cpp:1093-1096: sum += f[k] * f[h] * p[0][G1] * p[1][G2], k outer, h inner
  double t[10];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int h = k; h < 4; ++h) t[FreqProducts::at(k, h)] = F.g[FreqProducts::at(k, h)] * p[geno1(k, h)] * q[geno2(k, h)];
  double sum = t[FreqProducts::at(0, 0)];
#pragma unroll
  for (int kh = 1; kh < 16; ++kh) sum += t[FreqProducts::at(kh >> 2, kh & 3)];
  // gen_func.cpp:1098-1104
  // LIGHT: HALF the reference's tmp -- sum_h (f f) J instead of sum_h (2 f f) J.  Doubling commutes with every rounding where
  // nothing is denormal (each term, each partial sum, the quotient by `sum` are exactly half the reference's), which the LIGHT
  // bound guarantees (every product at least 2^-600); the caller adds the halves up and divides by x instead of 2 x at the end of
  // the step -- the same real number, rounded once.  Ten registers of doubled products less per lane.
  // (the general form doubles J, not the products of frequencies: (f f) (2 J) and (2 f f) J are the same real number rounded once,
  // and J is the individual's own -- nothing for the compiler to keep in ten more registers across the loop)
  double JJ[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) JJ[a][b] = LIGHT ? J[a][b] : J[a][b] + J[a][b];
  double tmp[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    tmp[k] = F.g[FreqProducts::at(k, 0)] * JJ[geno1(0, k)][geno2(0, k)];
#pragma unroll
    for (int h = 1; h < 4; ++h) tmp[k] += F.g[FreqProducts::at(k, h)] * JJ[geno1(h, k)][geno2(h, k)];
  }
  // The four IEEE divisions by ONE denominator.  What the compiler makes of `tmp / sum` on gfx950 is, per quotient:
  // v_div_scale x 2, v_rcp_f64, two Newton steps on the reciprocal (4 FMAs), q0 = n * r, e = fma(-d, q0, n), v_div_fmas
  // (fma(e, r, q0), then the scaling undone) and v_div_fixup (special operands) -- 11 instructions: the four of them were 44 of
  // the 135 instructions of an individual's step.  Where v_div_scale would hand both operands
  // through unchanged and v_div_fixup would hand the quotient through -- denominator and numerators ordinary numbers well inside
  // the exponent range, div_operand_plain above -- the refined reciprocal depends on the denominator only and is formed ONCE:
  // the same instructions on the same operands as the compiler's sequence, hence the same bits (rcp + 4 FMAs, then mul + 2 FMAs
  // per quotient; 33 instructions with the test).  A wavefront with one lane outside that range (likelihoods of ~1e-200, a
  // frequency that has reached a denormal) takes the compiler's divisions for that individual, all lanes:
  // tests/test_gpu_replay_lkl.py holds both ways to the host's quotients bit for bit.  Lane kernel 338 -> 315 ms for 31.2e6
  // pairs (profiles/r05/late/shared_rcp); since then it is bound by the bytes it re-reads, not by its instructions.
  bool odd;
  if (LIGHT) {
    odd = !(sum > 0.0);
  } else {
    int plain = div_operand_plain<false>(sum);
#pragma unroll
    for (int k = 0; k < 4; ++k) plain &= div_operand_plain<true>(tmp[k]);
    odd = plain == 0;
  }
  if (__builtin_amdgcn_ballot_w64(counts && odd) == 0) {
    double r = __builtin_amdgcn_rcp(sum);
    r = fma(r, fma(-sum, r, 1.0), r);
    r = fma(r, fma(-sum, r, 1.0), r);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double q0 = tmp[k] * r;
      out[k] = fma(fma(-sum, q0, tmp[k]), r, q0);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = tmp[k] / sum;
  }
}

// (s1, s2) of plan records met in increasing order: a full search for the first, steps afterwards
struct Cursor {
  uint64_t item = ~0ull;  // index into the plan's items
  Item it;
  uint32_t pop = 0;
  __device__ __forceinline__ bool seek(const ReplayLklArgs &A, uint64_t rec, uint32_t *s1, uint32_t *s2) {
    if (item == ~0ull || rec < it.first_record) {
      uint32_t lo = 0, hi = A.n_sites;  // largest row with row_off[row] <= rec
      while (lo + 1 < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (A.row_off[mid] <= rec) lo = mid; else hi = mid;
      }
      uint64_t il = A.item_off[lo], ih = A.item_off[lo + 1];
      if (il >= ih) return false;
      while (il + 1 < ih) {
        const uint64_t mid = il + (ih - il) / 2;
        if (A.items[mid].first_record <= rec) il = mid; else ih = mid;
      }
      item = il;
      it = A.items[item];
      pop = (uint32_t)__popcll(it.mask);
    }
    while (rec >= it.first_record + pop) {  // (items follow one another in record order; an item without a pair is stepped over)
      if (++item >= A.n_items) return false;
      it = A.items[item];
      pop = (uint32_t)__popcll(it.mask);
    }
    uint64_t k = rec - it.first_record, mk = it.mask;
    while (k--) mk &= mk - 1;
    *s1 = it.s1;
    *s2 = it.s2_begin + (uint32_t)(__ffsll((unsigned long long)mk) - 1);
    return true;
  }
};

// sum of two values over the team's lanes (every lane gets both); `red`: 2 * WAVES doubles of LDS
template <int WAVES>
__device__ __forceinline__ void team_sum2(double &u, double &v, double *red, int wave, int lane) {
  double w = 0;
  wave_sum3(u, v, w);
  if (WAVES > 1) {
    __syncthreads();
    if (lane == 0) {
      red[2 * wave] = u;
      red[2 * wave + 1] = v;
    }
    __syncthreads();
    u = 0;
    v = 0;
    for (int q = 0; q < WAVES; ++q) {
      u += red[2 * q];
      v += red[2 * q + 1];
    }
  }
}

template <int WAVES, int kSlots>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void replay_lkl_kernel(ReplayLklArgs A) {
  constexpr int kRow = WAVES * kSlots * 64 + 2;  // doubles per row of quotients (+2: the four chain lanes read different banks)
  __shared__ __attribute__((aligned(16))) double quo[4 * kRow];  // [haplotype][individual]
  __shared__ double fnew[4];
  __shared__ double red[2 * WAVES];
  __shared__ uint32_t sh_u32[2 + WAVES];  // [0] claimed chunk, [1] converged, [2 + w] individuals with data in wavefront w
  const int lane = threadIdx.x & 63;
  const int wave = WAVES == 1 ? 0 : (int)(threadIdx.x >> 6);
  const bool ign = A.ignore_miss != 0;
  const uint64_t n_words = (A.n_records + 31) / 32;
  constexpr int kTeamSlots = WAVES * kSlots;  // slot t of the team holds individuals 64 t .. 64 t + 63
  double a[kSlots][3], b[kSlots][3];
  uint32_t row_site = 0xffffffffu;
  if (A.after_lanes && A.flags[6] == 0) return;  // (the lane-per-pair kernel took everything: nothing to walk the bitmap for)
  for (;;) {
    if (threadIdx.x == 0) sh_u32[0] = atomicAdd(A.work, 1u);
    __syncthreads();
    const uint64_t w0 = (uint64_t)sh_u32[0] * A.chunk_words;
    __syncthreads();
    if (w0 >= n_words) break;
    Cursor cur;
    for (uint64_t w = w0; w < w0 + A.chunk_words && w < n_words; ++w) {
      uint32_t bits = A.bits[w] & ~A.host_bits[w];
      bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)bits);
      for (; bits; bits &= bits - 1) {
        const uint64_t slot = w * 32 + (uint64_t)(__ffs((int)bits) - 1);
        if (slot >= A.n_records) break;
        uint32_t s1 = 0, s2 = 0;
        if (!cur.seek(A, A.rec_base + slot, &s1, &s2)) continue;
        s1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s1);
        s2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s2);

        // ---- the two sites: this lane's individuals, in registers for the whole pair ----
        const double *pa = A.xplanes + (uint64_t)s1 * A.site_stride, *pb = A.xplanes + (uint64_t)s2 * A.site_stride;
        uint32_t valid = 0;
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
          const uint32_t i = (uint32_t)((wave * kSlots + j) * 64 + lane);
          const bool in = i < A.n_ind;
          if (s1 != row_site) {
#pragma unroll
            for (int g = 0; g < 3; ++g) a[j][g] = in ? pa[(uint64_t)g * A.np + i] : 0.0;
          }
#pragma unroll
          for (int g = 0; g < 3; ++g) b[j][g] = in ? pb[(uint64_t)g * A.np + i] : 0.0;
          if (in && !(ign && (no_data(a[j]) || no_data(b[j])))) valid |= 1u << j;  // gen_func.cpp:1089
        }
        row_site = s1;
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < kSlots; ++j) x += (uint32_t)__popcll(__ballot((valid >> j) & 1u));
        if (WAVES > 1) {
          if (lane == 0) sh_u32[2 + wave] = x;
          __syncthreads();
          x = 0;
          for (int v = 0; v < WAVES; ++v) x += sh_u32[2 + v];
        }

        // ---- r2_ExpG where the pair kernel's moment is ill conditioned (same test, same per-site values as write_pair) ----
        const double c1 = fabs(A.rsx[s1]), c2 = fabs(A.rsx[s2]);
        if (c1 != __builtin_inf() && c2 != __builtin_inf() && (double)A.n_ind * c1 * c2 > kPearsonCond) {
          // pearson_r (ngsLD.cpp:365-367) over ALL individuals of expected_geno = p1 + 2 p2 (ngsLD.cpp:113), two passes
          double ex[kSlots], ey[kSlots], sx = 0, sy = 0;
#pragma unroll
          for (int j = 0; j < kSlots; ++j) {
            ex[j] = a[j][1] + 2 * a[j][2];
            ey[j] = b[j][1] + 2 * b[j][2];
            sx += ex[j];  // (individuals beyond n_ind are zeros)
            sy += ey[j];
          }
          team_sum2<WAVES>(sx, sy, red, wave, lane);
          const double mx = sx / (double)A.n_ind, my = sy / (double)A.n_ind;
          double sxx = 0, syy = 0, sxy = 0;
#pragma unroll
          for (int j = 0; j < kSlots; ++j) {
            const bool in = (uint32_t)((wave * kSlots + j) * 64 + lane) < A.n_ind;
            const double dx = in ? ex[j] - mx : 0.0, dy = in ? ey[j] - my : 0.0;
            sxx += dx * dx;
            syy += dy * dy;
            sxy += dx * dy;
          }
          team_sum2<WAVES>(sxx, syy, red, wave, lane);
          double zero = 0;
          team_sum2<WAVES>(sxy, zero, red, wave, lane);
          const double r = sxy / (__dsqrt_rn(sxx) * __dsqrt_rn(syy)), r2 = r * r;
          const double n = (double)A.n_ind;
          const double ms1 = fabs(mx) * __dsqrt_rn(n / sxx), ms2 = fabs(my) * __dsqrt_rn(n / syy);
          // what GSL's recurrence may differ from this by (long double: 2^-64 per operation, the running means' rounding
          // carried into every centred term), plus this evaluation's own few ulp
          const double bound = 0x1p-50 + 2 * fabs(r) * n * 0x1p-63 * (2 + ms1 + ms2);
          const double t6 = r2 * 1e6;
          const bool on_edge = A.flag_text != 0 && fabs((t6 - floor(t6)) - 0.5) < bound * 1e6;
          if (!(ms1 <= kMeanOverStd) || !(ms2 <= kMeanOverStd) || !(sxx > 0) || !(syy > 0) || on_edge) {
            if (threadIdx.x == 0) {  // the host's: marked where the host looks for its pairs
              atomicOr(&A.host_bits[slot >> 5], 1u << (slot & 31u));
              const uint32_t kh = atomicAdd(&A.flags[1], 1u);
              if (kh < kFlagHostCap) reinterpret_cast<uint64_t *>(A.flags + kFlagListAt + 2u * A.flag_cap)[kh] = slot;
            }
            continue;
          }
          if (threadIdx.x == 0) A.out_std[slot].r2_ExpG = r2;
        }

        // ---- haplo_freq (gen_func.cpp:1027-1059) ----
        const double m1 = A.xmaf[s1], m2 = A.xmaf[s2];
        double f[4];
        uint32_t iter = 0;
        if (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1) {  // gen_func.cpp:1030-1031
          if (threadIdx.x == 0) atomicExch(A.status, (int)NGSLD_ERR_MAF_RANGE);
          f[0] = f[1] = f[2] = f[3] = __builtin_nan("");
          x = 0;
        } else {
          f[0] = (1 - m1) * (1 - m2);
          f[1] = (1 - m1) * m2;
          f[2] = m1 * (1 - m2);
          f[3] = m1 * m2;
          for (iter = 0; iter < (uint32_t)kMaxIter; ++iter) {
            // every individual's four quotients, parked in individual order (an individual left out adds +0: ff + 0 == ff)
            FreqProducts F;
            F.set(f);
#pragma unroll
            for (int j = 0; j < kSlots; ++j) {
              // (the nine products p[g1] * q[g2] do not change from iteration to iteration and the compiler would keep them --
              // 18 more registers per slot, spilled from six slots on; made opaque here they are formed again, 9 multiplies)
#pragma unroll
              for (int g = 0; g < 3; ++g) asm volatile("" : "+v"(b[j][g]));
              double o[4];
              const bool on = (valid >> j) & 1u;
              quotients(F, a[j], b[j], o, on);
              const int i = (wave * kSlots + j) * 64 + lane;
#pragma unroll
              for (int k = 0; k < 4; ++k) quo[k * kRow + i] = on ? o[k] : 0.0;
              __builtin_amdgcn_sched_barrier(0);  // (one slot at a time: interleaved, the slots' temporaries spill)
            }
            __syncthreads();
            // the chain, gen_func.cpp:1103: ff[k] += tmp / sum, individual by individual
            if (wave == 0) {
              double ff = 0;
              if (lane < 4) {
                const double *r = quo + lane * kRow;
#pragma unroll 4
                for (int i = 0; i < kTeamSlots * 64; i += 2) {
                  const double2 v = *reinterpret_cast<const double2 *>(r + i);
                  ff += v.x;
                  ff += v.y;
                }
              }
              const double twox = (double)(2 * (uint64_t)x);  // gen_func.cpp:1109: 2 * x is an integer product
              double g[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) g[k] = read_lane(ff, k) / twox;
              g[0] /= g[0] + g[1] + g[2] + g[3];  // gen_func.cpp:1112-1113: sequential -- f[0] is already divided when f[1] is
              g[1] /= g[0] + g[1] + g[2] + g[3];
              g[2] /= g[0] + g[1] + g[2] + g[3];
              g[3] /= g[0] + g[1] + g[2] + g[3];
              double eps = 0;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const double d = fabs(g[k] - f[k]);
                if (d > eps) eps = d;  // (a NaN never raises eps: an all-NaN step ends the loop here)
              }
              if (WAVES > 1 && lane == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) fnew[k] = g[k];
                sh_u32[1] = eps < kEps ? 1u : 0u;
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) f[k] = g[k];
              if (WAVES == 1) {
                if (eps < kEps) break;
                continue;
              }
            }
            if (WAVES > 1) {
              __syncthreads();
#pragma unroll
              for (int k = 0; k < 4; ++k) f[k] = fnew[k];
              const bool done = sh_u32[1] != 0;
              __syncthreads();  // (fnew / the quotients are written again in the next iteration)
              if (done) break;
            }
          }
        }
        // ---- ngsLD.cpp:296-306 ----
        if (threadIdx.x == 0) {
          const double hm0 = 1 - (f[0] + f[1]);
          const double hm1 = 1 - (f[0] + f[2]);
          const double D = f[0] * f[3] - f[1] * f[2];
          const double Dp = D / (D < 0 ? -ref_min(hm0 * hm1, (1 - hm0) * (1 - hm1)) : ref_min(hm0 * (1 - hm1), (1 - hm0) * hm1));
          const double rr = D / __dsqrt_rn(hm0 * hm1 * (1 - hm0) * (1 - hm1));
          ngsld_rec_std *o = A.out_std + slot;  // (r2_ExpG stays the pair kernel's: three stores, no read -- the records may sit in
          o->D = ref_nan(D);                     // pinned host memory, a read would cross the host link)
          o->Dp = ref_nan(Dp);
          o->r2 = ref_nan(rr * rr);
          if (A.out_ext != nullptr) {
            ngsld_rec_ext r;
            r.hap[0] = ref_nan(f[0]); r.hap[1] = ref_nan(f[1]); r.hap[2] = ref_nan(f[2]); r.hap[3] = ref_nan(f[3]);
            r.n_ind_data = x;
            r.n_iter = iter;
            A.out_ext[slot] = r;
          }
          atomicAdd(A.done, 1u);
        }
        if (WAVES > 1) __syncthreads();
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The lane-per-pair form.  The wavefront-per-pair kernel above spends a third of its issue slots on four lanes adding
// 4 x n_ind numbers one by one, and with several wavefronts per pair the others wait for that chain (n_ind 2,000: 6e6
// replayed pairs/s against 5.4e7 at 500).  Here a LANE owns a pair and walks the individuals in the reference's own order --
// no chain, no idle lanes: every pair has the same number of individuals, so the 64 lanes of a wavefront reach the end of
// their EM iteration together, and a lane whose pair has converged takes its next pair there.  ~101 executed VALU instructions
// per individual and iteration for 64 pairs at once (round 6; 144 when the kernel was written).  The lanes of a wavefront hold neighbouring pairs (claimed in list order),
// and the store is read individual-major -- xT[i][site][3] -- so that for one individual their 24-byte triples are neighbours
// in memory (a monomorphic row's candidates: consecutive sites, 1.5 KB contiguous).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_store_kernel(const double *__restrict__ xplanes, uint64_t site_stride, uint32_t np,
                                                              uint32_t n_ind, uint64_t pitch_sites, double *__restrict__ xT, uint64_t site_begin,
                                                              uint64_t n_sites, uint32_t *__restrict__ xdepth, const uint32_t *__restrict__ xperm) {
  // (sites [site_begin, n_sites) of a matrix of pitch_sites sites: the builder moves the store chunk by chunk)
  __shared__ double tile[3][64][65];  // [genotype][site][individual]
  const uint64_t s0 = site_begin + (uint64_t)blockIdx.x * 64;
  const uint32_t i0 = blockIdx.y * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int ls = w; ls < 64; ls += 4) {
    const uint64_t s = s0 + ls;
    const uint32_t i = i0 + lane;
    uint32_t depth = 0;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const double v = (s < n_sites && i < n_ind) ? xplanes[s * site_stride + (uint64_t)g * np + i] : 0.0;
      tile[g][ls][lane] = v;
      const uint32_t d = plain_depth(v);
      depth = d > depth ? d : depth;
    }
    // (xdepth, preset to zero: the site's largest plain_depth, see quotients<LIGHT>)
    if (xdepth != nullptr) {
      for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)depth, off);
        depth = o > depth ? o : depth;
      }
      if (lane == 0 && depth != 0 && s < n_sites) atomicMax(&xdepth[s], depth);
    }
  }
  __syncthreads();
  // individual li of the tile: its 64 sites x 3 values are 192 consecutive doubles of xT
  for (int li = w; li < 64; li += 4) {
    const uint32_t i = i0 + li;
    if (i >= n_ind) continue;
    for (int t = lane; t < 192; t += 64) {
      const int ls = t / 3, g = t % 3;
      if (s0 + ls < n_sites) xT[((uint64_t)i * pitch_sites + (xperm != nullptr ? xperm[s0 + ls] : s0 + ls)) * 3 + g] = tile[g][ls][li];
    }
  }
}

__global__ __launch_bounds__(256) void replay_expand_kernel(ReplayLklArgs A, ReplayEntry *list, uint64_t list_cap) {
  const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t n_words = (A.n_records + 31) / 32;
  if (w >= n_words) return;
  if (A.only_if_overflow && A.flags[0] <= A.flag_cap) return;
  uint32_t bits = A.bits[w] & ~A.host_bits[w];
  if (w == n_words - 1 && (A.n_records & 31)) bits &= (1u << (A.n_records & 31)) - 1u;
  if (!bits) return;
  // which of them the lane-per-pair kernel takes: all but the pairs whose Pearson moment is ill conditioned (write_pair's
  // test on the same per-site values: those need the two passes of the wavefront-per-pair kernel)
  Cursor cur;
  uint32_t take = 0, s1s[32], s2s[32];
  for (uint32_t m = bits; m; m &= m - 1) {
    const int b = __ffs((int)m) - 1;
    uint32_t s1 = 0, s2 = 0;
    if (!cur.seek(A, A.rec_base + w * 32 + (uint64_t)b, &s1, &s2)) continue;
    const double c1 = fabs(A.rsx[s1]), c2 = fabs(A.rsx[s2]);
    if (c1 != __builtin_inf() && c2 != __builtin_inf() && (double)A.n_ind * c1 * c2 > kPearsonCond) continue;
    take |= 1u << b;
    s1s[b] = s1;
    s2s[b] = s2;
  }
  if (bits & ~take) atomicAdd(&A.flags[6], (uint32_t)__popc(bits & ~take));  // (left in the bitmap for the wavefront-per-pair kernel)
  if (!take) return;
  const uint32_t n = (uint32_t)__popc(take);
  const uint32_t at = atomicAdd(&A.flags[4], n);
  if ((uint64_t)at + n > list_cap) {  // (no room: these stay in the bitmap; the counter is put back so that the lanes see only what was written)
    atomicSub(&A.flags[4], n);
    atomicAdd(&A.flags[6], n);
    return;
  }
  uint32_t k = 0;
  for (uint32_t m = take; m; m &= m - 1, ++k) {
    const int b = __ffs((int)m) - 1;
    ReplayEntry e;
    e.slot = w * 32 + (uint64_t)b;
    e.s1 = s1s[b];
    e.s2 = s2s[b];
    list[at + k] = e;
  }
  A.bits[w] &= ~take;  // (this thread owns the word)
}

// Sort keys of the listed pairs, from (the pair's rarer site, its other site).  A wavefront's lanes take NEIGHBOURS of the sorted
// list; in list order half the pairs of an un-called matrix (ordinary row, monomorphic candidate) had a cache line to themselves
// per lane and individual.  Sorted by (rarer site, other site) the 64 lanes read ONE triple of the site they share and 64
// (nearly) consecutive others per individual -- and, re-reading them in every EM step from a copy no cache holds, the kernel ran
// at what HBM delivers (1.6 TB a launch).  So the key is TILED: [rarer site / 2^tile_s][other / 2^tile_o][rarer %
// 2^tile_s][other % 2^tile_o] -- the same bits in another order.  Rare sites that are neighbours have the same partners (their
// windows overlap), so 64 neighbours of the sorted list are a block of a few rare sites x 8 of their partners: a dozen or two
// distinct triples per individual instead of 65.  (tile_s = tile_o = 0: the plain order.)
__global__ void replay_keys_kernel(ReplayLklArgs A, const ReplayEntry *list, uint64_t list_cap, uint64_t *keys, uint32_t *vals, int site_bits,
                                   int tile_s, int tile_o) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= list_cap) return;
  vals[i] = (uint32_t)i;
  if (i >= A.flags[4]) {
    keys[i] = ~0ull >> (64 - 2 * site_bits);  // (behind every real key)
    return;
  }
  const ReplayEntry e = list[i];
  const double m1 = A.xmaf[e.s1], m2 = A.xmaf[e.s2];
  const double r1 = m1 <= 0.5 ? m1 : 1 - m1, r2 = m2 <= 0.5 ? m2 : 1 - m2;  // (NaN: compares false, the row's site is the shared one)
  const bool by2 = r2 < r1;
  // (positions in the individual-major copy, where the rare sites stand together: ReplayLklArgs::xperm)
  const uint32_t p1 = A.xperm != nullptr ? A.xperm[e.s1] : e.s1, p2 = A.xperm != nullptr ? A.xperm[e.s2] : e.s2;
  const uint64_t shared = by2 ? p2 : p1, other = by2 ? p1 : p2;
  const uint64_t s_hi = shared >> tile_s, s_lo = shared & ((1ull << tile_s) - 1), o_hi = other >> tile_o, o_lo = other & ((1ull << tile_o) - 1);
  keys[i] = (((s_hi << (site_bits - tile_o)) | o_hi) << (tile_s + tile_o)) | (s_lo << tile_o) | o_lo;
}

constexpr uint32_t kLaneChunk = 256;  // sorted entries a wavefront claims at a time (fewer where the launch has few per wavefront: see the kernel)

// (CAPPED: short launches of large cohorts, ReplayLklArgs::lane_iter_cap -- a template so that the long launches' instruction
// stream is the one without the hand-back: with the test inside one kernel that stream came out 30 % slower, 447 against 343 ms
// for configs[2]'s 31e6 pairs)
// Four wavefronts to a SIMD: 128 registers, which the kernel meets with two 16-byte spills per EM STEP (none inside the loop over
// the individuals) once the doubled frequency products are gone (quotients<LIGHT>).  Same box, 20 % monomorphic sites: 136
// registers at three wavefronts 627.9 ms a pass, 128 at four 613.7 (profiles/r06/lane/g_ab.txt).
constexpr int kLaneWavesPerSimd = 4;
template <bool CAPPED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void replay_lane_kernel(ReplayLklArgs A, const ReplayEntry *list,
                                                                                                      const uint32_t *order,
                                                                                                      const double *xT) {
  const bool ign = A.ignore_miss != 0;
  const uint64_t row = A.xt_sites * 3;  // doubles from one individual to the next
  const uint32_t total = A.flags[4];
  // a claim: 256 neighbours of the sorted list (four pairs a lane: the lanes that refill stay inside the tile) where the launch has
  // dozens of claims per wavefront; a short launch (the driver's 10,000-site matrix: 3.2e6 pairs, three claims of 256 per
  // wavefront) ends with wavefronts idle beside others' last 256 -- there, down to 64
  uint32_t chunk = (total / (gridDim.x * 8u)) & ~63u;
  chunk = chunk < 64u ? 64u : (chunk > kLaneChunk ? kLaneChunk : chunk);
  bool have = false, dry = false;       // this lane holds a pair / the list has run out
  uint32_t sites_depth = kNotPlain;     // plain_depth of the pair's two sites, added up (ReplayLklArgs::xdepth)
  ReplayEntry e{};
  const double *pa = xT, *pb = xT;
  double f[4] = {0, 0, 0, 0};
  uint32_t iter = 0, pos = 0, end = 0;  // (pos, end: the wavefront's chunk of the sorted list, wave-uniform)
  // (pairs this lane settled / handed back: added to the launch's counters ONCE, at the end -- as `atomicAdd(A.done, 1)` per pair
  // the compiler turned it into one atomic per wavefront in the plain kernel but into one per LANE, all on one address, in the
  // capped one: 447 against 343 ms for configs[2]'s 31e6 pairs)
  uint32_t n_done = 0, n_back = 0;
  for (;;) {
    // lanes without a pair take the next entries of the wavefront's chunk, in lane order; a new chunk with one atomic
    for (;;) {
      const uint64_t need = __ballot(!have && !dry);
      if (!need) break;
      if (pos == end) {
        uint32_t c0 = 0;
        if ((uint32_t)__lane_id() == (uint32_t)(__ffsll((unsigned long long)need) - 1)) c0 = atomicAdd(&A.flags[5], chunk);
        c0 = (uint32_t)__builtin_amdgcn_readlane((int)c0, __ffsll((unsigned long long)need) - 1);
        if (c0 >= total) {
          if (!have) dry = true;
          break;
        }
        pos = c0;
        end = c0 + chunk < total ? c0 + chunk : total;
      }
      const uint32_t rank = (uint32_t)__popcll(need & ((1ull << __lane_id()) - 1ull));
      const bool take = !have && !dry && pos + rank < end;
      const uint32_t n_need = (uint32_t)__popcll(need);
      if (take) {
        e = list[order[pos + rank]];
        const double m1 = A.xmaf[e.s1], m2 = A.xmaf[e.s2];
        if (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1) {  // gen_func.cpp:1030-1031: error() in the reference
          atomicExch(A.status, (int)NGSLD_ERR_MAF_RANGE);
          f[0] = f[1] = f[2] = f[3] = __builtin_nan("");
        } else {
          f[0] = (1 - m1) * (1 - m2);  // gen_func.cpp:1034-1037
          f[1] = (1 - m1) * m2;
          f[2] = m1 * (1 - m2);
          f[3] = m1 * m2;
        }
        pa = xT + (uint64_t)(A.xperm != nullptr ? A.xperm[e.s1] : e.s1) * 3;
        pb = xT + (uint64_t)(A.xperm != nullptr ? A.xperm[e.s2] : e.s2) * 3;
        sites_depth = A.xdepth != nullptr ? A.xdepth[e.s1] + A.xdepth[e.s2] : kNotPlain;
        iter = 0;
        have = true;
      }
      pos = pos + n_need < end ? pos + n_need : end;
    }
    if (!__any(have)) break;
    // ---- one EM step (gen_func.cpp:1076-1119) for every lane that holds a pair ----
    double ff[4] = {0, 0, 0, 0};
    uint32_t x = 0;
    // the step's individuals with the one-instruction operand test (quotients<LIGHT>) where EVERY lane's pair qualifies in this
    // step, the general test otherwise
    uint32_t f_depth = plain_depth(f[0]);
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const uint32_t d = plain_depth(f[k]);
      f_depth = d > f_depth ? d : f_depth;
    }
    const bool light = __ballot(have && 2u * f_depth + sites_depth > 596u) == 0;
    auto individuals = [&](auto light_tag) {
      constexpr bool kLight = decltype(light_tag)::value;
      FreqProducts F;
      F.set(f);
      // The individuals in order, two register sets used in turn: the next individual's triples are on their way while this one
      // is worked on, and nothing is copied from a staging set into place (-1.1 % of a pass, profiles/r06/lane/ab2.txt)
      // (D register sets: an individual's triples are fetched D - 1 steps before they are used.  Three and four measured +-0 at the
      // same three wavefronts per SIMD, five and six -9 % at two: profiles/r06/lane/sets_ab.txt)
      constexpr int D = 2;
      double a[D][3], b[D][3];
      auto fetch = [&](double (&pa_)[3], double (&pb_)[3], uint32_t i) {
        const double *qa = pa + (uint64_t)i * row, *qb = pb + (uint64_t)i * row;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          pa_[g] = qa[g];
          pb_[g] = qb[g];
        }
      };
      auto step = [&](const double (&pa_)[3], const double (&pb_)[3]) {
        if (ign) {
          if (no_data(pa_) || no_data(pb_)) return;  // gen_func.cpp:1089
          ++x;
        }
        double o[4];
        quotients<kLight>(F, pa_, pb_, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) ff[k] += o[k];  // gen_func.cpp:1103, in the reference's order
      };
#pragma unroll
      for (int j = 0; j < D; ++j)
        if ((uint32_t)j < A.n_ind) fetch(a[j], b[j], (uint32_t)j);
      uint32_t i = 0;
      for (; i + D <= A.n_ind; i += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          step(a[j], b[j]);
          __builtin_amdgcn_sched_barrier(0);
          if (i + D + j < A.n_ind) fetch(a[j], b[j], i + D + j);
        }
      }
#pragma unroll
      for (int j = 0; j < D - 1; ++j)
        if (i + j < A.n_ind) step(a[j], b[j]);
    };
    if (have) {
      if (light)
        individuals(std::true_type());
      else
        individuals(std::false_type());
      if (!ign) x = A.n_ind;
      const double twox = light ? (double)x : (double)(2 * (uint64_t)x);  // gen_func.cpp:1109 (light: the sums are half the reference's)
      double g4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) g4[k] = ff[k] / twox;
      g4[0] /= g4[0] + g4[1] + g4[2] + g4[3];  // gen_func.cpp:1112-1113: sequential
      g4[1] /= g4[0] + g4[1] + g4[2] + g4[3];
      g4[2] /= g4[0] + g4[1] + g4[2] + g4[3];
      g4[3] /= g4[0] + g4[1] + g4[2] + g4[3];
      double eps = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double d = fabs(g4[k] - f[k]);
        if (d > eps) eps = d;  // (a NaN never raises eps)
        f[k] = g4[k];
      }
      bool finished = eps < kEps;             // gen_func.cpp:1054: break with n_iter = iter
      if (!finished && ++iter == (uint32_t)kMaxIter) finished = true;  // ... or the loop runs out: n_iter = ITER_MAX
      // (CAPPED) a long pair in a short launch (a text batch of a large cohort: a lane takes ~0.4 ms an iteration over 2,000
      // individuals and the launch would last as long as its slowest lane) is handed to the wavefront-per-pair kernel behind
      // this one -- its bit set again --, which starts it over
      const bool hand_back = CAPPED && !finished && iter >= A.lane_iter_cap;
      if (finished || hand_back) {
        have = false;
        if (hand_back) {
          atomicOr(&A.bits[e.slot >> 5], 1u << (e.slot & 31u));
          ++n_back;
        } else {
          const double hm0 = 1 - (f[0] + f[1]);  // ngsLD.cpp:296-306
          const double hm1 = 1 - (f[0] + f[2]);
          const double D = f[0] * f[3] - f[1] * f[2];
          const double Dp = D / (D < 0 ? -ref_min(hm0 * hm1, (1 - hm0) * (1 - hm1)) : ref_min(hm0 * (1 - hm1), (1 - hm0) * hm1));
          const double rr = D / __dsqrt_rn(hm0 * hm1 * (1 - hm0) * (1 - hm1));
          ngsld_rec_std *o = A.out_std + e.slot;  // (r2_ExpG stays the pair kernel's: three stores, no read)
          o->D = ref_nan(D);
          o->Dp = ref_nan(Dp);
          o->r2 = ref_nan(rr * rr);
          if (A.out_ext != nullptr) {
            ngsld_rec_ext r;
            r.hap[0] = ref_nan(f[0]); r.hap[1] = ref_nan(f[1]); r.hap[2] = ref_nan(f[2]); r.hap[3] = ref_nan(f[3]);
            r.n_ind_data = x;
            r.n_iter = iter;
            A.out_ext[e.slot] = r;
          }
          ++n_done;
        }
      }
    }
  }
  if (n_done) atomicAdd(A.done, n_done);
  if (CAPPED && n_back) atomicAdd(&A.flags[6], n_back);
}

}  // namespace

hipError_t launch_transpose_store(const double *xplanes, uint64_t site_stride, uint32_t np, uint32_t n_ind, uint64_t n_sites,
                                  double *xT, uint32_t *xdepth, const uint32_t *xperm, hipStream_t stream, uint64_t site_begin,
                                  uint64_t site_end) {
  if (site_end > n_sites) site_end = n_sites;
  if (site_begin >= site_end || n_ind == 0) return hipSuccess;
  const dim3 grid((unsigned)((site_end - site_begin + 63) / 64), (unsigned)((n_ind + 63) / 64));
  hipLaunchKernelGGL(transpose_store_kernel, grid, dim3(256), 0, stream, xplanes, site_stride, np, n_ind, n_sites, xT, site_begin, site_end, xdepth, xperm);
  return hipGetLastError();
}

hipError_t launch_replay_expand(const ReplayLklArgs &a, ReplayEntry *list, uint64_t list_cap, hipStream_t stream) {
  if (a.bits == nullptr || a.n_records == 0 || list == nullptr) return hipSuccess;
  const uint64_t n_words = (a.n_records + 31) / 32;
  hipLaunchKernelGGL(replay_expand_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, stream, a, list, list_cap);
  return hipGetLastError();
}

static int replay_site_bits(uint32_t n_sites) {
  int b = 1;
  while (b < 32 && (1ull << b) < (uint64_t)n_sites) ++b;
  return b;
}

size_t replay_sort_temp_bytes(uint64_t list_cap, uint32_t n_sites) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr,
                                           (uint32_t *)nullptr, (int)list_cap, 0, 2 * replay_site_bits(n_sites));
  return bytes;
}

hipError_t launch_replay_sort(const ReplayLklArgs &a, const ReplayEntry *list, uint64_t list_cap, uint64_t *keys_a, uint64_t *keys_b,
                              uint32_t *vals_a, uint32_t *vals_b, void *temp, size_t temp_bytes, hipStream_t stream) {
  if (list_cap == 0) return hipSuccess;
  if (list_cap > 0x7fffffffull) return hipErrorInvalidValue;
  const int bits = replay_site_bits(a.n_sites);
  int tile_s = 5, tile_o = 3;  // 32 consecutive sites x 8 partners (other tilings within 1 %: profiles/r05/late/tile, profiles/r06/lane/perm_ab.txt)
  tile_s = std::min(tile_s, bits);
  tile_o = std::min(tile_o, bits);
  hipLaunchKernelGGL(replay_keys_kernel, dim3((unsigned)((list_cap + 255) / 256)), dim3(256), 0, stream, a, list, list_cap, keys_a, vals_a, bits,
                     tile_s, tile_o);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_a, keys_b, vals_a, vals_b, (int)list_cap, 0, 2 * bits, stream);
}

hipError_t launch_replay_lanes(const ReplayLklArgs &a, const ReplayEntry *list, const uint32_t *order, const double *xT, int n_cus,
                               int waves_per_simd, hipStream_t stream) {
  if (list == nullptr || xT == nullptr || a.n_records == 0) return hipSuccess;
  // a persistent grid: as many wavefronts as a SIMD holds of this kernel (kLaneWavesPerSimd), never more lanes than the launch has records
  uint64_t waves = (uint64_t)n_cus * 4 * (uint64_t)(waves_per_simd < 1 ? 1 : (waves_per_simd > kLaneWavesPerSimd ? kLaneWavesPerSimd : waves_per_simd));
  const uint64_t most = (a.n_records + 63) / 64;
  if (waves > most) waves = most;
  if (a.lane_iter_cap != 0)
    hipLaunchKernelGGL(replay_lane_kernel<true>, dim3((unsigned)waves), dim3(64), 0, stream, a, list, order, xT);
  else
    hipLaunchKernelGGL(replay_lane_kernel<false>, dim3((unsigned)waves), dim3(64), 0, stream, a, list, order, xT);
  return hipGetLastError();
}

// Cohorts beyond the wavefront-per-pair kernel's 4,096 individuals: what the lanes left in the bitmap -- pairs whose Pearson
// moment is ill conditioned, pairs beyond the list -- is handed to the host (its own bitmap, its own list), a thread per word.
__global__ __launch_bounds__(256) void replay_leftover_kernel(ReplayLklArgs A) {
  const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t n_words = (A.n_records + 31) / 32;
  if (w >= n_words) return;
  if (A.only_if_overflow && A.flags[0] <= A.flag_cap) return;  // (called genotypes: nothing was expanded, nothing is left over)
  uint32_t bits = A.bits[w] & ~A.host_bits[w];
  if (w == n_words - 1 && (A.n_records & 31)) bits &= (1u << (A.n_records & 31)) - 1u;
  if (!bits) return;
  A.host_bits[w] |= bits;  // (this thread owns the word)
  const uint32_t n = (uint32_t)__popc(bits);
  uint32_t kh = atomicAdd(&A.flags[1], n);
  for (uint32_t m = bits; m; m &= m - 1, ++kh)
    if (kh < kFlagHostCap) reinterpret_cast<uint64_t *>(A.flags + kFlagListAt + 2u * A.flag_cap)[kh] = w * 32 + (uint64_t)(__ffs((int)m) - 1);
}

hipError_t launch_replay_leftover(const ReplayLklArgs &a, hipStream_t stream) {
  if (a.bits == nullptr || a.n_records == 0) return hipSuccess;
  const uint64_t n_words = (a.n_records + 31) / 32;
  hipLaunchKernelGGL(replay_leftover_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

uint32_t replay_lkl_waves(uint32_t n_ind) {
  const uint32_t per = 64u * kMaxSlots;
  for (uint32_t w = 1; w <= 8; w *= 2)
    if (n_ind <= per * w) return w;
  return 0;  // (beyond 4,096 individuals: the host's replay)
}

hipError_t launch_replay_lkl(const ReplayLklArgs &a_in, int n_cus, hipStream_t stream) {
  ReplayLklArgs a = a_in;
  if (a.bits == nullptr || a.n_records == 0) return hipSuccess;
  const uint32_t waves = replay_lkl_waves(a.n_ind);
  if (waves == 0) return hipErrorInvalidValue;
  // a persistent grid: as many teams as the device holds at two wavefronts per SIMD (the kernel's registers allow no more),
  // never more than there are chunks to claim
  const uint64_t n_words = (a.n_records + 31) / 32;
  uint64_t teams = (uint64_t)n_cus * 8 / waves;
  if (waves == 8) teams = (uint64_t)n_cus;  // (its quotient rows take 131 KB of LDS: one team per CU)
  // A claim is 128 records where the launch is large (a binary search of the plan per claim, the row's vector kept from pair
  // to pair) -- but a text batch is 2^19 records, 2 such claims per team, and a monomorphic row flags ALL its pairs: one
  // team then sat on 256 pairs (10 ms) while the others had gone.  Every team should find sixteen claims or more.
  a.chunk_words = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(kChunkWords, n_words / (teams * 16)));
  const uint64_t chunks = (n_words + a.chunk_words - 1) / a.chunk_words;
  if (teams > chunks) teams = chunks;
  if (teams < 1) teams = 1;
  const dim3 grid((unsigned)teams);
  const uint32_t slots = (a.n_ind + 64u * waves - 1) / (64u * waves);  // per wavefront
#define NGSLD_REPLAY_LKL(W, S) hipLaunchKernelGGL((replay_lkl_kernel<W, S>), grid, dim3(64 * W), 0, stream, a)
  if (waves == 1) {
    switch (slots) {
      case 1: NGSLD_REPLAY_LKL(1, 1); break;
      case 2: NGSLD_REPLAY_LKL(1, 2); break;
      case 3: NGSLD_REPLAY_LKL(1, 3); break;
      case 4: NGSLD_REPLAY_LKL(1, 4); break;
      case 5: NGSLD_REPLAY_LKL(1, 5); break;
      case 6: NGSLD_REPLAY_LKL(1, 6); break;
      case 7: NGSLD_REPLAY_LKL(1, 7); break;
      default: NGSLD_REPLAY_LKL(1, 8); break;
    }
  } else if (waves == 2) {
    if (slots <= 6) NGSLD_REPLAY_LKL(2, 6); else NGSLD_REPLAY_LKL(2, 8);
  } else if (waves == 4) {
    if (slots <= 6) NGSLD_REPLAY_LKL(4, 6); else NGSLD_REPLAY_LKL(4, 8);
  } else {
    if (slots <= 6) NGSLD_REPLAY_LKL(8, 6); else NGSLD_REPLAY_LKL(8, 8);
  }
#undef NGSLD_REPLAY_LKL
  return hipGetLastError();
}

}  // namespace ngsld
