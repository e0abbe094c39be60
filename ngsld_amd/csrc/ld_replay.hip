// ld_replay.hip -- exact-order replay ON THE DEVICE for matrices of called genotypes (gfx950).
//
// The pair kernels flag the pairs whose outcome the reference's own rounding decides (ld_device.h, write_pair) and the
// engine re-evaluates those in the reference's operation order -- on the host (replay.cpp), because the order is sequential
// over the individuals and the inputs must be the reference's bits.  For likelihood matrices that is a few dozen pairs per
// 10^8.  For CALLED genotypes it is 26,000: with 2n haplotypes eps lands on EPSILON exactly, and the host's 0.2-0.4 s per
// pass left the product at 3.6e8 pairs/s for a pass whose kernel runs 1.7e9.  But called genotypes are the one
// input whose values are the SAME bits on the device as on the host -- every triple is (1,0,0), (0,1,0), (0,0,1) or "no
// data" -- so the sequential evaluation can run here, one LANE per flagged pair:
//   * est_maf (gen_func.cpp:974-1009) over the individuals in order, from the site's bit sets;
//   * pair_freq_iter (gen_func.cpp:1076-1119): an individual's sum and its four tmp / sum depend on its genotype combination
//     only, so the sixteen combinations are evaluated once per iteration -- in the reference's loop order, one rounding per
//     operation, IEEE division -- and then ADDED individual by individual in the reference's order: the same bits as the
//     reference's loop, ~25 instructions per individual instead of ~170 flops; the sequential renormalisation; the
//     convergence test of haplo_freq (gen_func.cpp:1041-1056);
//   * D, D', r2 (ngsLD.cpp:296-306).
// r2_ExpG is left as the pair kernel wrote it (GSL's long double recurrence has no device twin): pairs flagged for a reason
// that concerns it carry kFlagHostOnly and stay with the host.  "No data" individuals are taken with the HOST's values of
// the triple --call_geno leaves (log(1/3) three times through the host's exp: handed in as constants); a matrix whose
// missing triples may be anything else (miss_ok == 0) keeps the pairs of sites with missing individuals on the host.
#include "ld_common.h"
#include "ld_replay.h"

namespace ngsld {
namespace {

constexpr double kEps = 1e-5;  // EPSILON, gen_func.hpp:16
constexpr int kMaxIter = 100;  // ITER_MAX, gen_func.hpp:18

__device__ __forceinline__ double ref_abs(double x) { return x >= 0 ? x : -x; }           // gen_func.hpp:21-23: macros
__device__ __forceinline__ double ref_min(double a, double b) { return a <= b ? a : b; }

// genotype at site 1 / site 2 of the haplotype pair (h, k): bit 1 = allele at site 1, bit 0 = allele at site 2
__device__ __forceinline__ constexpr int geno1(int h, int k) { return ((h >> 1) & 1) + ((k >> 1) & 1); }
__device__ __forceinline__ constexpr int geno2(int h, int k) { return (h & 1) + (k & 1); }

// genotype class of individual i at a site: 0, 1, 2 = called genotype, 3 = no data
__device__ __forceinline__ int site_class(uint64_t m1, uint64_t m2, uint64_t m3, int b) {
  return (int)((m1 >> b) & 1ull) + 2 * (int)((m2 >> b) & 1ull) + 3 * (int)((m3 >> b) & 1ull);
}

// est_maf (gen_func.cpp:974-1009, indF == NULL) from a site's bit sets: the individuals in order, two passes as the
// reference's do-while takes them (num / den are not reset between passes)
__device__ double site_maf(const uint64_t *m, uint32_t W, uint32_t n_ind, bool ignore_miss, double u_pp, bool *has_missing) {
  double num = 0, den = 0, freq = 0.01, prev;
  int iters = 0;
  do {
    prev = freq;
    for (uint32_t w = 0; w < W; ++w) {
      const uint64_t m0 = m[w], m1 = m[W + w], m2 = m[2 * W + w], m3 = m[3 * W + w];
      const int nb = (int)(n_ind - 64u * w < 64u ? n_ind - 64u * w : 64u);
      for (int b = 0; b < nb; ++b) {
        double pp0, pp1, pp2;
        if ((m3 >> b) & 1ull) {
          *has_missing = true;
          if (ignore_miss) continue;
          pp0 = pp1 = pp2 = u_pp;
        } else {
          pp0 = (double)((m0 >> b) & 1ull); pp1 = (double)((m1 >> b) & 1ull); pp2 = (double)((m2 >> b) & 1ull);
        }
        const double F = 0;
        num += pp1 + pp2 * (2 - F);
        den += 2 * pp1 + (pp0 + pp2) * (2 - F);
      }
    }
    freq = num / den;
  } while (ref_abs(prev - freq) > kEps && iters++ < 100);
  return freq;
}

// (s1, s2) of the plan's record `rec` (engine_replay.hip: locate_record); false: no such record
__device__ bool locate_hard(const ReplayHardArgs &A, uint64_t rec, uint32_t *ps1, uint32_t *ps2) {
  uint32_t lo = 0, hi = A.n_sites;  // largest row with row_off[row] <= rec
  while (lo + 1 < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (A.row_off[mid] <= rec) lo = mid; else hi = mid;
  }
  const uint32_t row = lo;
  uint64_t il = A.item_off[row], ih = A.item_off[row + 1];
  if (il >= ih) return false;
  while (il + 1 < ih) {
    const uint64_t mid = il + (ih - il) / 2;
    if (A.items[mid].first_record <= rec) il = mid; else ih = mid;
  }
  const Item it = A.items[il];
  uint64_t k = rec - it.first_record, mk = it.mask;
  if (k >= (uint64_t)__popcll(mk)) return false;
  while (k--) mk &= mk - 1;
  *ps1 = it.s1;
  *ps2 = it.s2_begin + (uint32_t)(__ffsll((unsigned long long)mk) - 1);
  return true;
}

// One flagged pair in the reference's operation order (see the header of this file); val: the calling lane's column of the
// workgroup's LDS table.  false: the pair is left to the host (a site with missing individuals whose triples may be anything).
__device__ bool replay_hard_pair(const ReplayHardArgs &A, double *val, int lane, uint64_t slot, uint32_t s1, uint32_t s2) {

  const uint32_t W = A.words;
  const uint64_t *ma = A.masks + (uint64_t)s1 * 4 * W, *mb = A.masks + (uint64_t)s2 * 4 * W;
  const bool ign = A.ignore_miss != 0;
  bool missing = false;
  const double m1 = site_maf(ma, W, A.n_ind, ign, A.u_pp, &missing);
  const double m2 = site_maf(mb, W, A.n_ind, ign, A.u_pp, &missing);
  if (missing && !A.miss_ok) return false;  // (the host has the caller's raw values for these)

  // ---- haplo_freq (gen_func.cpp:1027-1059) ----
  double f[4];
  uint64_t x = 0, iter = 0;
  if (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1) {  // gen_func.cpp:1030-1031
    atomicExch(A.status, (int)NGSLD_ERR_MAF_RANGE);
    f[0] = f[1] = f[2] = f[3] = __builtin_nan("");
  } else {
    f[0] = (1 - m1) * (1 - m2);
    f[1] = (1 - m1) * m2;
    f[2] = m1 * (1 - m2);
    f[3] = m1 * m2;
    const double u = A.u_lkl;
    for (iter = 0; iter < (uint64_t)kMaxIter; ++iter) {
      const double last[4] = {f[0], f[1], f[2], f[3]};
      // the sixteen genotype combinations: sum and tmp_k / sum exactly as pair_freq_iter forms them for one individual
#pragma unroll
      for (int c1 = 0; c1 < 4; ++c1) {
        const double p[3] = {c1 == 0 ? 1.0 : (c1 == 3 ? u : 0.0), c1 == 1 ? 1.0 : (c1 == 3 ? u : 0.0), c1 == 2 ? 1.0 : (c1 == 3 ? u : 0.0)};
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
          const double q[3] = {c2 == 0 ? 1.0 : (c2 == 3 ? u : 0.0), c2 == 1 ? 1.0 : (c2 == 3 ? u : 0.0), c2 == 2 ? 1.0 : (c2 == 3 ? u : 0.0)};
          double sum = 0;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int h = 0; h < 4; ++h) sum += f[kk] * f[h] * p[geno1(kk, h)] * q[geno2(kk, h)];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            double tmp = 0;
#pragma unroll
            for (int h = 0; h < 4; ++h)
              tmp += f[kk] * f[h] * (p[geno1(h, kk)] * q[geno2(h, kk)] + p[geno1(kk, h)] * q[geno2(kk, h)]);
            val[((c1 * 4 + c2) * 4 + kk) * 64 + lane] = tmp / sum;
          }
        }
      }
      // the individuals, in the reference's order: ff[k] += tmp_k / sum
      double ff0 = 0, ff1 = 0, ff2 = 0, ff3 = 0;
      x = 0;
      for (uint32_t w = 0; w < W; ++w) {
        const uint64_t a1 = ma[W + w], a2 = ma[2 * W + w], a3 = ma[3 * W + w];
        const uint64_t b1 = mb[W + w], b2 = mb[2 * W + w], b3 = mb[3 * W + w];
        const int nb = (int)(A.n_ind - 64u * w < 64u ? A.n_ind - 64u * w : 64u);
        for (int b = 0; b < nb; ++b) {
          const int c1 = site_class(a1, a2, a3, b), c2 = site_class(b1, b2, b3, b);
          if (ign && (c1 == 3 || c2 == 3)) continue;  // gen_func.cpp:1089
          ++x;
          const double *v = val + (c1 * 4 + c2) * 4 * 64 + lane;
          ff0 += v[0]; ff1 += v[64]; ff2 += v[128]; ff3 += v[192];
        }
      }
      f[0] = ff0 / (double)(2 * x); f[1] = ff1 / (double)(2 * x); f[2] = ff2 / (double)(2 * x); f[3] = ff3 / (double)(2 * x);
      f[0] /= f[0] + f[1] + f[2] + f[3];  // gen_func.cpp:1112-1113: sequential -- f[0] is already divided when f[1] is
      f[1] /= f[0] + f[1] + f[2] + f[3];
      f[2] /= f[0] + f[1] + f[2] + f[3];
      f[3] /= f[0] + f[1] + f[2] + f[3];
      double eps = 0;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double d = fabs(f[kk] - last[kk]);
        if (d > eps) eps = d;  // (a NaN never raises eps: an all-NaN step ends the loop here)
      }
      if (eps < kEps) break;
    }
  }
  // ---- ngsLD.cpp:296-306 ----
  const double hm0 = 1 - (f[0] + f[1]);
  const double hm1 = 1 - (f[0] + f[2]);
  const double D = f[0] * f[3] - f[1] * f[2];
  const double Dp = D / (D < 0 ? -ref_min(hm0 * hm1, (1 - hm0) * (1 - hm1)) : ref_min(hm0 * (1 - hm1), (1 - hm0) * hm1));
  const double rr = D / __dsqrt_rn(hm0 * hm1 * (1 - hm0) * (1 - hm1));
  ngsld_rec_std o = A.out_std[slot];  // (r2_ExpG stays the pair kernel's)
  o.D = ref_nan(D);
  o.Dp = ref_nan(Dp);
  o.r2 = ref_nan(rr * rr);
  A.out_std[slot] = o;
  if (A.out_ext != nullptr) {
    ngsld_rec_ext r;
    r.hap[0] = ref_nan(f[0]); r.hap[1] = ref_nan(f[1]); r.hap[2] = ref_nan(f[2]); r.hap[3] = ref_nan(f[3]);
    r.n_ind_data = (uint32_t)x;
    r.n_iter = (uint32_t)iter;
    A.out_ext[slot] = r;
  }
  return true;
}

__global__ __launch_bounds__(64) void replay_hard_kernel(ReplayHardArgs A) {
  __shared__ double val[64 * 64];  // [16 combinations x 4 haplotypes][lane]: this lane's tmp_k / sum per combination
  const int lane = threadIdx.x;
  if (A.list != nullptr && A.flags[0] > A.flag_cap) return;  // (more flagged pairs than the list holds: replay_hard_list_kernel takes all of them)
  const uint32_t count = A.flags[0] < A.flag_cap ? A.flags[0] : A.flag_cap;
  const uint32_t e = blockIdx.x * 64u + (uint32_t)lane;
  if (e >= count) return;
  uint64_t *list = reinterpret_cast<uint64_t *>(A.flags + kFlagListAt);
  const uint64_t entry = list[e];
  if (entry & (kFlagHostOnly | kFlagDone)) return;
  const uint64_t slot = entry & kFlagIndexMask;
  uint32_t s1 = 0, s2 = 0;
  if (!locate_hard(A, A.rec_base + slot, &s1, &s2)) return;
  if (replay_hard_pair(A, val, lane, slot, s1, s2)) list[e] = entry | kFlagDone;
}

// A launch that flagged MORE pairs than its list holds (a called-genotype matrix with monomorphic sites: every pair of such a
// site; rounds 2-4 left all of those to the host's threads -- bench --hard-calls --mono-frac 0.2: 2.1e8 pairs/s for a pass whose
// kernel runs 1.6e9): the launch's bitmap has been turned into a list of located pairs (launch_replay_expand, shared with the
// likelihood replay), a persistent grid works through it, one lane per pair.  What this kernel cannot settle (see
// replay_hard_pair) is added to the host-only bitmap and list; flags[7] tells the host that the rest is done.
__global__ __launch_bounds__(64) void replay_hard_list_kernel(ReplayHardArgs A) {
  __shared__ double val[64 * 64];
  const int lane = threadIdx.x;
  if (A.flags[0] <= A.flag_cap) return;
  if (blockIdx.x == 0 && lane == 0) A.flags[7] = 1u;
  const uint32_t total = A.flags[4];
  for (;;) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&A.flags[5], 64u);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (base >= total) return;
    const uint32_t idx = base + (uint32_t)lane;
    if (idx < total) {  // (no `continue` for the others: the claim above is the wavefront's, every lane must come back to it together)
      const ReplayEntry e = A.list[idx];
      if (replay_hard_pair(A, val, lane, e.slot, e.s1, e.s2)) {
        atomicAdd(&A.flags[2], 1u);
      } else {
        atomicOr(&A.host_bits[e.slot >> 5], 1u << (e.slot & 31u));
        const uint32_t kh = atomicAdd(&A.flags[1], 1u);
        if (kh < kFlagHostCap) reinterpret_cast<uint64_t *>(A.flags + kFlagListAt + 2u * A.flag_cap)[kh] = e.slot;
      }
    }
  }
}

}  // namespace

hipError_t launch_replay_hard(const ReplayHardArgs &a, uint64_t n_records, hipStream_t stream) {
  if (a.flags == nullptr || n_records == 0) return hipSuccess;
  const uint64_t most = n_records < (uint64_t)a.flag_cap ? n_records : (uint64_t)a.flag_cap;
  hipLaunchKernelGGL(replay_hard_kernel, dim3((unsigned)((most + 63) / 64)), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_replay_hard_list(const ReplayHardArgs &a, int n_cus, hipStream_t stream) {
  if (a.flags == nullptr || a.list == nullptr) return hipSuccess;
  hipLaunchKernelGGL(replay_hard_list_kernel, dim3((unsigned)(n_cus * 8)), dim3(64), 0, stream, a);
  return hipGetLastError();
}

}  // namespace ngsld
