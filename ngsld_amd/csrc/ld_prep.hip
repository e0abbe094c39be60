// ld_prep.hip -- per-site preprocessing, work-item construction and the primitive self test (gfx950).
//
// prep_sites_kernel does on the device what read_geno's binary branch (shared/read_data.cpp:28-47)
// and main() (ngsLD.cpp:103-114) do on the host in the reference, one workgroup per site:
//   raw [site][ind][3] doubles -> log (-inf -> -1e15) -> log-normalise -> NaN check -> est_maf ->
//   exp -> planes [site][geno][np] (+ mean and centred second moment of the expected genotypes, which
//   is everything pearson_r needs per site).
#include "ld_prep.h"
#include "taus.h"

namespace ngsld {

// shared/gen_func.cpp:135-151 logsum over a triple (macro max, left-to-right sum)
__device__ __forceinline__ double logsum3(double g0, double g1, double g2) {
  double M = g0;
  M = g1 >= M ? g1 : M;
  M = g2 >= M ? g2 : M;
  if (M == -__builtin_inf()) return -__builtin_inf();
  double sum = 0.0;
  sum += exp(g0 - M);
  sum += exp(g1 - M);
  sum += exp(g2 - M);
  return log(sum) + M;
}

__device__ __forceinline__ double conv_log(double g) {  // conv_space(log), gen_func.cpp:123-130
  g = log(g);
  return g == -__builtin_inf() ? -1e15 : g;
}

// gen_func.cpp:886-914 call_geno(geno, 3, log_scale = true, N_thresh, call_thresh, miss_data = 0) as called at
// ngsLD.cpp:97; array_max_pos / array_min_pos (gen_func.cpp:73-98) keep the FIRST extreme.
__device__ __forceinline__ void call_geno(double &g0, double &g1, double &g2, double N_thresh, double call_thresh) {
  int max_pos = 0;
  double mx = -__builtin_inf();
  if (g0 > mx) { max_pos = 0; mx = g0; }
  if (g1 > mx) { max_pos = 1; mx = g1; }
  if (g2 > mx) { max_pos = 2; mx = g2; }
  double mn = __builtin_inf();
  if (g0 < mn) mn = g0;
  if (g1 < mn) mn = g1;
  if (g2 < mn) mn = g2;
  const double vmax = max_pos == 0 ? g0 : (max_pos == 1 ? g1 : g2);
  double max_pp = exp(vmax);
  if (mn == vmax) max_pp = -1.0;  // missing data
  if (max_pp < N_thresh) g0 = g1 = g2 = log(1.0 / 3.0);
  if (max_pp >= call_thresh) {
    g0 = g1 = g2 = -1e15;
    if (max_pos == 0) g0 = 0.0;
    if (max_pos == 1) g1 = 0.0;
    if (max_pos == 2) g2 = 0.0;
  }
}

// fixed-order workgroup sum of up to 3 values (256 threads)
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double (*sh)[N]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = wave_sum1(v[k]);
  __syncthreads();
  if (lane == 0)
    for (int k = 0; k < N; ++k) sh[wave][k] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = ((sh[0][k] + sh[1][k]) + sh[2][k]) + sh[3][k];
}

// workgroup min and max of one value per thread (256 threads), through shuffles + LDS
__device__ __forceinline__ void block_minmax(double (&mn)[1], double (&mx)[1], double (*sh)[1]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) {
    const double a = __shfl_xor(mn[0], off), b = __shfl_xor(mx[0], off);
    mn[0] = a < mn[0] ? a : mn[0];
    mx[0] = b > mx[0] ? b : mx[0];
  }
  __syncthreads();
  if (lane == 0) sh[wave][0] = mn[0];
  __syncthreads();
  double m = sh[0][0];
  for (int w = 1; w < 4; ++w) m = sh[w][0] < m ? sh[w][0] : m;
  __syncthreads();
  if (lane == 0) sh[wave][0] = mx[0];
  __syncthreads();
  double M = sh[0][0];
  for (int w = 1; w < 4; ++w) M = sh[w][0] > M ? sh[w][0] : M;
  mn[0] = m;
  mx[0] = M;
}

// One individual's raw triple -> normal-space normalised likelihoods a0..a2 and its two est_maf terms.
//
// The reference's sequence (read_data.cpp:37-45, gen_func.cpp:974-1009, ngsLD.cpp:110): log, log-normalise (logsum:
// three exp and a log), NaN check, [call_geno], est_maf's own logsum and three exp, three exp back to normal space -- 12
// exp and 5 log per individual, ~750 f64 instructions: the prep kernel was bound by them, not by memory (1.5 TB/s).
// Mathematically the result is a_k = raw_k / (raw_0 + raw_1 + raw_2) and est_maf's posterior is a itself; the chain of
// logs only adds rounding noise (~|log raw| * 1e-16).  FAST path: exactly that quotient, for triples that are finite,
// non-negative, not all zero and in the normal range -- every case in which the chain has no special behaviour.  Everything
// else (zeros everywhere, negative / NaN / inf values, log-scale or text input, --call_geno) takes the chain as written.
// Agreement with the reference's compiled est_maf stays far inside the 1e-12 bar (tests: ref_maf of every golden fixture).
struct PrepFlags {
  bool log_scale, ignore_miss, text_semantics, call_geno, fast_ok;
  double N_thresh, call_thresh;
};

__device__ __forceinline__ bool close_in_log(double x, double y) {  // |log x - log y| < EPSILON (miss_data on log values)
  constexpr double kExpEps = 1.00001000005000016667;                // exp(1e-5)
  return x == y || (x < y * kExpEps && y < x * kExpEps);
}

__device__ __forceinline__ void prep_individual(const PrepFlags &F, double g0, double g1, double g2, double &a0, double &a1,
                                                double &a2, double &num, double &den, bool &nan_seen) {
  double mx = g0 > g1 ? g0 : g1;
  mx = g2 > mx ? g2 : mx;
  if (F.fast_ok && g0 >= 0.0 && g1 >= 0.0 && g2 >= 0.0 && mx >= 0x1p-960 && mx <= 0x1p+960) {
    const double r = 1.0 / ((g0 + g1) + g2);
    a0 = g0 * r;
    a1 = g1 * r;
    a2 = g2 * r;
    if (!(F.ignore_miss && close_in_log(a0, a1) && close_in_log(a1, a2))) {  // miss_data on LOG values, gen_func.cpp:985
      num += a1 + a2 * 2.0;
      den += 2.0 * a1 + (a0 + a2) * 2.0;
    }
    return;
  }
  if (!F.log_scale) {
    if (F.text_semantics) {  // read_data.cpp:86: plain log(), -inf stays
      g0 = log(g0);
      g1 = log(g1);
      g2 = log(g2);
    } else {  // read_data.cpp:37-38
      g0 = conv_log(g0);
      g1 = conv_log(g1);
      g2 = conv_log(g2);
    }
  }
  const double norm = logsum3(g0, g1, g2);  // post_prob, read_data.cpp:40
  g0 -= norm;
  g1 -= norm;
  g2 -= norm;
  if (!F.text_semantics && (g0 != g0 || g1 != g1 || g2 != g2)) nan_seen = true;  // read_data.cpp:42-45
  if (F.call_geno) call_geno(g0, g1, g2, F.N_thresh, F.call_thresh);             // ngsLD.cpp:92-98
  // est_maf (gen_func.cpp:974-1009, indF == NULL): closed form of its two identical passes
  if (!(F.ignore_miss && miss_data(g0, g1, g2))) {  // miss_data on LOG values, :985
    const double n2 = logsum3(g0, g1, g2);
    const double p0 = exp(g0 - n2), p1 = exp(g1 - n2), p2 = exp(g2 - n2);
    num += p1 + p2 * 2.0;
    den += 2.0 * p1 + (p0 + p2) * 2.0;
  }
  a0 = exp(g0);  // ngsLD.cpp:110
  a1 = exp(g1);
  a2 = exp(g2);
}

__device__ __forceinline__ PrepFlags prep_flags(const PrepArgs &A) {
  PrepFlags F;
  F.log_scale = A.log_scale != 0;
  F.ignore_miss = A.ignore_miss != 0;
  F.text_semantics = A.text_semantics != 0;
  F.call_geno = A.call_geno != 0;
  F.fast_ok = !F.log_scale && !F.text_semantics && !F.call_geno && A.exact_chain == 0;
  F.N_thresh = A.N_thresh;
  F.call_thresh = A.call_thresh;
  return F;
}

// A site's three per-site scalars from its sums.  A site whose expected genotypes are all the same value must come out with
// variance exactly 0 (gsl_stats_correlation's running mean has delta == 0 there and returns 0/0): min == max makes the
// mean that value itself rather than a rounded sum / n.
// (rsx = +inf there: NaN on every path.  Nearly constant sites are dealt with per PAIR: ld_device.h, kPearsonCond.)
__device__ __forceinline__ double site_rsx(double sq) { return 1.0 / sqrt(sq); }  // sq = sum (e - mean)^2
// The pair kernels relabel the alleles of a site whose frequency is above 1/2 (ld_device.h, Relabel) and take the Pearson
// moment from the relabelled expected genotypes, 2 - e -- which is what p1 + 2 p0 is only if every triple sums to 1.  The
// reference's own chain leaves triples that do not: all-zero natural-scale input comes out as 0.3247 three times (the
// log-sum at -1e15 rounds log 3 to 1.125), and ngsld_set_geno_lkl takes whatever the caller normalised.  A site that holds
// such a triple AND would be relabelled says so in the SIGN of its rsx: write_pair takes the magnitude and flags every pair
// of the site for the exact-order replay (r2_ExpG would be off in the second decimal otherwise).
// A site whose expected genotypes are all EQUAL here need not be constant in the reference: the quotient path gives 1/3 three
// times for (1, 1, 1) and for (0.001, 0.001, 0.001) alike, the reference's log / exp chain 1.0 and 0.9999999999999996 -- and
// gsl_stats_correlation then correlates that rounding noise (two individuals: r2_ExpG = 1 where this side says NaN; found by the
// round-5 fuzz over matrices that are not SNP-called).  Individuals with IDENTICAL raw triples are identical there too; a
// site that is constant here over raw triples that differ says so in the sign of its (infinite) rsx: its pairs go to the host's
// replay, which has the caller's values.
__device__ __forceinline__ uint64_t triple_key(double g0, double g1, double g2) {
  const uint64_t a = (uint64_t)__double_as_longlong(g0), b = (uint64_t)__double_as_longlong(g1), c = (uint64_t)__double_as_longlong(g2);
  return a ^ ((b << 21) | (b >> 43)) ^ ((c << 42) | (c >> 22));
}
constexpr double kSumTol = 0x1p-40;
__device__ __forceinline__ bool odd_triple(double a0, double a1, double a2) { return !(fabs((a0 + a1 + a2) - 1.0) <= kSumTol); }
__device__ __forceinline__ double signed_rsx(double rsx, bool odd, double maf) { return (odd && maf > 0.5 - 1e-9) ? -rsx : rsx; }

// n_ind <= 2048: one WAVEFRONT per site, a lane holds the expected genotypes of its <= MAXJ individuals in registers -- no
// barrier, no second pass over the planes (the workgroup-per-site form re-read them twice and met ten times per site).
template <int MAXJ>
__global__ __launch_bounds__(256) void prep_sites_wave_kernel(PrepArgs A) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const PrepFlags F = prep_flags(A);
  for (uint64_t site = (uint64_t)blockIdx.x * 4 + wave; site < A.n_sites; site += (uint64_t)gridDim.x * 4) {
    const double *__restrict__ raw = A.raw + site * (uint64_t)A.n_ind * 3;
    double *__restrict__ pl = A.planes + (A.site0 + site) * A.site_stride;
    double e[MAXJ];
    double num = 0.0, den = 0.0, esum = 0.0;
    double mn = __builtin_inf(), mx = -__builtin_inf();
    uint64_t kmin = ~0ull, kmax = 0ull;
    bool nan_seen = false, odd = false;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const uint32_t i = (uint32_t)lane + 64u * (uint32_t)j;
      e[j] = 0.0;
      if (i < A.np) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        if (i < A.n_ind) {
          const double g0 = raw[3 * (uint64_t)i], g1 = raw[3 * (uint64_t)i + 1], g2 = raw[3 * (uint64_t)i + 2];
          const uint64_t key = triple_key(g0, g1, g2);
          kmin = key < kmin ? key : kmin;
          kmax = key > kmax ? key : kmax;
          if (!A.normalised_input) {
            prep_individual(F, g0, g1, g2, a0, a1, a2, num, den, nan_seen);
          } else {
            a0 = g0;
            a1 = g1;
            a2 = g2;
          }
          odd = odd || odd_triple(a0, a1, a2);
          const double ev = fma(2.0, a2, a1);  // expected genotype, ngsLD.cpp:113
          e[j] = ev;
          esum += ev;
          mn = ev < mn ? ev : mn;
          mx = ev > mx ? ev : mx;
        }
        pl[i] = a0;
        pl[A.np + i] = a1;
        pl[2 * (uint64_t)A.np + i] = a2;
      }
    }
    wave_sum3(num, den, esum);
    for (int off = 32; off > 0; off >>= 1) {
      const double x = __shfl_xor(mn, off), y = __shfl_xor(mx, off);
      mn = x < mn ? x : mn;
      mx = y > mx ? y : mx;
      const uint64_t p = (uint64_t)__shfl_xor((long long)kmin, off), q = (uint64_t)__shfl_xor((long long)kmax, off);
      kmin = p < kmin ? p : kmin;
      kmax = q > kmax ? q : kmax;
    }
    const double mean = mn == mx ? mn : esum / (double)A.n_ind;
    double sq = 0.0;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const uint32_t i = (uint32_t)lane + 64u * (uint32_t)j;
      const double d = i < A.n_ind ? e[j] - mean : 0.0;
      sq = fma(d, d, sq);
    }
    sq = wave_sum1(sq);
    const bool any_odd = __ballot(odd) != 0;
    if (lane == 0) {
      const double maf = A.normalised_input ? A.maf_in[site] : num / den;
      A.maf[A.site0 + site] = maf;
      A.mean_e[A.site0 + site] = mean;
      double rsx = signed_rsx(site_rsx(sq), any_odd, maf);
      if (mn == mx && kmin != kmax && !A.normalised_input) rsx = -__builtin_inf();  // (constant here over raw triples that differ: see triple_key)
      A.rsx[A.site0 + site] = rsx;
    }
    if (nan_seen) atomicExch(A.status, (int)NGSLD_ERR_NAN);
  }
}

// Any n_ind: one workgroup per site, fixed-order workgroup sums.
__global__ __launch_bounds__(256) void prep_sites_kernel(PrepArgs A) {
  __shared__ double sh3[4][3];
  __shared__ double sh1[4][1];
  const PrepFlags F = prep_flags(A);
  for (uint64_t site = blockIdx.x; site < A.n_sites; site += gridDim.x) {
    const double *__restrict__ raw = A.raw + site * (uint64_t)A.n_ind * 3;
    double *__restrict__ pl = A.planes + (A.site0 + site) * A.site_stride;
    double acc[3] = {0.0, 0.0, 0.0};  // num, den (est_maf), sum of expected genotypes
    double mn[1] = {__builtin_inf()}, mx[1] = {-__builtin_inf()};
    uint64_t key0 = 0;
    bool have_key = false, differs = false;
    bool nan_seen = false, odd = false;
    for (uint32_t i = threadIdx.x; i < A.np; i += 256) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      if (i < A.n_ind) {
        const double g0 = raw[3 * (uint64_t)i], g1 = raw[3 * (uint64_t)i + 1], g2 = raw[3 * (uint64_t)i + 2];
        const uint64_t key = triple_key(g0, g1, g2);
        if (!have_key) {
          key0 = key;
          have_key = true;
        } else if (key != key0) {
          differs = true;
        }
        if (!A.normalised_input) {
          prep_individual(F, g0, g1, g2, a0, a1, a2, acc[0], acc[1], nan_seen);
        } else {
          a0 = g0;
          a1 = g1;
          a2 = g2;
        }
        odd = odd || odd_triple(a0, a1, a2);
        const double ev = fma(2.0, a2, a1);  // expected genotype, ngsLD.cpp:113
        acc[2] += ev;
        mn[0] = ev < mn[0] ? ev : mn[0];
        mx[0] = ev > mx[0] ? ev : mx[0];
      }
      pl[i] = a0;
      pl[A.np + i] = a1;
      pl[2 * (uint64_t)A.np + i] = a2;
    }
    block_sum<3>(acc, sh3);
    block_minmax(mn, mx, sh1);
    const double mean = mn[0] == mx[0] ? mn[0] : acc[2] / (double)A.n_ind;
    double sq[1] = {0.0};
    for (uint32_t i = threadIdx.x; i < A.n_ind; i += 256) {  // (its own stores: visible to the thread that made them)
      const double d = fma(2.0, pl[2 * (uint64_t)A.np + i], pl[A.np + i]) - mean;
      sq[0] = fma(d, d, sq[0]);
    }
    block_sum<1>(sq, sh1);
    const bool any_odd = __syncthreads_or(odd ? 1 : 0) != 0;
    // (raw triples all identical: every thread's own are, and every thread's first equals individual 0's)
    const uint64_t first_key = triple_key(raw[0], raw[1], raw[2]);
    const bool raw_differs = __syncthreads_or((differs || (have_key && key0 != first_key)) ? 1 : 0) != 0;
    if (threadIdx.x == 0) {
      const double maf = A.normalised_input ? A.maf_in[site] : acc[0] / acc[1];
      A.maf[A.site0 + site] = maf;
      A.mean_e[A.site0 + site] = mean;
      double rsx = signed_rsx(site_rsx(sq[0]), any_odd, maf);
      if (mn[0] == mx[0] && raw_differs && !A.normalised_input) rsx = -__builtin_inf();  // (see triple_key)
      A.rsx[A.site0 + site] = rsx;
    }
    if (nan_seen) atomicExch(A.status, (int)NGSLD_ERR_NAN);
  }
}

// see PrepArgs::odd_missing
__global__ __launch_bounds__(256) void missing_check_kernel(const double *__restrict__ raw, uint64_t n_triples, double canon, int *odd) {
  const uint64_t T = (uint64_t)gridDim.x * blockDim.x;
  const unsigned long long cb = (unsigned long long)__double_as_longlong(canon);
  bool bad = false;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_triples; t += T) {
    const double g0 = raw[3 * t], g1 = raw[3 * t + 1], g2 = raw[3 * t + 2];
    const double mn = fmin(g0, fmin(g1, g2)), mx = fmax(g0, fmax(g1, g2));
    if (mx - mn < 1e-9) {  // (planes come out equal only where the raw values are equal to rounding; a NaN compares false and is the NaN check's)
      const bool canonical = (unsigned long long)__double_as_longlong(g0) == cb && (unsigned long long)__double_as_longlong(g1) == cb &&
                             (unsigned long long)__double_as_longlong(g2) == cb;
      bad |= !canonical;
    }
  }
  if (bad) *odd = 1;
}

hipError_t launch_prep(const PrepArgs &a, hipStream_t stream) {
  if (a.n_sites == 0) return hipSuccess;
  if (a.odd_missing != nullptr) {
    const uint64_t n_triples = a.n_sites * (uint64_t)a.n_ind;
    const uint64_t wgs = (n_triples + 255) / 256;
    hipLaunchKernelGGL(missing_check_kernel, dim3((unsigned)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, stream, a.raw, n_triples, a.missing_canon,
                       a.odd_missing);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (a.np <= 2048) {
    const uint64_t wgs = (a.n_sites + 3) / 4;
    const unsigned grid = (unsigned)(wgs < 65536 ? wgs : 65536);
    if (a.np <= 512)
      hipLaunchKernelGGL(prep_sites_wave_kernel<8>, dim3(grid), dim3(256), 0, stream, a);
    else if (a.np <= 1024)
      hipLaunchKernelGGL(prep_sites_wave_kernel<16>, dim3(grid), dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL(prep_sites_wave_kernel<32>, dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
  }
  const unsigned grid = (unsigned)(a.n_sites < 65536 ? a.n_sites : 65536);
  hipLaunchKernelGGL(prep_sites_kernel, dim3(grid), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// Which sites are DEGENERATE: sites whose every pair (nearly) the exact-order replay will settle anyway.  On a matrix that is
// not SNP-called (README.md:73) a site monomorphic in the sample sends the pair EM's minor margins towards zero until eps <
// EPSILON stops it, and D', r2 of such a pair are quotients by those margins: write_pair flags it once 2^-49 (1/q0 + 1/q1)
// max(|D'|, r2) > 2.5e-10, i.e. q < 7e-6 max(|D'|, r2) -- and the replay starts the pair over.  The margin of a site in the
// two-locus EM moves as the ONE-locus EM of that site does (exactly so without LD), so the predictor is that EM: from the
// site's est_maf frequency (where haplo_freq starts, gen_func.cpp:1034-1037), HWE weights, until its own step is below
// EPSILON (where the pair's loop would stop on this site's account, gen_func.cpp:1054); a site that ends below kSkipBelow is
// marked.  The pair kernels then leave the EM of such a site's pairs out (NaN frequencies: flagged) and the replay, which is
// exact for ANY pair, is their only evaluation: a wrong mark costs time, never a digit.  Measured on bench.py's generator, 500
// individuals (profiles/r06/skip/predictor.txt): with 20 % monomorphic sites the mark catches 98.9 % of the flagged pairs and
// 1.7 % of all pairs are marked without need; log-uniform spectrum 93.1 % / 0.6 %.
constexpr double kSkipBelow = 3e-6;
template <int MAXJ>  // individuals per lane kept in registers (0: the planes are read again in every step)
__global__ __launch_bounds__(256) void site_skip_kernel(const double *__restrict__ planes, uint64_t site_stride, uint32_t np, uint32_t n_ind,
                                                        int ignore_miss, const double *__restrict__ maf, uint64_t n_sites,
                                                        uint8_t *__restrict__ skip, uint32_t *count) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint64_t site = (uint64_t)blockIdx.x * 4 + wave; site < n_sites; site += (uint64_t)gridDim.x * 4) {
    const double *pl = planes + site * site_stride;
    const double m_in = maf[site];
    const bool flip = m_in > 0.5;  // (the minor allele's frequency is the one that goes to zero)
    double m = flip ? 1.0 - m_in : m_in;
    const double *p0 = pl + (flip ? 2ull * np : 0ull), *p1 = pl + np, *p2 = pl + (flip ? 0ull : 2ull * np);
    constexpr int kRegs = MAXJ > 0 ? MAXJ : 1;
    double a0[kRegs], a1[kRegs], a2[kRegs];
    uint32_t valid = 0, x = 0;
    const uint32_t n_slots = (n_ind + 63) / 64;
    auto is_valid = [&](uint32_t i, double g0, double g1, double g2) {
      return i < n_ind && !(ignore_miss && miss_data(g0, g1, g2));  // gen_func.cpp:1089 (normal space)
    };
    if (MAXJ > 0) {
#pragma unroll
      for (int j = 0; j < kRegs; ++j) {
        const uint32_t i = (uint32_t)lane + 64u * (uint32_t)j;
        const bool in = i < n_ind;
        a0[j] = in ? p0[i] : 0.0;
        a1[j] = in ? p1[i] : 0.0;
        a2[j] = in ? p2[i] : 0.0;
        if (is_valid(i, a0[j], a1[j], a2[j])) valid |= 1u << j;
      }
      x = 0;
#pragma unroll
      for (int j = 0; j < kRegs; ++j) x += (uint32_t)__popcll(__ballot((valid >> j) & 1u));
    }
    bool mark = false;
    for (int iter = 0; iter < kIterMax; ++iter) {
      const double w0 = (1 - m) * (1 - m), w1 = 2 * m * (1 - m), w2 = m * m;
      double acc = 0.0;
      uint32_t cnt = 0;
      auto one = [&](double g0, double g1, double g2) {
        const double s = fma(w2, g2, fma(w1, g1, w0 * g0));
        acc = fma(fma(2.0 * w2, g2, w1 * g1), rcp_refined(s), acc);
      };
      if (MAXJ > 0) {
#pragma unroll
        for (int j = 0; j < kRegs; ++j)
          if ((valid >> j) & 1u) one(a0[j], a1[j], a2[j]);
        cnt = x;
      } else {
        for (uint32_t j = 0; j < n_slots; ++j) {
          const uint32_t i = (uint32_t)lane + 64u * j;
          const double g0 = i < n_ind ? p0[i] : 0.0, g1 = i < n_ind ? p1[i] : 0.0, g2 = i < n_ind ? p2[i] : 0.0;
          const bool ok = is_valid(i, g0, g1, g2);
          if (ok) one(g0, g1, g2);
          cnt += (uint32_t)__popcll(__ballot(ok));
        }
      }
      const double mn = wave_sum1(acc) / (2.0 * (double)cnt);
      const bool stop = !(fabs(mn - m) >= kEpsilon);  // (a NaN stops too, and marks nothing)
      m = mn;
      if (stop) {
        mark = m < kSkipBelow;
        break;
      }
    }
    if (lane == 0) {
      skip[site] = mark ? 1 : 0;
      if (mark) atomicAdd(count, 1u);
    }
  }
}

hipError_t launch_site_skip(const double *planes, uint64_t site_stride, uint32_t np, uint32_t n_ind, int ignore_miss, const double *maf,
                            uint64_t n_sites, uint8_t *skip, uint32_t *count, hipStream_t stream) {
  if (n_sites == 0) return hipSuccess;
  const uint64_t wgs = (n_sites + 3) / 4;
  const dim3 grid((unsigned)(wgs < 65536 ? wgs : 65536)), block(256);
  if (np <= 512)
    hipLaunchKernelGGL(site_skip_kernel<8>, grid, block, 0, stream, planes, site_stride, np, n_ind, ignore_miss, maf, n_sites, skip, count);
  else if (np <= 1024)
    hipLaunchKernelGGL(site_skip_kernel<16>, grid, block, 0, stream, planes, site_stride, np, n_ind, ignore_miss, maf, n_sites, skip, count);
  else
    hipLaunchKernelGGL(site_skip_kernel<0>, grid, block, 0, stream, planes, site_stride, np, n_ind, ignore_miss, maf, n_sites, skip, count);
  return hipGetLastError();
}

// {maf, mean_e, rsx, degenerate} of every site side by side: the run kernel fetches a site's scalars with one 32-byte copy.
// (degenerate: 1.0 where site_skip_kernel marked the site, 0.0 elsewhere or without marks)
__global__ void pack_scalars_kernel(const double *maf, const double *mean_e, const double *rsx, const uint8_t *skip, double *sc4, uint64_t n) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  sc4[4 * s] = maf[s];
  sc4[4 * s + 1] = mean_e[s];
  sc4[4 * s + 2] = rsx[s];
  sc4[4 * s + 3] = (skip != nullptr && skip[s]) ? 1.0 : 0.0;
}

hipError_t launch_pack_scalars(const double *maf, const double *mean_e, const double *rsx, const uint8_t *skip, double *sc4, uint64_t n,
                               hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(pack_scalars_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, maf, mean_e, rsx, skip, sc4, n);
  return hipGetLastError();
}

__global__ void items_kernel(ItemArgs A) {
  const uint32_t s1 = blockIdx.x * blockDim.x + threadIdx.x;
  if (s1 >= A.n_sites) return;
  const uint32_t end = A.row_end[s1];
  const bool sampling = A.row_seed != nullptr;
  Taus rng;
  if (sampling) rng.set(A.row_seed[s1]);
  uint64_t kept = 0;
  uint64_t k = A.count_only ? 0 : A.item_off[s1];
  const uint64_t base = A.count_only ? 0 : A.row_off[s1];
  for (uint32_t b = s1 + 1; b < end; b += A.span) {
    const uint32_t cnt = end - b < A.span ? end - b : A.span;
    uint64_t mask = 0;
    for (uint32_t c = 0; c < cnt; ++c) {
      if (!A.keep[b + c]) continue;                                 // ngsLD.cpp:270-275
      if (sampling && rng.uniform() > A.rnd_sample) continue;      // ngsLD.cpp:277-282, one draw per surviving pair
      mask |= 1ull << c;
    }
    if (!A.count_only) {
      Item it;
      it.s1 = s1;
      it.s2_begin = b;
      it.count = cnt;
      it.reserved = 0;
      it.mask = mask;
      it.first_record = base + kept;
      A.items[k++] = it;
    }
    kept += (uint64_t)__popcll(mask);
  }
  if (A.count_only) A.row_count[s1] = kept;
}

hipError_t launch_items(const ItemArgs &a, hipStream_t stream) {
  if (a.n_sites == 0) return hipSuccess;
  hipLaunchKernelGGL(items_kernel, dim3((a.n_sites + 255) / 256), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// Self test: out[0..3] = wave_sum4 of in[k*64 + lane]; out[4 + lane] = rcp_refined(in[256 + lane]);
// out[68..70] = wave_sum3 of in[k*64 + lane], k = 0..2.
__global__ void selftest_kernel(const double *in, double *out) {
  const int lane = threadIdx.x;
  double t0 = in[lane], t1 = in[64 + lane], t2 = in[128 + lane], t3 = in[192 + lane];
  wave_sum4(t0, t1, t2, t3);
  if (lane == 0) {
    out[0] = t0;
    out[1] = t1;
    out[2] = t2;
    out[3] = t3;
  }
  out[4 + lane] = rcp_refined(in[256 + lane]);
  double u1 = in[lane], u2 = in[64 + lane], u3 = in[128 + lane];
  wave_sum3(u1, u2, u3);
  if (lane == 0) {
    out[68] = u1;
    out[69] = u2;
    out[70] = u3;
  }
}

hipError_t launch_selftest(const double *in, double *out, hipStream_t stream) {
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, stream, in, out);
  return hipGetLastError();
}

}  // namespace ngsld
