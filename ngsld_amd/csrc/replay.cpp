// replay.cpp -- see replay.h.  Built with -ffp-contract=off and no -march flag: x86-64 baseline arithmetic (no FMA),
// one rounding per operation, i.e. what the reference's `g++ -O3` (Makefile:9) produces for the same expressions.
// Every expression below keeps the association and the order of the reference line it cites.
#include "replay.h"

#include "../../include/ngsld_host.h"

#include <cmath>
#include <cstring>

namespace ngsld {
namespace {

constexpr double kInf = 1e15;     // INF, gen_func.hpp:15
constexpr double kEps = 1e-5;     // EPSILON, gen_func.hpp:16
constexpr int kMaxIter = 100;     // ITER_MAX, gen_func.hpp:18

// gen_func.hpp:21-23: abs / min / max are MACROS in the reference (NaN falls through to the second branch)
inline double ref_abs(double x) { return x >= 0 ? x : -x; }
inline double ref_min(double a, double b) { return a <= b ? a : b; }
inline double ref_max(double a, double b) { return a >= b ? a : b; }

// gen_func.cpp:135-151 logsum over a triple
double logsum3(const double *a) {
  double top = a[0];
  top = ref_max(a[1], top);
  top = ref_max(a[2], top);
  if (top == -INFINITY) return -INFINITY;
  double acc = 0;
  for (int g = 0; g < 3; ++g) acc += std::exp(a[g] - top);
  return std::log(acc) + top;
}

// gen_func.cpp:920-932 post_prob with prior == NULL, in place
void normalise_log(double *g) {
  const double norm = logsum3(g);
  for (int k = 0; k < 3; ++k) g[k] -= norm;
}

// gen_func.cpp:862-868
bool no_data(const double *g) { return ref_abs(g[0] - g[1]) < kEps && ref_abs(g[1] - g[2]) < kEps; }

// gen_func.cpp:886-914 as ngsLD.cpp:97 calls it: log_scale = true, miss_data = 0; array_max_pos / array_min_pos
// (gen_func.cpp:73-98) keep the first extreme
void harden(double *g, double n_thresh, double call_thresh) {
  int hi = 0, lo = 0;
  for (int k = 1; k < 3; ++k) {
    if (g[k] > g[hi]) hi = k;
    if (g[k] < g[lo]) lo = k;
  }
  double best = std::exp(g[hi]);
  if (g[lo] == g[hi]) best = -1;
  if (best < n_thresh)
    for (int k = 0; k < 3; ++k) g[k] = std::log((double)1 / 3);
  if (best >= call_thresh) {
    for (int k = 0; k < 3; ++k) g[k] = -kInf;
    g[hi] = std::log(1);
  }
}

// gen_func.cpp:1073-1119: one EM step.  Haplotype k: bit 1 = allele at site 1, bit 0 = allele at site 2.
inline int geno1(int h, int k) { return ((h >> 1) & 1) + ((k >> 1) & 1); }
inline int geno2(int h, int k) { return (h & 1) + (k & 1); }

uint64_t em_step(double f[4], const double *s1, const double *s2, uint64_t n, bool ignore_miss) {
  double ff[4] = {0, 0, 0, 0};
  uint64_t x = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const double *p = s1 + 3 * i, *q = s2 + 3 * i;
    if ((no_data(p) || no_data(q)) && ignore_miss) continue;
    ++x;
    double sum = 0;
    for (int k = 0; k < 4; ++k)
      for (int h = 0; h < 4; ++h) sum += f[k] * f[h] * p[geno1(k, h)] * q[geno2(k, h)];
    for (int k = 0; k < 4; ++k) {
      double tmp = 0;
      for (int h = 0; h < 4; ++h)
        tmp += f[k] * f[h] * (p[geno1(h, k)] * q[geno2(h, k)] + p[geno1(k, h)] * q[geno2(k, h)]);
      ff[k] += tmp / sum;
    }
  }
  for (int k = 0; k < 4; ++k) f[k] = ff[k] / (2 * x);             // 2 * x is an integer product, then converted
  for (int k = 0; k < 4; ++k) f[k] /= f[0] + f[1] + f[2] + f[3];  // sequential: f[0] is already divided when f[1] is
  return x;
}

// covariance recurrence of gsl_stats_correlation (GSL statistics/covar_source.c): long double accumulators, the two
// square roots taken in double
double correlation(const std::vector<double> &x, const std::vector<double> &y) {
  long double sxx = 0, syy = 0, sxy = 0, mx = x[0], my = y[0];
  for (size_t i = 1; i < x.size(); ++i) {
    const long double ratio = i / (i + 1.0);
    const long double dx = x[i] - mx, dy = y[i] - my;
    sxx += dx * dx * ratio;
    syy += dy * dy * ratio;
    sxy += dx * dy * ratio;
    mx += dx / (i + 1.0);
    my += dy / (i + 1.0);
  }
  const long double r = sxy / (std::sqrt((double)sxx) * std::sqrt((double)syy));
  return (double)r;
}

// One individual's triple through the chain above, as a pure function of its three raw values and the options: the
// log-space triple is only needed for est_maf's terms, which depend on nothing else either.  Matrices repeat triples -- a
// likelihood is a function of a handful of reads -- and the chain is 17 libm calls per triple: a small direct-mapped memo per
// thread returns the SAME bits for a triple seen before (the key is compared in full) at a twentieth of the cost.
struct TripleOut {
  double lkl[3];        // exp of the normalised logs, ngsLD.cpp:110
  double t_num, t_den;  // what est_maf adds to num / den for this individual (gen_func.cpp:992-993)
  bool skip_in_maf;     // miss_data on the log values (gen_func.cpp:985), used under --ignore_miss_data
};

void triple_chain(const double *raw, const ngsld_geno_opts &o, TripleOut *t) {
  double g[3];
  for (int k = 0; k < 3; ++k) {
    double v = raw[k];
    if (!o.log_scale) {
      v = std::log(v);                                         // read_data.cpp:37-38 / :86
      if (!o.text_semantics && v == -INFINITY) v = -kInf;      // conv_space, gen_func.cpp:127-128 (binary input only)
    }
    g[k] = v;
  }
  normalise_log(g);                                            // read_data.cpp:40 / :98
  if (o.call_geno) harden(g, o.N_thresh, o.call_thresh);       // ngsLD.cpp:92-98
  t->skip_in_maf = no_data(g);
  double pp[3] = {g[0], g[1], g[2]};                           // est_maf's posterior, gen_func.cpp:986-990
  normalise_log(pp);
  for (int k = 0; k < 3; ++k) pp[k] = std::exp(pp[k]);
  const double F = 0;
  t->t_num = pp[1] + pp[2] * (2 - F);
  t->t_den = 2 * pp[1] + (pp[0] + pp[2]) * (2 - F);
  for (int k = 0; k < 3; ++k) t->lkl[k] = std::exp(g[k]);      // ngsLD.cpp:110
}

struct TripleMemo {
  static constexpr size_t kEntries = 1u << 12;
  struct Entry {
    uint64_t key[3];
    bool used = false;
    TripleOut out;
  };
  std::vector<Entry> tab;
  ngsld_geno_opts opts{};
  const TripleOut &get(const double *raw, const ngsld_geno_opts &o) {
    if (tab.empty() || o.log_scale != opts.log_scale || o.text_semantics != opts.text_semantics || o.call_geno != opts.call_geno ||
        std::memcmp(&o.N_thresh, &opts.N_thresh, sizeof(double)) != 0 || std::memcmp(&o.call_thresh, &opts.call_thresh, sizeof(double)) != 0) {
      tab.assign(kEntries, Entry());
      opts = o;
    }
    uint64_t k[3];
    std::memcpy(k, raw, sizeof(k));
    uint64_t h = k[0] * 0x9E3779B97F4A7C15ull;
    h = (h ^ (h >> 29) ^ k[1]) * 0xBF58476D1CE4E5B9ull;
    h = (h ^ (h >> 32) ^ k[2]) * 0x94D049BB133111EBull;
    Entry &e = tab[(h >> 40) & (kEntries - 1)];
    if (!(e.used && e.key[0] == k[0] && e.key[1] == k[1] && e.key[2] == k[2])) {
      triple_chain(raw, o, &e.out);
      e.key[0] = k[0]; e.key[1] = k[1]; e.key[2] = k[2];
      e.used = true;
    }
    return e.out;
  }
};

}  // namespace

void replay_site_from_raw(const double *raw, uint64_t n_ind, const ngsld_geno_opts &o, ReplaySite *out) {
  thread_local TripleMemo memo;
  thread_local std::vector<TripleOut> vals;  // (copies: a later individual of the site may evict an entry)
  vals.resize(n_ind);
  for (uint64_t i = 0; i < n_ind; ++i) vals[i] = memo.get(raw + 3 * i, o);
  // est_maf, gen_func.cpp:974-1009 with indF == NULL (ngsLD.cpp:104-105): num / den live outside the do-while (they are not
  // reset between passes) and the posterior does not depend on freq, so the loop ends after its second pass; kept as the loop
  const bool ignore_miss = o.ignore_miss_data != 0;
  double num = 0, den = 0, freq = 0.01, prev;
  int iters = 0;
  do {
    prev = freq;
    for (uint64_t i = 0; i < n_ind; ++i) {
      if (vals[i].skip_in_maf && ignore_miss) continue;
      num += vals[i].t_num;
      den += vals[i].t_den;
    }
    freq = num / den;
  } while (ref_abs(prev - freq) > kEps && iters++ < 100);
  out->maf = freq;
  out->lkl.resize(3 * n_ind);
  out->e.resize(n_ind);
  for (uint64_t i = 0; i < n_ind; ++i) {
    double *p = out->lkl.data() + 3 * i;
    for (int k = 0; k < 3; ++k) p[k] = vals[i].lkl[k];
    out->e[i] = p[1] + 2 * p[2];                                   // ngsLD.cpp:113
  }
}

void replay_site_planes(const double *raw, uint64_t n_ind, const ngsld_geno_opts &o, uint64_t np, double *planes, double *maf) {
  thread_local ReplaySite site;  // (its vectors keep their capacity from site to site)
  replay_site_from_raw(raw, n_ind, o, &site);
  *maf = site.maf;
  for (int g = 0; g < 3; ++g) {
    double *pl = planes + (uint64_t)g * np;
    for (uint64_t i = 0; i < n_ind; ++i) pl[i] = site.lkl[3 * i + g];
    for (uint64_t i = n_ind; i < np; ++i) pl[i] = 0.0;
  }
}

void replay_missing_constants(double *u_lkl, double *u_pp) {
  double g[3];
  for (int k = 0; k < 3; ++k) g[k] = std::log((double)1 / 3);  // harden(): gen_func.cpp:903-905
  *u_lkl = std::exp(g[0]);                                    // ngsLD.cpp:110
  double pp[3] = {g[0], g[1], g[2]};
  normalise_log(pp);                                          // site_maf(): post_prob, then exp
  *u_pp = std::exp(pp[0]);
}

double replay_missing_raw_text() { return ngsld_host_missing_call_log(); }  // read_data.cpp:94: what host_io.cpp's reader stores

void replay_missing_constants_text(double *u_lkl, double *u_pp) {
  double g[3];
  for (int k = 0; k < 3; ++k) g[k] = replay_missing_raw_text();
  normalise_log(g);                                           // read_data.cpp:98
  *u_lkl = std::exp(g[0]);                                    // ngsLD.cpp:110
  double pp[3] = {g[0], g[1], g[2]};
  normalise_log(pp);                                          // site_maf(): post_prob, then exp
  *u_pp = std::exp(pp[0]);
}

void replay_site_from_lkl(const double *lkl, double maf, uint64_t n_ind, ReplaySite *out) {
  out->maf = maf;
  out->lkl.assign(lkl, lkl + 3 * n_ind);
  out->e.resize(n_ind);
  for (uint64_t i = 0; i < n_ind; ++i) out->e[i] = lkl[3 * i + 1] + 2 * lkl[3 * i + 2];
}

void replay_pair(const ReplaySite &a, const ReplaySite &b, uint64_t n_ind, bool ignore_miss, ngsld_rec_std *sr,
                 ngsld_rec_ext *er, int *status) {
  const double rho = correlation(a.e, b.e);  // pearson_r, ngsLD.cpp:365-367: pow(r, 2) is r * r under gcc -O1 and up
  double f[4];
  uint64_t x = 0, it = 0;
  const double m1 = a.maf, m2 = b.maf;
  if (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1) {  // gen_func.cpp:1030-1031: error() in the reference
    if (status) *status = NGSLD_ERR_MAF_RANGE;
    f[0] = f[1] = f[2] = f[3] = NAN;
  } else {
    f[0] = (1 - m1) * (1 - m2);  // gen_func.cpp:1034-1037
    f[1] = (1 - m1) * m2;
    f[2] = m1 * (1 - m2);
    f[3] = m1 * m2;
    for (it = 0; it < (uint64_t)kMaxIter; ++it) {  // gen_func.cpp:1041-1056
      double last[4], eps = 0;
      std::memcpy(last, f, sizeof(last));
      x = em_step(f, a.lkl.data(), b.lkl.data(), n_ind, ignore_miss);
      for (int k = 0; k < 4; ++k) {
        const double d = std::fabs(f[k] - last[k]);
        if (d > eps) eps = d;  // a NaN never raises eps: an all-NaN step ends the loop here
      }
      if (eps < kEps) break;
    }
  }
  // ngsLD.cpp:296-306
  double maf[2];
  maf[0] = 1 - (f[0] + f[1]);
  maf[1] = 1 - (f[0] + f[2]);
  const double D = f[0] * f[3] - f[1] * f[2];
  const double Dp = D / (D < 0 ? -ref_min(maf[0] * maf[1], (1 - maf[0]) * (1 - maf[1]))
                               : ref_min(maf[0] * (1 - maf[1]), (1 - maf[0]) * maf[1]));
  const double rr = D / std::sqrt(maf[0] * maf[1] * (1 - maf[0]) * (1 - maf[1]));
  sr->r2_ExpG = rho * rho;
  sr->D = D;
  sr->Dp = Dp;
  sr->r2 = rr * rr;
  if (er != nullptr) {
    for (int k = 0; k < 4; ++k) er->hap[k] = f[k];
    er->n_ind_data = (uint32_t)x;
    er->n_iter = (uint32_t)it;
  }
}

}  // namespace ngsld

// C door (include/ngsld_host.h): one pair in the reference's own arithmetic, no device needed
extern "C" int ngsld_host_replay_pair(const double *raw1, const double *raw2, uint64_t n_ind, const ngsld_geno_opts *opts,
                                      ngsld_rec_std *std_rec, ngsld_rec_ext *ext_rec, double *maf_out) try {
  if (raw1 == nullptr || raw2 == nullptr || n_ind == 0 || opts == nullptr || std_rec == nullptr) return NGSLD_ERR_INVALID;
  ngsld::ReplaySite a, b;
  ngsld::replay_site_from_raw(raw1, n_ind, *opts, &a);
  ngsld::replay_site_from_raw(raw2, n_ind, *opts, &b);
  int status = NGSLD_OK;
  ngsld::replay_pair(a, b, n_ind, opts->ignore_miss_data != 0, std_rec, ext_rec, &status);
  if (maf_out != nullptr) {
    maf_out[0] = a.maf;
    maf_out[1] = b.maf;
  }
  return status;
} catch (...) {
  return NGSLD_ERR_NOMEM;
}
