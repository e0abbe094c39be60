// replay.h -- exact-order replay of single sites and single pairs on the host.
//
// The HIP kernels compute every pair with reordered (tree / FMA) arithmetic, which agrees with the reference to
// ~1e-15 wherever the outcome is well conditioned.  A few outcomes are decided by the reference's own rounding
// noise: D' and r2 of a pair with a site (nearly) monomorphic in the estimated haplotypes (0/0-type quotients,
// ngsLD.cpp:296-306), the iteration at which eps crosses EPSILON when it lands within 1e-12 of it
// (gen_func.cpp:1054), the maf[s] < min_maf tests when a frequency ties the threshold (ngsLD.cpp:264-275) and
// the correlation of expected genotypes at a site whose expected genotypes are constant up to rounding
// (ngsLD.cpp:365-367).  The kernels flag those pairs; the engine re-evaluates them here, one at a time, in the
// reference's own operation order -- sequential sums over individuals, the 16-term `sum`, tmp / sum, ff / (2x),
// the sequential renormalisation, no fused multiply-add -- and overwrites their records.
//
// This file is product code (host side of libngsld.so), written against the reference's source text.
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/ngsld.h"

namespace ngsld {

// One site as the reference's main() holds it when calc_pair_LD runs (ngsLD.cpp:86-114).
struct ReplaySite {
  std::vector<double> lkl;  // [n_ind][3] normal space: pars->geno_lkl[s] after conv_space(exp), ngsLD.cpp:110
  std::vector<double> e;    // [n_ind] expected genotypes, ngsLD.cpp:113
  double maf = 0.0;         // est_maf on the log-space values, gen_func.cpp:974-1009
};

// raw: [n_ind][3] values as ngsld_set_geno_raw_opts takes them.  Does what read_geno (read_data.cpp:37-45 binary,
// :83-99 text), call_geno (ngsLD.cpp:92-98), est_maf (ngsLD.cpp:104-105) and the exp loop (ngsLD.cpp:107-114) do
// to one site, with the host's libm and the reference's operation order.
void replay_site_from_raw(const double *raw, uint64_t n_ind, const ngsld_geno_opts &opts, ReplaySite *out);

// The same site written as the device holds it: planes[g * np + i] = lkl[i][g] for i < n_ind, zero up to np; *maf = est_maf.
// What the exact store of the device-side replay (ld_replay_lkl.hip) is built from, site by site, on the replay threads.
void replay_site_planes(const double *raw, uint64_t n_ind, const ngsld_geno_opts &opts, uint64_t np, double *planes, double *maf);

// The same from values that are already normalised normal-space likelihoods + maf (ngsld_set_geno_lkl, or the
// device's own planes read back when the caller registered no source): only the expected genotypes are derived.
void replay_site_from_lkl(const double *lkl, double maf, uint64_t n_ind, ReplaySite *out);

// calc_pair_LD's arithmetic for one pair (ngsLD.cpp:290-306): pearson_r, haplo_freq, D, D', r2.
// status (may be null) is set to NGSLD_ERR_MAF_RANGE where haplo_freq would call error() (gen_func.cpp:1030).
void replay_pair(const ReplaySite &a, const ReplaySite &b, uint64_t n_ind, bool ignore_miss_data, ngsld_rec_std *std_rec,
                 ngsld_rec_ext *ext_rec, int *status);

// The "no data" triple call_geno leaves (gen_func.cpp:903-905: log(1/3) three times) as the HOST's libm turns it into the
// values calc_pair_LD and est_maf see: u_lkl = exp(log(1/3)) (ngsLD.cpp:110), u_pp = the est_maf posterior of that triple
// (gen_func.cpp:920-932, 986-990).  The device-side replay of called genotypes (ld_replay.hip) takes them as constants.
void replay_missing_constants(double *u_lkl, double *u_pp);
// ... and of a text genotype file's missing call (read_data.cpp:94-98: log(1/3) three times, THEN post_prob; no --call_geno):
// the raw value the reader stores, and what the site's likelihood and est_maf's posterior of such an individual are
double replay_missing_raw_text();
void replay_missing_constants_text(double *u_lkl, double *u_pp);

}  // namespace ngsld
