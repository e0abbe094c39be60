/*
 * ngsld_oracle.h -- CPU ORACLE for the ngsLD pairwise-LD hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (fgvieira/ngsLD v1.2.1), written by
 * reading the reference and citing the file:line each function follows.  It is the checker the
 * HIP path is compared against.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load or execute anything under oracle/; the product (ngsld_amd/, include/) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - EM (haplo_freq / pair_freq_iter), est_maf, post_prob/logsum, conv_space, miss_data, the binary
 *     GL reader and the pos reader are checked BIT-FOR-BIT against the reference's own code compiled
 *     from /root/reference (oracle/build_ref.sh -> oracle/_ref/libngsld_ref.so; only the GSL-free part
 *     of shared/gen_func.cpp + shared/read_data.cpp can be built here, GSL is not installed and no
 *     stand-in for it is written).  Golden vectors from that build are committed under tests/golden/.
 *   - ngsLD.cpp as a whole needs GSL to compile, but its in-tree arithmetic does not: since round 4
 *     build_ref.sh cuts the GSL-free line ranges of calc_pair_LD out of the reference's file by anchor
 *     and compiles them verbatim -- the s2 walk with its running dist and filters (ngsLD.cpp:240-275),
 *     D / D' / r2 / hap_maf (:296-306), the float chi2 (:328-333), both fprintf formats (:314-351) and
 *     the header line (:77).  orc_row / orc_pair_stats / orc_print_pair / orc_print_header are checked
 *     BIT-FOR-BIT / BYTE-FOR-BYTE against them (tests/test_oracle_vs_ref.py: 10^5 haplotype vectors
 *     incl. degenerate ones, 8,000 formatted rows, six filter sets), and every golden row was held to
 *     the reference's own fprintf lines when the fixtures were generated.
 *   - main() and calc_pair_LD WHOLE (ngsLD.cpp:27-359) are compiled too, minus the statements that need GSL (gsl_rng: dropped;
 *     the --rnd_sample block: dropped; the one pearson_r call: a lookup of the caller's value): ref_main runs the reference's
 *     program flow on real files, and tests/test_ref_main_run.py holds the oracle CLI's TSV to its output byte for byte on 15
 *     fixtures (r2_ExpG handed in from the oracle).
 *   - parse_args.cpp (option table, defaults, argument echo, validation messages) is compiled whole into oracle/_ref as
 *     well; tests/test_cli_args_vs_ref.py holds the drop-in binary's parser to it (exit status and stderr, 23 argv vectors).
 *   - PARITY UNPINNED at the GSL boundary only: the Pearson r2 of expected genotypes
 *     (gsl_stats_correlation, ngsLD.cpp:365-367) and the --rnd_sample draws (gsl_rng_taus,
 *     ngsLD.cpp:69-70,165-166,277).  GSL is absent from /root/reference and from the image; both are
 *     restated from GSL's published algorithms (the generator reproduces GSL's published self-test
 *     value, the correlation is held to the textbook formula at 1e-12).
 */
#ifndef NGSLD_ORACLE_H
#define NGSLD_ORACLE_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* constants: shared/gen_func.hpp:14-18 */
#define ORC_N_GENO 3
#define ORC_INF 1e15
#define ORC_EPSILON 1e-5
#define ORC_ITER_MAX 100

/* mirrors `params` (ngsLD.hpp:11-44), flat arrays instead of jagged pointers */
typedef struct {
  const char *in_geno;
  int in_logscale;
  uint64_t n_ind;
  uint64_t n_sites;
  const char *in_pos;
  int in_pos_header;
  uint64_t max_kb_dist;
  uint64_t max_snp_dist;
  double min_maf;
  int ignore_miss_data;
  int extend_out;
  int n_threads;
  double rnd_sample; /* 1 = keep every pair (default, parse_args.cpp:22); < 1: per-pair Tausworthe draw */
  uint64_t seed;     /* --seed: seeds the master stream (ngsLD.cpp:69-70) */

  double *geno_lkl;      /* [n_sites][n_ind][3]; log space after read, normal space after preprocess */
  double *maf;           /* [n_sites] */
  double *expected_geno; /* [n_sites][n_ind] */
  double *pos_dist;      /* [n_sites] */
  char **labels;         /* [n_sites] or NULL */
} orc_params;

/* one result per surviving pair (full precision; what calc_pair_LD would print) */
typedef struct {
  uint64_t s1, s2;
  double dist;
  double r2pear, D, Dp, r2;
  uint64_t n_ind_data;
  double hap[4];
  double hap_maf[2];
  float chi2;
  uint64_t n_iter;
} orc_pair;

/* --- math kernels (each cites the reference lines it follows in the .c file) --- */
double orc_logsum(const double *a, uint64_t n);
void orc_post_prob(double *pp, const double *lkl, uint64_t n_geno);
void orc_conv_space_log(double *g, int n);
void orc_conv_space_exp(double *g, int n);
int orc_miss_data(const double *g);
double orc_est_maf(uint64_t n_ind, const double *pdg /* [n_ind][3] log space */, int ignore_miss_data);
uint64_t orc_pair_freq_iter(double f[4], const double *s1, const double *s2, uint64_t n, int ignore_miss_data,
                            int *err);
uint64_t orc_haplo_freq(double hap_freq[4], uint64_t *n, const double *gl1, const double *gl2, double maf1,
                        double maf2, uint64_t n_ind, int ignore_miss_data, int *err);
double orc_correlation(const double *x, const double *y, uint64_t n);
double orc_pearson_r2(const double *x, const double *y, uint64_t n);
void orc_pair_stats(const double hap[4], double *D, double *Dp, double *r2, double hap_maf[2], float *chi2);

/* --- data stages --- */
/* binary GL reader: returns 0 ok, <0 error (message in errbuf). out = [n_sites][n_ind][3] log-normalised */
int orc_read_geno_bin(const char *path, int log_scale, uint64_t n_ind, uint64_t n_sites, double *out, char *errbuf,
                      size_t errlen);
/* text (.gz or plain) reader, read_data.cpp:48-104: GL/posterior triples (in_probs) or called genotypes
   {-1,0,1,2}; uses the last n_ind*n_geno numeric fields of every line; out = [n_sites][n_ind][3] log-normalised */
int orc_read_geno_text(const char *path, int in_probs, int log_scale, uint64_t n_ind, uint64_t n_sites, double *out,
                       char *errbuf, size_t errlen);
/* gen_func.cpp:886-914 call_geno on one log-space triple (miss_data mode 0, as ngsLD.cpp:97 calls it) */
void orc_call_geno(double *geno, double N_thresh, double call_thresh);
/* ngsLD.cpp:92-98 over the whole matrix (log space, before orc_preprocess) */
void orc_call_geno_all(orc_params *p, double N_thresh, double call_thresh);
/* same arithmetic as the reader on an in-memory raw buffer (raw is not modified) */
int orc_normalise_raw(const double *raw, int log_scale, uint64_t n_ind, uint64_t n_sites, double *out);
/* ngsLD.cpp:103-114: maf on log GL, exp() in place, expected genotypes */
void orc_preprocess(orc_params *p);
/* read_dist + labels; allocates p->pos_dist, p->labels.  returns 0 ok */
int orc_read_pos(orc_params *p, char *errbuf, size_t errlen);
void orc_free_pos(orc_params *p);

/* --- gsl_rng_taus (GSL is not in the reference tree): L'Ecuyer's 3-component Tausworthe generator restated from
   GSL's published rng/taus.c; pinned by GSL's own self-test value (seed 1, 10000th output 2733957125) --- */
typedef struct {
  uint32_t s1, s2, s3;
} orc_taus;
void orc_taus_set(orc_taus *r, unsigned long seed);
uint32_t orc_taus_get(orc_taus *r);
double orc_taus_uniform(orc_taus *r);
/* per-row seeds: for s1 = 0,1,2,... (unsigned long) draw_rnd(master, 0, INF)  (ngsLD.cpp:166, gen_func.cpp:117-119) */
void orc_row_seeds(uint64_t seed, uint64_t n_sites, uint64_t *out);

/* --- the hot path --- */
/* number of pairs row s1 emits; when out != NULL the records are written there (capacity cap). */
uint64_t orc_row(const orc_params *p, uint64_t s1, orc_pair *out, uint64_t cap, int *err);
/* all rows [s1_begin, s1_end), n_threads pthreads striped over rows. returns #pairs; records optional,
   written in (s1,s2) order when out != NULL (two-pass: count, then fill). */
uint64_t orc_run(const orc_params *p, uint64_t s1_begin, uint64_t s1_end, orc_pair *out, uint64_t cap, int *err);
/* cpu_baseline leg of bench.py: compute every pair of rows [s1_begin, s1_end) with n_threads pthreads and
   keep only a checksum (sum of finite r2) so nothing is stored; returns #pairs. */
uint64_t orc_bench(const orc_params *p, uint64_t s1_begin, uint64_t s1_end, double *checksum, uint64_t *iters);
/* the walk + pearson_r (gsl_stats_correlation restated) alone over the same rows: (#pairs; checksum = sum of finite r2_ExpG) */
uint64_t orc_bench_pearson(const orc_params *p, uint64_t s1_begin, uint64_t s1_end, double *checksum);
/* exclusive end of the contiguous s2 range row s1 walks (before the maf[s2] skip) */
uint64_t orc_row_end(const orc_params *p, uint64_t s1);

/* --- text --- */
void orc_print_header(FILE *fh, int extend_out);
void orc_print_pair(FILE *fh, const orc_params *p, const orc_pair *r);
long orc_format_header(char *buf, size_t cap, int extend_out);
long orc_format_pair(char *buf, size_t cap, const orc_params *p, const orc_pair *r);

#ifdef __cplusplus
}
#endif
#endif
