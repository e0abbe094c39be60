/*
 * ref_shim.cpp -- extern "C" doors into the REFERENCE's own functions (test infrastructure).
 *
 * This file is appended by oracle/build_ref.sh to a translation unit made of the reference's
 * shared/gen_func.{hpp,cpp}, shared/read_data.{hpp,cpp}, shared/threadpool.h, ngsLD.hpp (GSL lines dropped) and the
 * GSL-free line ranges of ngsLD.cpp that build_ref.sh wraps into ref_walk_row / ref_pair_stats / ref_format_row /
 * ref_print_header -- all read where they lie under /root/reference at build time (never copied into the repo), so
 * every function called below is the reference's code, compiled here.  The shim itself adds no arithmetic except the flat <-> jagged
 * array plumbing and, in ref_preprocess, the one expression of ngsLD.cpp:113 (p1 + 2*p2) that lives
 * in main() and cannot be compiled without GSL.
 *
 * NB: the reference headers define abs/min/max as MACROS (gen_func.hpp:21-23); do not use those
 * names here and do not include further C++ standard headers (pthread.h is plain C and untouched by them).
 */
#include <pthread.h>

static double **ref_rows(const double *flat, uint64_t n) {
  double **p = new double *[n];
  for (uint64_t i = 0; i < n; i++) p[i] = const_cast<double *>(flat) + 3 * i;
  return p;
}

extern "C" {

/* The s2 walk of calc_pair_LD (ngsLD.cpp:240-275, compiled verbatim into ref_walk_row by build_ref.sh) for row `site`, from
 * flat arguments: the `params` the reference's lines read is filled here (verbose = 0: labels and expected genotypes are
 * not touched).  Returns the number of pairs that pass the distance / SNP-count / maf filters; their s2 and running dist. */
uint64_t ref_walk(uint64_t n_sites, double *pos_dist, double *maf, uint64_t max_kb_dist, uint64_t max_snp_dist,
                  double min_maf, uint64_t site, uint64_t *out_s2, double *out_dist, uint64_t cap) {
  params pars;
  memset(&pars, 0, sizeof(pars));
  pars.n_sites = n_sites;
  pars.pos_dist = pos_dist;
  pars.maf = maf;
  pars.max_kb_dist = max_kb_dist;
  pars.max_snp_dist = max_snp_dist;
  pars.min_maf = min_maf;
  return ref_walk_row(&pars, site, out_s2, out_dist, cap);
}

/* r2_ExpG for ref_main's calc_pair_LD: the one value of a row's output that comes from GSL.  The caller supplies it per pair
 * ((s1, s2) in increasing order, first[s1] = index of row s1's first pair, first[n_sites] = number of pairs); a pair that is
 * not in the table gets NaN. */
static const uint64_t *g_r2_first = NULL, *g_r2_s2 = NULL;
static const double *g_r2_val = NULL;
static uint64_t g_r2_sites = 0;
void ref_set_r2pear(const uint64_t *first, const uint64_t *s2, const double *val, uint64_t n_sites) {
  g_r2_first = first; g_r2_s2 = s2; g_r2_val = val; g_r2_sites = n_sites;
}
double ref_r2pear_lookup(uint64_t s1, uint64_t s2) {
  if (g_r2_first == NULL || s1 >= g_r2_sites) return NAN;
  uint64_t lo = g_r2_first[s1], hi = g_r2_first[s1 + 1];
  while (lo < hi) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (g_r2_s2[mid] < s2) lo = mid + 1; else hi = mid;
  }
  return (lo < g_r2_first[s1 + 1] && g_r2_s2[lo] == s2) ? g_r2_val[lo] : NAN;
}

/* The reference's own command-line parser (parse_args.cpp, compiled whole by build_ref.sh): init_pars + parse_cmd_args on
 * argv.  Whatever they print goes to stderr as in the reference, an invalid argument ends the PROCESS through error()
 * (gen_func.cpp:12-18: exit(-1)) -- callers run this in a child process; on return the parsed fields are printed on stdout
 * as one line (seed left out: its default is the clock). */
int ref_parse_args(int argc, char **argv) {
  params pars;
  init_pars(&pars);
  optind = 1;
  parse_cmd_args(&pars, argc, argv);
  printf("PARSED geno=%s probs=%d log_scale=%d n_ind=%lu n_sites=%lu pos=%s posH=%d max_kb_dist=%lu max_snp_dist=%lu min_maf=%.17g "
         "ignore_miss_data=%d call_geno=%d N_thresh=%.17g call_thresh=%.17g rnd_sample=%.17g extend_out=%d out=%s n_threads=%u verbose=%u\n",
         pars.in_geno ? pars.in_geno : "(null)", (int)pars.in_probs, (int)pars.in_logscale, (unsigned long)pars.n_ind,
         (unsigned long)pars.n_sites, pars.in_pos ? pars.in_pos : "(null)", (int)pars.in_pos_header,
         (unsigned long)pars.max_kb_dist, (unsigned long)pars.max_snp_dist, pars.min_maf, (int)pars.ignore_miss_data,
         (int)pars.call_geno, pars.N_thresh, pars.call_thresh, pars.rnd_sample, (int)pars.extend_out,
         pars.out ? pars.out : "(null)", pars.n_threads, pars.verbose);
  fflush(stdout);
  return 0;
}

double ref_logsum(double *a, uint64_t n) { return logsum(a, n); }
void ref_post_prob(double *pp, double *lkl, uint64_t n) { post_prob(pp, lkl, NULL, n); }
int ref_miss_data(double *g) { return miss_data(g) ? 1 : 0; }
void ref_conv_space_log(double *g, int n) { conv_space(g, n, log); }
void ref_conv_space_exp(double *g, int n) { conv_space(g, n, exp); }

double ref_est_maf(uint64_t n_ind, const double *pdg, int ignore_miss) {
  double **rows = ref_rows(pdg, n_ind);
  double m = est_maf(n_ind, rows, (double *)NULL, ignore_miss != 0);
  delete[] rows;
  return m;
}

uint64_t ref_pair_freq_iter(double f[4], const double *gl1, const double *gl2, uint64_t n, int ignore_miss) {
  double **a = ref_rows(gl1, n), **b = ref_rows(gl2, n);
  uint64_t x = pair_freq_iter(f, a, b, n, ignore_miss != 0);
  delete[] a;
  delete[] b;
  return x;
}

uint64_t ref_haplo_freq(double hap[4], uint64_t *n, const double *gl1, const double *gl2, double maf1, double maf2,
                        uint64_t n_ind, int ignore_miss) {
  double **a = ref_rows(gl1, n_ind), **b = ref_rows(gl2, n_ind);
  double loglkl = 0;
  uint64_t it = haplo_freq(hap, &loglkl, n, a, b, maf1, maf2, n_ind, ignore_miss != 0, false); /* ngsLD.cpp:294 */
  delete[] a;
  delete[] b;
  return it;
}

/* read_geno (binary) + transp_matrix, flattened to [site][ind][3] (log space, normalised) */
int ref_read_geno_bin(const char *path, int log_scale, uint64_t n_ind, uint64_t n_sites, double *out) {
  bool ls = log_scale != 0;
  double ***tmp = read_geno(const_cast<char *>(path), true, true, &ls, n_ind, n_sites); /* ngsLD.cpp:87 */
  double ***g = transp_matrix(tmp, n_ind, n_sites);                                    /* ngsLD.cpp:88 */
  for (uint64_t s = 0; s < n_sites; s++)
    for (uint64_t i = 0; i < n_ind; i++)
      for (int k = 0; k < 3; k++) out[(s * n_ind + i) * 3 + k] = g[s][i][k];
  free_ptr((void ***)tmp, n_ind, n_sites);
  free_ptr((void **)g, n_sites);
  return 0;
}

/* read_geno, text branch (in_bin = false), flattened like ref_read_geno_bin */
int ref_read_geno_text(const char *path, int in_probs, int log_scale, uint64_t n_ind, uint64_t n_sites, double *out) {
  bool ls = log_scale != 0;
  double ***tmp = read_geno(const_cast<char *>(path), false, in_probs != 0, &ls, n_ind, n_sites);
  double ***g = transp_matrix(tmp, n_ind, n_sites);
  for (uint64_t s = 0; s < n_sites; s++)
    for (uint64_t i = 0; i < n_ind; i++)
      for (int k = 0; k < 3; k++) out[(s * n_ind + i) * 3 + k] = g[s][i][k];
  free_ptr((void ***)tmp, n_ind, n_sites);
  free_ptr((void **)g, n_sites);
  return 0;
}

/* ngsLD.cpp:97: call_geno(geno, N_GENO, in_logscale = true, N_thresh, call_thresh, 0) on one triple */
void ref_call_geno(double *geno, double N_thresh, double call_thresh) {
  call_geno(geno, N_GENO, true, N_thresh, call_thresh, 0);
}

/* ngsLD.cpp:103-114 with the reference's est_maf / conv_space; in/out flat [site][ind][3] */
void ref_preprocess(double *gl, uint64_t n_ind, uint64_t n_sites, int ignore_miss, double *maf, double *expg) {
  for (uint64_t s = 0; s < n_sites; s++) maf[s] = ref_est_maf(n_ind, gl + s * n_ind * 3, ignore_miss);
  for (uint64_t s = 0; s < n_sites; s++)
    for (uint64_t i = 0; i < n_ind; i++) {
      double *t = gl + (s * n_ind + i) * 3;
      conv_space(t, N_GENO, exp);
      expg[s * n_ind + i] = t[1] + 2 * t[2];
    }
}

int ref_read_dist(const char *path, int header, uint64_t n_sites, double *out) {
  double *d = read_dist(const_cast<char *>(path), header ? 1 : 0, n_sites); /* ngsLD.cpp:120 */
  for (uint64_t s = 0; s < n_sites; s++) out[s] = d[s];
  delete[] d;
  return 0;
}

/* ngsLD.cpp:124-132: labels = lines of the pos file with the first TAB turned into ':' */
uint64_t ref_read_labels(const char *path, int header, char *out, uint64_t stride, uint64_t cap) {
  char **lines = NULL;
  uint64_t n = read_file(path, &lines, header ? 1 : 0);
  for (uint64_t s = 0; s < n && s < cap; s++) {
    char *t = strchr(lines[s], '\t');
    if (t != NULL) *t = ':';
    strncpy(out + s * stride, lines[s], stride - 1);
    out[s * stride + stride - 1] = '\0';
  }
  return n;
}

/* Timing door (bench.py cpu_baseline, kind "reference-subset"): the reference's OWN compiled haplo_freq
 * (gen_func.cpp:1027, the EM that dominates calc_pair_LD) over every pair (s1, s2) with s1 in [s1_begin, s1_end),
 * s2 in (s1, row_end[s1]), rows dealt round-robin to n_threads pthreads -- the granularity of the reference's own
 * thread pool (one calc_pair_LD job per s1, ngsLD.cpp:159-186).  gl: normal-space [site][ind][3] as calc_pair_LD sees
 * it; the jagged row pointers haplo_freq wants are built once per site.  pearson_r (GSL) and the fprintf are not
 * part of it: this is a SUBSET of the reference's per-pair work, hence an upper bound on its pairs/s. */
struct ref_bench_job {
  double ***rows;
  const double *maf;
  const uint64_t *row_end;
  uint64_t n_ind, s1_begin, s1_end, pairs, iters;
  int tid, n_threads, ignore_miss;
  double sum;
};

static void *ref_bench_worker(void *arg) {
  ref_bench_job *j = (ref_bench_job *)arg;
  for (uint64_t s1 = j->s1_begin + (uint64_t)j->tid; s1 < j->s1_end; s1 += (uint64_t)j->n_threads)
    for (uint64_t s2 = s1 + 1; s2 < j->row_end[s1]; s2++) {
      double hap[4], loglkl = 0;
      uint64_t n = 0;
      uint64_t it = haplo_freq(hap, &loglkl, &n, j->rows[s1], j->rows[s2], j->maf[s1], j->maf[s2], j->n_ind,
                               j->ignore_miss != 0, false);
      j->sum += hap[0];
      j->iters += it < ITER_MAX ? it + 1 : ITER_MAX;
      j->pairs++;
    }
  return NULL;
}

uint64_t ref_bench_haplo_freq(const double *gl, const double *maf, const uint64_t *row_end, uint64_t n_ind,
                              uint64_t n_sites, uint64_t s1_begin, uint64_t s1_end, int ignore_miss, int n_threads,
                              uint64_t *iters, double *checksum) {
  if (n_threads < 1) n_threads = 1;
  double ***rows = new double **[n_sites];
  for (uint64_t s = 0; s < n_sites; s++) rows[s] = ref_rows(gl + s * n_ind * 3, n_ind);
  pthread_t *th = new pthread_t[n_threads];
  ref_bench_job *jobs = new ref_bench_job[n_threads];
  for (int t = 0; t < n_threads; t++) {
    ref_bench_job q = {rows, maf, row_end, n_ind, s1_begin, s1_end, 0, 0, t, n_threads, ignore_miss, 0.0};
    jobs[t] = q;
    pthread_create(&th[t], NULL, ref_bench_worker, &jobs[t]);
  }
  uint64_t pairs = 0, it = 0;
  double sum = 0;
  for (int t = 0; t < n_threads; t++) {
    pthread_join(th[t], NULL);
    pairs += jobs[t].pairs;
    it += jobs[t].iters;
    sum += jobs[t].sum;
  }
  if (iters) *iters = it;
  if (checksum) *checksum = sum;
  for (uint64_t s = 0; s < n_sites; s++) delete[] rows[s];
  delete[] rows;
  delete[] th;
  delete[] jobs;
  return pairs;
}

} /* extern "C" */
