/*
 * ngsld_oracle_main.c -- command-line front end of the CPU ORACLE (test infrastructure).
 *
 * Accepts the reference's flags (parse_args.cpp:35-59, defaults :6-29, validation :168-183) and
 * writes the reference's TSV (ngsLD.cpp:77,314-351) for BINARY GL input, so that the product's CLI
 * can be compared text-for-text (sorted md5) against it.  Not built into, linked by, or shipped with
 * the product.
 *
 * Extra flag (oracle only): --dump FILE writes the full-precision records (orc_pair structs).
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/stat.h>

#include "ngsld_oracle.h"

static void die(const char *func, const char *msg) { /* gen_func.cpp:12-18 */
  fflush(stdout);
  fprintf(stderr, "\n=====\nERROR: [%s] %s\n=====\n\n", func, msg);
  perror("\t");
  fflush(stderr);
  exit(-1);
}

int main(int argc, char **argv) {
  orc_params P;
  memset(&P, 0, sizeof(P));
  P.max_kb_dist = 100;
  P.n_threads = 1;
  P.rnd_sample = 1;
  P.seed = 12345; /* the reference defaults to time(NULL) + rand() % 1000 (parse_args.cpp:23): tests always pass --seed */
  const char *out = NULL, *dump = NULL;
  int verbose = 1, in_probs = 0, call_geno = 0;
  double rnd_sample = 1, N_thresh = 0, call_thresh = 0;

  static struct option lopts[] = {{"geno", required_argument, NULL, 'g'},
                                  {"probs", no_argument, NULL, 'p'},
                                  {"log_scale", no_argument, NULL, 'l'},
                                  {"n_ind", required_argument, NULL, 'n'},
                                  {"n_sites", required_argument, NULL, 's'},
                                  {"pos", required_argument, NULL, 'a'},
                                  {"posH", required_argument, NULL, 'A'},
                                  {"max_kb_dist", required_argument, NULL, 'd'},
                                  {"max_snp_dist", required_argument, NULL, 'D'},
                                  {"min_maf", required_argument, NULL, 'f'},
                                  {"ignore_miss_data", no_argument, NULL, 'm'},
                                  {"call_geno", no_argument, NULL, 'c'},
                                  {"N_thresh", required_argument, NULL, 'N'},
                                  {"call_thresh", required_argument, NULL, 'C'},
                                  {"rnd_sample", required_argument, NULL, 'r'},
                                  {"seed", required_argument, NULL, 'S'},
                                  {"extend_out", no_argument, NULL, 'x'},
                                  {"out", required_argument, NULL, 'o'},
                                  {"outH", required_argument, NULL, 'O'},
                                  {"n_threads", required_argument, NULL, 't'},
                                  {"verbose", required_argument, NULL, 'V'},
                                  {"dump", required_argument, NULL, 1000},
                                  {0, 0, 0, 0}};
  int c;
  while ((c = getopt_long_only(argc, argv, "g:pln:s:Z:d:D:f:mcN:C:r:S:xo:t:V:", lopts, NULL)) != -1) switch (c) {
      case 'g': P.in_geno = optarg; break;
      case 'p': in_probs = 1; break;
      case 'l': P.in_logscale = 1; in_probs = 1; break;
      case 'n': P.n_ind = (uint64_t)atoi(optarg); break;
      case 's': P.n_sites = (uint64_t)atoi(optarg); break;
      case 'a': P.in_pos = optarg; P.in_pos_header = 0; break;
      case 'A': P.in_pos = optarg; P.in_pos_header = 1; break;
      case 'd': P.max_kb_dist = (uint64_t)atoi(optarg); break;
      case 'D': P.max_snp_dist = (uint64_t)atoi(optarg); break;
      case 'f': P.min_maf = atof(optarg); break;
      case 'm': P.ignore_miss_data = 1; break;
      case 'c': call_geno = 1; break;
      case 'N': N_thresh = atof(optarg); call_geno = 1; break;
      case 'C': call_thresh = atof(optarg); call_geno = 1; break;
      case 'r': rnd_sample = atof(optarg); break;
      case 'S': P.seed = (uint64_t)atoi(optarg); break;
      case 'x': P.extend_out = 1; break;
      case 'o': out = optarg; break;
      case 't': P.n_threads = atoi(optarg); break;
      case 'V': verbose = atoi(optarg); break;
      case 1000: dump = optarg; break;
      default: exit(-1); /* includes --outH, which has no case in the reference (parse_args.cpp:55,130) */
    }
  (void)verbose;

  if (P.in_geno == NULL) die("parse_cmd_args", "genotype input file (--geno) missing!");
  if (P.n_ind == 0) die("parse_cmd_args", "number of individuals (--n_ind) missing!");
  if (P.n_sites == 0) die("parse_cmd_args", "number of sites (--n_sites) missing!");
  if (P.in_pos == NULL && P.max_kb_dist > 0)
    die("parse_cmd_args", "position file necessary in order to filter by maximum distance!");
  if (P.min_maf < 0 || P.min_maf > 1) die("parse_cmd_args", "minimum allele frequency must be in [0,1]!");
  if (call_geno && !in_probs) die("parse_cmd_args", "can only call genotypes from likelihoods/probabilities!");
  if (rnd_sample <= 0 || rnd_sample > 1)
    die("parse_cmd_args", "proportion of comparisons to sample must be in ]0,1]!");
  if (P.n_threads < 1) die("parse_cmd_args", "number of threads cannot be less than 1!");
  P.rnd_sample = rnd_sample;

  struct stat st;
  if (stat(P.in_geno, &st) != 0) die("main", "cannot check GENO file size!");
  const char *dot = strrchr(P.in_geno, '.');
  int in_bin = !(dot != NULL && strcmp(dot, ".gz") == 0); /* ngsLD.cpp:45-57 */
  if (in_bin) {
    in_probs = 1;
    if (P.n_sites != (uint64_t)st.st_size / sizeof(double) / P.n_ind / ORC_N_GENO) /* ngsLD.cpp:55-56 */
      die("main", "invalid/corrupt genotype input file!");
  }

  FILE *fh = stdout;
  if (out != NULL) fh = fopen(out, "w");
  if (fh == NULL) die("main", "cannot open output file!");
  orc_print_header(fh, P.extend_out);

  char err[256];
  P.geno_lkl = (double *)malloc(P.n_sites * P.n_ind * 3 * sizeof(double));
  P.maf = (double *)malloc(P.n_sites * sizeof(double));
  P.expected_geno = (double *)malloc(P.n_sites * P.n_ind * sizeof(double));
  if (in_bin ? orc_read_geno_bin(P.in_geno, P.in_logscale, P.n_ind, P.n_sites, P.geno_lkl, err, sizeof(err))
             : orc_read_geno_text(P.in_geno, in_probs, P.in_logscale, P.n_ind, P.n_sites, P.geno_lkl, err, sizeof(err)))
    die("read_geno", err);
  if (call_geno) { /* ngsLD.cpp:92-98 */
    if (N_thresh > call_thresh)
      die("call_geno", "missing data threshold must be smaller than calling genotype threshold!");
    orc_call_geno_all(&P, N_thresh, call_thresh);
  }
  orc_preprocess(&P);
  if (P.in_pos) {
    if (orc_read_pos(&P, err, sizeof(err))) /* the reference's __FUNCTION__: read_file opens (gen_func.cpp:244-246), read_split counts fields (read_data.cpp:145-146) */
      die(!strcmp(err, "cannot open file!") ? (access(P.in_pos, R_OK) == 0 ? "read_split" : "read_file") : !strcmp(err, "invalid number of fields in file!") ? "read_split" : "read_dist", err);
  } else {
    P.pos_dist = (double *)malloc(P.n_sites * sizeof(double));
    for (uint64_t s = 0; s < P.n_sites; s++) P.pos_dist[s] = INFINITY; /* ngsLD.cpp:134 */
    P.labels = NULL;
  }

  int e = 0;
  uint64_t n = orc_run(&P, 0, P.n_sites, NULL, 0, &e);
  orc_pair *rec = (orc_pair *)malloc((n ? n : 1) * sizeof(orc_pair));
  orc_run(&P, 0, P.n_sites, rec, n, &e);
  if (e == 1) die("pair_freq_iter", "invalid number of individuals!");
  if (e == 2) die("haplo_freq", "invalid allele frequencies");
  for (uint64_t k = 0; k < n; k++) orc_print_pair(fh, &P, &rec[k]);
  if (dump) {
    FILE *d = fopen(dump, "wb");
    if (d == NULL) die("main", "cannot open dump file!");
    fwrite(rec, sizeof(orc_pair), n, d);
    fclose(d);
  }
  if (fh != stdout) fclose(fh);
  free(rec);
  return 0;
}
