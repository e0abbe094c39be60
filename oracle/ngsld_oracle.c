/*
 * ngsld_oracle.c -- CPU ORACLE (test infrastructure, see ngsld_oracle.h for the pinning status).
 *
 * Plain-C restatement of the ngsLD v1.2.1 pair-LD path.  Every function cites the reference
 * file:line it follows (paths relative to /root/reference).  Operation order is kept identical to
 * the reference wherever the reference is in-tree code, so that this file is bit-identical to the
 * reference's own functions when both are built with the same compiler flags (checked by
 * tests/test_oracle_vs_ref.py against oracle/_ref/libngsld_ref.so, and frozen in tests/golden/).
 *
 * Build: see oracle/Makefile (-O3 -ffp-contract=off, no -march flags: x86-64 baseline has no FMA,
 * the same as the reference's `g++ -O3`, Makefile:9).
 */
#define _GNU_SOURCE
#include "ngsld_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* gen_func.hpp:21-23 -- the reference uses these MACROS (not fabs/fmin/fmax); NaN behaviour differs
   from libm, so restate them literally. */
#define ORC_ABS(x) ((x) >= 0 ? (x) : -(x))
#define ORC_MIN(a, b) ((a) <= (b) ? (a) : (b))
#define ORC_MAX(a, b) ((a) >= (b) ? (a) : (b))

#define ORC_BUFF_LEN 500000 /* gen_func.hpp:17 */

/* ------------------------------------------------------------------------------------------------
 * gen_func.cpp:135-151  logsum(double*, n): max-shifted log-sum-exp; all -inf -> -inf
 * ---------------------------------------------------------------------------------------------- */
double orc_logsum(const double *a, uint64_t n) {
  double sum = 0;
  double M = a[0];
  for (uint64_t i = 1; i < n; i++) M = ORC_MAX(a[i], M);
  if (M == -INFINITY) return -INFINITY;
  for (uint64_t i = 0; i < n; i++) sum += exp(a[i] - M);
  return log(sum) + M;
}

/* gen_func.cpp:920-932  post_prob(pp, lkl, prior=NULL, n_geno): pp = lkl - logsum(lkl) */
void orc_post_prob(double *pp, const double *lkl, uint64_t n_geno) {
  for (uint64_t c = 0; c < n_geno; c++) pp[c] = lkl[c];
  double norm = orc_logsum(pp, n_geno);
  for (uint64_t c = 0; c < n_geno; c++) pp[c] -= norm;
}

/* gen_func.cpp:123-130  conv_space(geno, n, log): -inf is replaced by -INF (= -1e15) */
void orc_conv_space_log(double *g, int n) {
  for (int k = 0; k < n; k++) {
    g[k] = log(g[k]);
    if (g[k] == -INFINITY) g[k] = -ORC_INF;
  }
}

/* gen_func.cpp:123-130  conv_space(geno, n, exp) */
void orc_conv_space_exp(double *g, int n) {
  for (int k = 0; k < n; k++) {
    g[k] = exp(g[k]);
    if (g[k] == -INFINITY) g[k] = -ORC_INF;
  }
}

/* gen_func.cpp:862-868  miss_data: all three genotype values (nearly) equal */
int orc_miss_data(const double *g) {
  if (ORC_ABS(g[0] - g[1]) < ORC_EPSILON && ORC_ABS(g[1] - g[2]) < ORC_EPSILON) return 1;
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * gen_func.cpp:974-1009  est_maf(n_ind, pdg, indF = NULL, ignore_miss_data)
 * num/den are NOT reset between passes of the do-while (:976-977 sit outside the loop); with
 * indF == NULL the posterior does not depend on freq, so the loop ends after its second pass.
 * Restated as the loop, not as the closed form, to keep the rounding identical.
 * ---------------------------------------------------------------------------------------------- */
double orc_est_maf(uint64_t n_ind, const double *pdg, int ignore_miss_data) {
  int iters = 0;
  double num = 0, den = 0;
  double F, prev_freq, freq = 0.01;
  double pp[ORC_N_GENO];

  do {
    prev_freq = freq;
    for (uint64_t i = 0; i < n_ind; i++) {
      const double *g = pdg + 3 * i;
      if (orc_miss_data(g) && ignore_miss_data) continue; /* :985, evaluated on LOG-space values */
      F = 0;
      orc_post_prob(pp, g, ORC_N_GENO);                   /* :989 */
      orc_conv_space_exp(pp, ORC_N_GENO);                 /* :997 */
      num += pp[1] + pp[2] * (2 - F);                     /* :999 */
      den += 2 * pp[1] + (pp[0] + pp[2]) * (2 - F);       /* :1000 */
    }
    freq = num / den;
  } while (ORC_ABS(prev_freq - freq) > ORC_EPSILON && iters++ < 100);
  return freq;
}

/* ------------------------------------------------------------------------------------------------
 * gen_func.cpp:1073-1119  pair_freq_iter: one EM step on the 4 haplotype frequencies.
 * haplotype index k: bit 1 = allele at site 1, bit 0 = allele at site 2.
 * *err is set to 1 where the reference calls error() (:1115-1116).
 * ---------------------------------------------------------------------------------------------- */
#define ORC_G1(h, k) ((h >> 1 & 1) + (k >> 1 & 1))
#define ORC_G2(h, k) ((h & 1) + (k & 1))

uint64_t orc_pair_freq_iter(double f[4], const double *s1, const double *s2, uint64_t n, int ignore_miss_data,
                            int *err) {
  double ff[4];
  int k, h;
  uint64_t x = 0;
  memset(ff, 0, 4 * sizeof(double));

  for (uint64_t i = 0; i < n; ++i) {
    const double *p0 = s1 + 3 * i, *p1 = s2 + 3 * i;
    double sum, tmp;
    if ((orc_miss_data(p0) || orc_miss_data(p1)) && ignore_miss_data) continue; /* :1089 */
    x++;
    sum = 0;
    for (k = 0; k < 4; ++k)
      for (h = 0; h < 4; ++h) sum += f[k] * f[h] * p0[ORC_G1(k, h)] * p1[ORC_G2(k, h)]; /* :1093-1096 */
    for (k = 0; k < 4; ++k) {
      tmp = 0;
      for (h = 0; h < 4; ++h)
        tmp += f[k] * f[h] *
               (p0[ORC_G1(h, k)] * p1[ORC_G2(h, k)] + p0[ORC_G1(k, h)] * p1[ORC_G2(k, h)]); /* :1100-1101 */
      ff[k] += tmp / sum;                                                                    /* :1103 */
    }
  }
  for (k = 0; k < 4; ++k) f[k] = ff[k] / (2 * x);            /* :1108-1109, 2*x is uint64 */
  for (k = 0; k < 4; k++) f[k] /= f[0] + f[1] + f[2] + f[3]; /* :1112-1113, SEQUENTIAL normalise */
  if (!((ignore_miss_data && x <= n) || (!ignore_miss_data && x == n))) {
    if (err) *err = 1; /* :1115-1116 "invalid number of individuals!" */
  }
  return x;
}

/* gen_func.cpp:1027-1059  haplo_freq (log_scale = false branch, the only one ngsLD.cpp:294 uses).
 * returns n_iter: index of the iteration that converged (0-based), ITER_MAX if none did. */
uint64_t orc_haplo_freq(double hap_freq[4], uint64_t *n, const double *gl1, const double *gl2, double maf1,
                        double maf2, uint64_t n_ind, int ignore_miss_data, int *err) {
  double last[4];
  if (maf1 < 0 || maf1 > 1 || maf2 < 0 || maf2 > 1) { /* :1030-1031 "invalid allele frequencies" */
    if (err) *err = 2;
    hap_freq[0] = hap_freq[1] = hap_freq[2] = hap_freq[3] = NAN;
    *n = 0;
    return 0;
  }
  hap_freq[0] = (1 - maf1) * (1 - maf2); /* :1034-1037 */
  hap_freq[1] = (1 - maf1) * maf2;
  hap_freq[2] = maf1 * (1 - maf2);
  hap_freq[3] = maf1 * maf2;

  uint64_t n_iter;
  for (n_iter = 0; n_iter < ORC_ITER_MAX; n_iter++) {
    double eps = 0;
    memcpy(last, hap_freq, 4 * sizeof(double));
    *n = orc_pair_freq_iter(hap_freq, gl1, gl2, n_ind, ignore_miss_data, err);
    for (uint64_t j = 0; j < 4; j++) {
      double x = fabs(hap_freq[j] - last[j]);
      if (x > eps) eps = x; /* NaN never raises eps: a NaN step "converges" (:1049-1055) */
    }
    if (eps < ORC_EPSILON) break;
  }
  return n_iter;
}

/* ------------------------------------------------------------------------------------------------
 * ngsLD.cpp:365-367 pearson_r -> gsl_stats_correlation(x,1,y,1,n), squared with pow(.,2).
 * GSL is NOT in /root/reference and not installed (README.md:20 "gsl v1.15 tested", no pin).
 * Restated from GSL's published statistics/covar_source.c: one-pass mean/co-moment recurrence with
 * long double accumulators, double sqrt.  PARITY UNPINNED at this boundary (no GSL to run); any
 * stable formula agrees to ~1e-15, the GPU path is held to 1e-9.
 * ---------------------------------------------------------------------------------------------- */
double orc_correlation(const double *x, const double *y, uint64_t n) {
  long double sum_xsq = 0.0, sum_ysq = 0.0, sum_cross = 0.0;
  long double ratio, delta_x, delta_y, mean_x, mean_y, r;
  mean_x = x[0];
  mean_y = y[0];
  for (uint64_t i = 1; i < n; ++i) {
    ratio = i / (i + 1.0);
    delta_x = x[i] - mean_x;
    delta_y = y[i] - mean_y;
    sum_xsq += delta_x * delta_x * ratio;
    sum_ysq += delta_y * delta_y * ratio;
    sum_cross += delta_x * delta_y * ratio;
    mean_x += delta_x / (i + 1.0);
    mean_y += delta_y / (i + 1.0);
  }
  r = sum_cross / (sqrt((double)sum_xsq) * sqrt((double)sum_ysq));
  return (double)r;
}

double orc_pearson_r2(const double *x, const double *y, uint64_t n) { return pow(orc_correlation(x, y, n), 2); }

/* ngsLD.cpp:296-306 (maf, D, D', r2) and :328-333 (chi2 in FLOAT arithmetic) */
void orc_pair_stats(const double hap[4], double *D, double *Dp, double *r2, double hap_maf[2], float *chi2) {
  double maf[2];
  maf[0] = 1 - (hap[0] + hap[1]);
  maf[1] = 1 - (hap[0] + hap[2]);
  *D = hap[0] * hap[3] - hap[1] * hap[2];
  *Dp = *D / (*D < 0 ? -ORC_MIN(maf[0] * maf[1], (1 - maf[0]) * (1 - maf[1]))
                     : ORC_MIN(maf[0] * (1 - maf[1]), (1 - maf[0]) * maf[1]));
  *r2 = pow(*D / sqrt(maf[0] * maf[1] * (1 - maf[0]) * (1 - maf[1])), 2);
  hap_maf[0] = maf[0];
  hap_maf[1] = maf[1];

  float c = 0;
  float freq_A = hap[0] + hap[1];
  float freq_B = hap[0] + hap[2];
  float exp_hap_freq[4] = {freq_A * freq_B, freq_A * (1 - freq_B), (1 - freq_A) * freq_B,
                           (1 - freq_A) * (1 - freq_B)};
  for (int i = 0; i < 4; i++) c += pow(hap[i] - exp_hap_freq[i], 2) / exp_hap_freq[i]; /* double expr, += to float */
  *chi2 = c;
}

/* ------------------------------------------------------------------------------------------------
 * read_data.cpp:28-47 (binary branch of read_geno) + :106-116 (EOF check).
 * Disk order is [site][ind][3] little-endian doubles; the reference stores [ind][site][3] and
 * transposes later (ngsLD.cpp:88) -- here the site-major layout is written directly.
 * ---------------------------------------------------------------------------------------------- */
static int normalise_triple(double *g, int log_scale) {
  if (!log_scale) orc_conv_space_log(g, ORC_N_GENO); /* :37-38 */
  orc_post_prob(g, g, ORC_N_GENO);                   /* :40 */
  if (isnan(g[0]) || isnan(g[1]) || isnan(g[2])) return -1; /* :42-45 */
  return 0;
}

int orc_normalise_raw(const double *raw, int log_scale, uint64_t n_ind, uint64_t n_sites, double *out) {
  uint64_t n = n_sites * n_ind;
  for (uint64_t k = 0; k < n; k++) {
    double *g = out + 3 * k;
    g[0] = raw[3 * k];
    g[1] = raw[3 * k + 1];
    g[2] = raw[3 * k + 2];
    if (normalise_triple(g, log_scale)) return -3;
  }
  return 0;
}

int orc_read_geno_bin(const char *path, int log_scale, uint64_t n_ind, uint64_t n_sites, double *out, char *errbuf,
                      size_t errlen) {
  gzFile fh = (strcmp(path, "-") == 0) ? gzdopen(fileno(stdin), "rb") : gzopen(path, "rb");
  if (fh == NULL) {
    snprintf(errbuf, errlen, "cannot open GENO file!");
    return -1;
  }
  gzbuffer(fh, 1 << 20);
  for (uint64_t s = 0; s < n_sites; s++)
    for (uint64_t i = 0; i < n_ind; i++) {
      double *g = out + (s * n_ind + i) * 3;
      if (gzread(fh, g, 3 * sizeof(double)) != (int)(3 * sizeof(double))) {
        snprintf(errbuf, errlen, "%s",
                 gzeof(fh) ? "GENO file at premature EOF. Check GENO file and number of sites!"
                           : "cannot read binary GENO file. Check GENO file and number of sites!");
        gzclose(fh);
        return -2;
      }
      if (normalise_triple(g, log_scale)) {
        snprintf(errbuf, errlen, "NaN found! Is the file format correct?");
        gzclose(fh);
        return -3;
      }
    }
  char c;
  gzread(fh, &c, 1); /* :107-109 */
  if (!gzeof(fh)) {
    snprintf(errbuf, errlen, "GENO file not at EOF. Check GENO file and number of sites!");
    gzclose(fh);
    return -4;
  }
  gzclose(fh);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * read_data.cpp:48-104 (text branch of read_geno).  Fields are split on ' ' and TAB; only tokens that
 * strtod consumes completely count as fields (split(char*, sep, double**), gen_func.cpp:381-410), so
 * chromosome names / alleles vanish and the LAST n_ind*n_geno numeric fields are the data (:80-81).
 * A first line with fewer numeric fields is a header (:64-72).  No -inf -> -INF replacement and no NaN
 * check in this branch (unlike the binary one).
 * Deviations: lines of any length are accepted (the reference cuts at BUFF_LEN = 500000 bytes); an empty
 * line is an error (the reference silently leaves that site uninitialised, :58-59).
 * ---------------------------------------------------------------------------------------------- */
static uint64_t split_doubles(char *line, double **out, uint64_t *cap) {
  uint64_t n = 0;
  char *p = line;
  while (*p) {
    size_t len = strcspn(p, " \t");
    char save = p[len];
    p[len] = '\0';
    if (len > 0) {
      char *end;
      double v = strtod(p, &end);
      if (*end == '\0') {
        if (n == *cap) {
          *cap = *cap ? *cap * 2 : 4096;
          *out = (double *)realloc(*out, *cap * sizeof(double));
        }
        (*out)[n++] = v;
      }
    }
    p[len] = save;
    p += len;
    if (*p) p++;
  }
  return n;
}

int orc_read_geno_text(const char *path, int in_probs, int log_scale, uint64_t n_ind, uint64_t n_sites, double *out,
                       char *errbuf, size_t errlen) {
  gzFile fh = (strcmp(path, "-") == 0) ? gzdopen(fileno(stdin), "r") : gzopen(path, "r");
  if (fh == NULL) {
    snprintf(errbuf, errlen, "cannot open GENO file!");
    return -1;
  }
  gzbuffer(fh, 1 << 20);
  const uint64_t n_geno = in_probs ? ORC_N_GENO : 1;
  size_t lcap = 1 << 16, llen;
  char *line = (char *)malloc(lcap);
  double *t = NULL;
  uint64_t tcap = 0;
  int rc = 0, empty_seen = 0;
  for (uint64_t s = 0; s < n_sites && rc == 0; s++) {
    llen = 0;
    for (;;) { /* one line of any length */
      if (gzgets(fh, line + llen, (int)(lcap - llen)) == NULL) break;
      llen += strlen(line + llen);
      if (llen > 0 && line[llen - 1] == '\n') break;
      if (llen + 1 < lcap) break; /* EOF without newline */
      lcap *= 2;
      line = (char *)realloc(line, lcap);
    }
    if (llen == 0) {
      snprintf(errbuf, errlen, "%s", gzeof(fh) ? "GENO file at premature EOF. Check GENO file and number of sites!"
                                               : "cannot read GZip GENO file. Check GENO file and number of sites!");
      rc = -2;
      break;
    }
    if (line[llen - 1] == '\n' || line[llen - 1] == '\r') line[--llen] = '\0'; /* chomp */
    if (llen == 0) { /* :58-59 -- the for loop's s++ runs: the empty line takes this site's place and leaves it unfilled */
      empty_seen = 1;
      continue;
    }
    uint64_t n_fields = split_doubles(line, &t, &tcap);
    if (!n_fields || (s == 0 && n_fields < n_ind * n_geno)) { /* header, :64-72 */
      s--;
      continue;
    }
    if (n_fields < n_ind * n_geno) {
      snprintf(errbuf, errlen, "wrong GENO file format. Less fields than expected!");
      rc = -6;
      break;
    }
    const double *ptr = t + (n_fields - n_ind * n_geno);
    for (uint64_t i = 0; i < n_ind; i++) {
      double *g = out + (s * n_ind + i) * 3;
      g[0] = g[1] = g[2] = -ORC_INF; /* init_ptr(..., -INF), read_data.cpp:21 */
      if (in_probs) {
        for (int k = 0; k < 3; k++) g[k] = log_scale ? ptr[i * 3 + k] : log(ptr[i * 3 + k]); /* :85-86 */
      } else {
        int gg = (int)ptr[i];
        if (gg >= 0) {
          if (gg > 2) {
            snprintf(errbuf, errlen, "wrong GENO file format. Genotypes must be coded as {-1,0,1,2} !");
            rc = -7;
            break;
          }
          g[gg] = log(1);
        } else {
          g[0] = g[1] = g[2] = log((double)1 / ORC_N_GENO);
        }
      }
      orc_post_prob(g, g, ORC_N_GENO); /* :98 */
    }
  }
  if (rc == 0) {
    char c;
    gzread(fh, &c, 1); /* :107-109 */
    if (!gzeof(fh)) {
      snprintf(errbuf, errlen, "GENO file not at EOF. Check GENO file and number of sites!");
      rc = -4;
    }
  }
  if (rc == 0 && empty_seen) { /* where this restatement stops following: the reference goes on with that site uninitialised */
    snprintf(errbuf, errlen, "empty line in GENO file");
    rc = -5;
  }
  gzclose(fh);
  free(line);
  free(t);
  return rc;
}

/* gen_func.cpp:886-914 call_geno(geno, 3, log_scale = true, N_thresh, call_thresh, miss_data = 0), as called
 * from ngsLD.cpp:97 (in_logscale is true once read_geno has returned, read_data.cpp:114).
 * array_max_pos / array_min_pos (gen_func.cpp:73-98) return the FIRST extreme. */
void orc_call_geno(double *geno, double N_thresh, double call_thresh) {
  int max_pos = 0, min_pos = 0;
  double mx = -INFINITY, mn = +INFINITY;
  for (int k = 0; k < 3; k++) {
    if (geno[k] > mx) {
      max_pos = k;
      mx = geno[k];
    }
  }
  for (int k = 0; k < 3; k++) {
    if (geno[k] < mn) {
      min_pos = k;
      mn = geno[k];
    }
  }
  double max_pp = exp(geno[max_pos]);
  if (geno[min_pos] == geno[max_pos]) max_pp = -1; /* missing data, mode 0 */
  if (max_pp < N_thresh)
    for (int g = 0; g < 3; g++) geno[g] = log((double)1 / 3);
  if (max_pp >= call_thresh) {
    for (int g = 0; g < 3; g++) geno[g] = -ORC_INF;
    geno[max_pos] = log(1);
  }
}

void orc_call_geno_all(orc_params *p, double N_thresh, double call_thresh) { /* ngsLD.cpp:92-98 */
  for (uint64_t k = 0; k < p->n_sites * p->n_ind; k++) orc_call_geno(p->geno_lkl + 3 * k, N_thresh, call_thresh);
}

/* ngsLD.cpp:103-114: est_maf on the log GLs, then exp() in place and expected genotype p1 + 2*p2 */
void orc_preprocess(orc_params *p) {
  for (uint64_t s = 0; s < p->n_sites; s++)
    p->maf[s] = orc_est_maf(p->n_ind, p->geno_lkl + s * p->n_ind * 3, p->ignore_miss_data);
  for (uint64_t s = 0; s < p->n_sites; s++)
    for (uint64_t i = 0; i < p->n_ind; i++) {
      double *g = p->geno_lkl + (s * p->n_ind + i) * 3;
      orc_conv_space_exp(g, ORC_N_GENO);
      p->expected_geno[s * p->n_ind + i] = g[1] + 2 * g[2];
    }
}

/* ------------------------------------------------------------------------------------------------
 * Position file.  gen_func.cpp:238-282 read_file (skip empty and '#' lines, skip `offset` header
 * lines, a last line without '\n' is lost to the gzeof() test at :253), read_data.cpp:165-218
 * read_dist (fields split on TAB only; first site's gap is pos-0; INFINITY on chromosome change;
 * gap < 1 fatal; prev_pos parsed with strtoul base 0), ngsLD.cpp:124-132 labels (first TAB -> ':').
 * Deviation: a line whose 2nd field parses to 0 makes the reference spin forever
 * (read_data.cpp:188-195); here it is an error.
 * ---------------------------------------------------------------------------------------------- */
static int read_lines(const char *path, uint64_t offset, char ***out, uint64_t *n_out) {
  gzFile fh = (strcmp(path, "-") == 0) ? gzdopen(fileno(stdin), "r") : gzopen(path, "r");
  if (fh == NULL) return -1;
  gzbuffer(fh, 1 << 16);
  char *buf = (char *)malloc(ORC_BUFF_LEN);
  char **lines = NULL;
  uint64_t cnt = 0, cap = 0;
  for (;;) {
    buf[0] = '\0';
    gzgets(fh, buf, ORC_BUFF_LEN);
    if (gzeof(fh)) break;
    size_t L = strlen(buf);
    if (L > 0 && (buf[L - 1] == '\n' || buf[L - 1] == '\r')) buf[L - 1] = '\0'; /* chomp, gen_func.cpp:190-197 */
    if (strlen(buf) == 0 || buf[0] == '#') continue;
    if (offset > 0) {
      offset--;
      continue;
    }
    if (cnt == cap) {
      cap = cap ? cap * 2 : 1024;
      lines = (char **)realloc(lines, cap * sizeof(char *));
    }
    lines[cnt++] = strdup(buf);
  }
  free(buf);
  gzclose(fh);
  *out = lines;
  *n_out = cnt;
  return 0;
}

/* number of TAB-separated fields the way split(char*, sep, char***) counts them (gen_func.cpp:305-327,
   :413-426): empty fields are kept, a trailing TAB yields one more empty field. */
static uint64_t count_tab_fields(const char *s) {
  uint64_t n = 1;
  for (; *s; s++)
    if (*s == '\t') n++;
  return n;
}

int orc_read_pos(orc_params *p, char *errbuf, size_t errlen) {
  char **lines = NULL;
  uint64_t n_rows = 0;
  if (read_lines(p->in_pos, p->in_pos_header ? 1 : 0, &lines, &n_rows)) {
    snprintf(errbuf, errlen, "cannot open file!");
    return -1;
  }
  int rc = 0;
  if (n_rows == 0) { /* read_file leaves its array NULL; read_split takes that for a file it could not open (read_data.cpp:134-136) */
    snprintf(errbuf, errlen, "cannot open file!");
    return -1;
  }
  uint64_t n_fields = 0;
  for (uint64_t i = 0; rc == 0 && i < n_rows; i++) { /* read_split, all of the lines, before read_dist counts them (read_data.cpp:139-147) */
    uint64_t nf = count_tab_fields(lines[i]);
    if (n_fields == 0) n_fields = nf;
    if (nf != n_fields) {
      snprintf(errbuf, errlen, "invalid number of fields in file!");
      rc = -3;
    }
  }
  if (rc == 0 && n_rows != p->n_sites) { /* read_data.cpp:178-179 */
    snprintf(errbuf, errlen, "wrong number of lines in POS file!");
    rc = -2;
  }
  if (rc == 0 && n_fields < 2) {
    snprintf(errbuf, errlen, "wrong POS file format!");
    rc = -4;
  }
  if (rc) {
    for (uint64_t i = 0; i < n_rows; i++) free(lines[i]);
    free(lines);
    return rc;
  }

  p->pos_dist = (double *)malloc(p->n_sites * sizeof(double));
  p->labels = lines;
  char *prev_chr = NULL;
  uint64_t prev_pos = 0;
  for (uint64_t s = 0; s < p->n_sites; s++) {
    char *line = lines[s];
    char *t1 = strchr(line, '\t');
    size_t chr_len = (size_t)(t1 - line);
    char *f1 = t1 + 1;
    char *t2 = strchr(f1, '\t');
    char save = 0;
    if (t2) {
      save = *t2;
      *t2 = '\0';
    }
    double posd = strtod(f1, NULL);
    unsigned long posu = strtoul(f1, NULL, 0);
    if (t2) *t2 = save;
    if (posd == 0) {
      snprintf(errbuf, errlen, "header line in POS file (use --posH)");
      rc = -5;
      break;
    }
    /* read_data.cpp:199-200: an EMPTY stored name (none yet, or a line whose first field was empty) takes this line's */
    if (prev_chr != NULL && prev_chr[0] == '\0') {
      free(prev_chr);
      prev_chr = NULL;
    }
    int same = prev_chr == NULL || (strlen(prev_chr) == chr_len && strncmp(prev_chr, line, chr_len) == 0);
    if (prev_chr == NULL) prev_chr = strndup(line, chr_len);
    if (same) {
      p->pos_dist[s] = posd - prev_pos;
      if (p->pos_dist[s] < 1) {
        snprintf(errbuf, errlen, "invalid distance between adjacent sites!");
        rc = -6;
        break;
      }
    } else {
      p->pos_dist[s] = INFINITY;
      free(prev_chr);
      prev_chr = strndup(line, chr_len);
    }
    prev_pos = posu;
  }
  free(prev_chr);
  if (rc) {
    orc_free_pos(p);
    return rc;
  }
  for (uint64_t s = 0; s < p->n_sites; s++) { /* ngsLD.cpp:128-132 */
    char *t = strchr(lines[s], '\t');
    if (t) *t = ':';
  }
  return 0;
}

void orc_free_pos(orc_params *p) {
  if (p->labels) {
    for (uint64_t s = 0; s < p->n_sites; s++) free(p->labels[s]);
    free(p->labels);
    p->labels = NULL;
  }
  free(p->pos_dist);
  p->pos_dist = NULL;
}

/* ------------------------------------------------------------------------------------------------
 * gsl_rng_taus.  GSL is an external dependency of the reference (README.md:20) and is not installed
 * here; this restates the published algorithm of GSL's rng/taus.c ("taus", not "taus2": no minimum
 * values are forced on the state).  State s1,s2,s3 are 32-bit values;
 *   TAUSWORTHE(s,a,b,c,d) = (((s & c) << d) & MASK) ^ ((((s << a) & MASK) ^ s) >> b)
 *   s1 = T(s1,13,19,4294967294,12); s2 = T(s2,2,25,4294967288,4); s3 = T(s3,3,11,4294967280,17); out = s1^s2^s3
 * seeding: s == 0 -> 1; s1 = LCG(s), s2 = LCG(s1), s3 = LCG(s2) with LCG(n) = (69069*n) & 0xffffffff, then six
 * warm-up draws; uniform = get() / 4294967296.0.
 * ---------------------------------------------------------------------------------------------- */
#define ORC_TAUS(s, a, b, c, d) (((((s) & (c)) << (d)) & 0xffffffffUL) ^ (((((s) << (a)) & 0xffffffffUL) ^ (s)) >> (b)))

uint32_t orc_taus_get(orc_taus *r) {
  unsigned long s1 = r->s1, s2 = r->s2, s3 = r->s3;
  s1 = ORC_TAUS(s1, 13, 19, 4294967294UL, 12);
  s2 = ORC_TAUS(s2, 2, 25, 4294967288UL, 4);
  s3 = ORC_TAUS(s3, 3, 11, 4294967280UL, 17);
  r->s1 = (uint32_t)s1;
  r->s2 = (uint32_t)s2;
  r->s3 = (uint32_t)s3;
  return (uint32_t)(s1 ^ s2 ^ s3);
}

void orc_taus_set(orc_taus *r, unsigned long s) {
  if (s == 0) s = 1;
#define ORC_LCG(n) ((69069UL * (n)) & 0xffffffffUL)
  unsigned long a = ORC_LCG(s), b = ORC_LCG(a), c = ORC_LCG(b);
  r->s1 = (uint32_t)a;
  r->s2 = (uint32_t)b;
  r->s3 = (uint32_t)c;
  for (int k = 0; k < 6; k++) orc_taus_get(r);
}

double orc_taus_uniform(orc_taus *r) { return orc_taus_get(r) / 4294967296.0; }

/* ngsLD.cpp:69-70,165-166: master stream seeded with --seed; row s1's generator is seeded with
 * (unsigned long) draw_rnd(master, 0, INF) = (unsigned long)(0 + uniform * (1e15 - 0)), drawn for s1 = 0,1,2,... */
void orc_row_seeds(uint64_t seed, uint64_t n_sites, uint64_t *out) {
  orc_taus m;
  orc_taus_set(&m, (unsigned long)seed);
  for (uint64_t s = 0; s < n_sites; s++) {
    const uint64_t mn = 0, mx = (uint64_t)ORC_INF;
    out[s] = (uint64_t)(unsigned long)(mn + orc_taus_uniform(&m) * (mx - mn));
  }
}

/* ------------------------------------------------------------------------------------------------
 * ngsLD.cpp:229-359 calc_pair_LD: the walk over s2 for one s1, filters in the reference's order:
 * dist (:252) -> snp dist (:258) -> maf[s1] (:264, break) -> maf[s2] (:270, skip) -> random
 * sub-sampling (:277: one uniform draw from the row's own generator per pair that got this far; the
 * pair is kept iff draw <= rnd_sample).  At the default --rnd_sample 1 nothing is ever rejected and
 * the draws have no effect, so they are skipped.
 * ---------------------------------------------------------------------------------------------- */
static uint64_t orc_row_seed_of(const orc_params *p, uint64_t s1) {
  /* row seeds are a serial stream over s1; recomputing the prefix is O(s1) -- fine for an oracle */
  orc_taus m;
  orc_taus_set(&m, (unsigned long)p->seed);
  uint64_t v = 0;
  for (uint64_t s = 0; s <= s1; s++) v = (uint64_t)(unsigned long)(0 + orc_taus_uniform(&m) * ((uint64_t)ORC_INF - 0));
  return v;
}
uint64_t orc_row(const orc_params *p, uint64_t s1, orc_pair *out, uint64_t cap, int *err) {
  uint64_t s2 = s1 + 1, n_out = 0;
  double dist = 0;
  const int sampling = p->rnd_sample > 0 && p->rnd_sample < 1;
  orc_taus rng;
  if (sampling) orc_taus_set(&rng, (unsigned long)orc_row_seed_of(p, s1));
  while (s2 < p->n_sites) {
    dist += p->pos_dist[s2];
    if (p->max_kb_dist > 0 && p->max_kb_dist * 1000 < dist) break;
    if (p->max_snp_dist > 0 && p->max_snp_dist < s2 - s1) break;
    if (p->maf[s1] < p->min_maf) break;
    if (p->maf[s2] < p->min_maf) {
      s2++;
      continue;
    }
    if (sampling && orc_taus_uniform(&rng) > p->rnd_sample) { /* ngsLD.cpp:277-282 */
      s2++;
      continue;
    }
    if (out != NULL && n_out < cap) {
      orc_pair *r = out + n_out;
      r->s1 = s1;
      r->s2 = s2;
      r->dist = dist;
      r->r2pear =
          orc_pearson_r2(p->expected_geno + s1 * p->n_ind, p->expected_geno + s2 * p->n_ind, p->n_ind); /* :290 */
      r->n_iter = orc_haplo_freq(r->hap, &r->n_ind_data, p->geno_lkl + s1 * p->n_ind * 3,
                                 p->geno_lkl + s2 * p->n_ind * 3, p->maf[s1], p->maf[s2], p->n_ind,
                                 p->ignore_miss_data, err); /* :294 */
      orc_pair_stats(r->hap, &r->D, &r->Dp, &r->r2, r->hap_maf, &r->chi2);
    }
    n_out++;
    s2++;
  }
  return n_out;
}

uint64_t orc_row_end(const orc_params *p, uint64_t s1) {
  uint64_t s2 = s1 + 1;
  double dist = 0;
  while (s2 < p->n_sites) {
    dist += p->pos_dist[s2];
    if (p->max_kb_dist > 0 && p->max_kb_dist * 1000 < dist) break;
    if (p->max_snp_dist > 0 && p->max_snp_dist < s2 - s1) break;
    if (p->maf[s1] < p->min_maf) break;
    s2++;
  }
  return s2;
}

typedef struct {
  const orc_params *p;
  uint64_t s1_begin, s1_end;
  int tid, n_threads;
  uint64_t *row_off; /* [rows+1] when filling, NULL when counting */
  uint64_t *row_cnt;
  orc_pair *out;
  int err;
} orc_job;

static void *orc_worker(void *arg) {
  orc_job *j = (orc_job *)arg;
  for (uint64_t s1 = j->s1_begin + (uint64_t)j->tid; s1 < j->s1_end; s1 += (uint64_t)j->n_threads) {
    uint64_t r = s1 - j->s1_begin;
    if (j->row_off == NULL)
      j->row_cnt[r] = orc_row(j->p, s1, NULL, 0, &j->err);
    else
      orc_row(j->p, s1, j->out + j->row_off[r], j->row_off[r + 1] - j->row_off[r], &j->err);
  }
  return NULL;
}

/* ngsLD.cpp:153-198: one task per s1; here rows are striped over n_threads pthreads (results do not
   depend on scheduling; the reference's output order is arbitrary for --n_threads > 1). */
uint64_t orc_run(const orc_params *p, uint64_t s1_begin, uint64_t s1_end, orc_pair *out, uint64_t cap, int *err) {
  uint64_t rows = s1_end > s1_begin ? s1_end - s1_begin : 0;
  int nt = p->n_threads > 0 ? p->n_threads : 1;
  uint64_t *cnt = (uint64_t *)calloc(rows + 1, sizeof(uint64_t));
  uint64_t *off = (uint64_t *)calloc(rows + 1, sizeof(uint64_t));
  pthread_t *th = (pthread_t *)malloc((size_t)nt * sizeof(pthread_t));
  orc_job *jobs = (orc_job *)calloc((size_t)nt, sizeof(orc_job));
  uint64_t total = 0;

  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      if (out == NULL) break;
      for (uint64_t r = 0; r < rows; r++) off[r + 1] = off[r] + cnt[r];
      if (off[rows] > cap) break;
    }
    for (int t = 0; t < nt; t++) {
      jobs[t].p = p;
      jobs[t].s1_begin = s1_begin;
      jobs[t].s1_end = s1_end;
      jobs[t].tid = t;
      jobs[t].n_threads = nt;
      jobs[t].row_off = pass ? off : NULL;
      jobs[t].row_cnt = cnt;
      jobs[t].out = out;
      pthread_create(&th[t], NULL, orc_worker, &jobs[t]);
    }
    for (int t = 0; t < nt; t++) {
      pthread_join(th[t], NULL);
      if (jobs[t].err && err) *err = jobs[t].err;
    }
    if (pass == 0)
      for (uint64_t r = 0; r < rows; r++) total += cnt[r];
  }
  free(cnt);
  free(off);
  free(th);
  free(jobs);
  return total;
}

typedef struct {
  const orc_params *p;
  uint64_t s1_begin, s1_end;
  int tid, n_threads;
  uint64_t pairs, iters;
  double sum;
  int pearson_only; /* orc_bench_pearson: the walk and pearson_r alone (ngsLD.cpp:290, 365-367) */
} orc_bjob;

static void *orc_bench_worker(void *arg) {
  orc_bjob *j = (orc_bjob *)arg;
  orc_pair r[64];
  int err = 0;
  for (uint64_t s1 = j->s1_begin + (uint64_t)j->tid; s1 < j->s1_end; s1 += (uint64_t)j->n_threads) {
    /* same work as orc_row, a row at a time in chunks of 64 records */
    orc_params q = *j->p;
    uint64_t end = orc_row_end(&q, s1);
    for (uint64_t b = s1 + 1; b < end; b += 64) {
      uint64_t e = b + 64 < end ? b + 64 : end;
      for (uint64_t s2 = b; s2 < e; s2++) {
        if (q.maf[s2] < q.min_maf) continue;
        orc_pair *o = &r[s2 - b];
        o->r2pear = orc_pearson_r2(q.expected_geno + s1 * q.n_ind, q.expected_geno + s2 * q.n_ind, q.n_ind);
        if (j->pearson_only) {
          if (isfinite(o->r2pear)) j->sum += o->r2pear;
          j->pairs++;
          continue;
        }
        o->n_iter = orc_haplo_freq(o->hap, &o->n_ind_data, q.geno_lkl + s1 * q.n_ind * 3, q.geno_lkl + s2 * q.n_ind * 3,
                                   q.maf[s1], q.maf[s2], q.n_ind, q.ignore_miss_data, &err);
        orc_pair_stats(o->hap, &o->D, &o->Dp, &o->r2, o->hap_maf, &o->chi2);
        if (isfinite(o->r2)) j->sum += o->r2;
        j->iters += o->n_iter < ORC_ITER_MAX ? o->n_iter + 1 : ORC_ITER_MAX;
        j->pairs++;
      }
    }
  }
  return NULL;
}

static uint64_t orc_bench_mode(const orc_params *p, uint64_t s1_begin, uint64_t s1_end, double *checksum, uint64_t *iters,
                               int pearson_only);
uint64_t orc_bench(const orc_params *p, uint64_t s1_begin, uint64_t s1_end, double *checksum, uint64_t *iters) {
  return orc_bench_mode(p, s1_begin, s1_end, checksum, iters, 0);
}
/* the same rows, pearson_r alone: what the reference's program compiled without GSL (oracle/_ref: ref_main) leaves out */
uint64_t orc_bench_pearson(const orc_params *p, uint64_t s1_begin, uint64_t s1_end, double *checksum) {
  return orc_bench_mode(p, s1_begin, s1_end, checksum, NULL, 1);
}
static uint64_t orc_bench_mode(const orc_params *p, uint64_t s1_begin, uint64_t s1_end, double *checksum, uint64_t *iters,
                               int pearson_only) {
  int nt = p->n_threads > 0 ? p->n_threads : 1;
  pthread_t *th = (pthread_t *)malloc((size_t)nt * sizeof(pthread_t));
  orc_bjob *jobs = (orc_bjob *)calloc((size_t)nt, sizeof(orc_bjob));
  for (int t = 0; t < nt; t++) {
    jobs[t].p = p;
    jobs[t].s1_begin = s1_begin;
    jobs[t].s1_end = s1_end;
    jobs[t].tid = t;
    jobs[t].n_threads = nt;
    jobs[t].pearson_only = pearson_only;
    pthread_create(&th[t], NULL, orc_bench_worker, &jobs[t]);
  }
  uint64_t pairs = 0, it = 0;
  double sum = 0;
  for (int t = 0; t < nt; t++) {
    pthread_join(th[t], NULL);
    pairs += jobs[t].pairs;
    it += jobs[t].iters;
    sum += jobs[t].sum;
  }
  if (checksum) *checksum = sum;
  if (iters) *iters = it;
  free(th);
  free(jobs);
  return pairs;
}

/* ngsLD.cpp:77 header, :314-351 rows */
void orc_print_header(FILE *fh, int extend_out) {
  fprintf(fh, "site1\tsite2\tdist\tr2_ExpG\tD\tDp\tr2%s\n",
          extend_out ? "\tsample_size\tmaf1\tmaf2\thap00\thap01\thap10\thap11\thap_maf1\thap_maf2\tchi2\tloglike\tnIter"
                     : "");
}

/* the same two printers into memory (tests: byte comparison with the reference's own fprintf lines, oracle/_ref) */
long orc_format_header(char *buf, size_t cap, int extend_out) {
  FILE *fh = fmemopen(buf, cap, "w");
  if (fh == NULL) return -1;
  orc_print_header(fh, extend_out);
  long n = ftell(fh);
  fclose(fh);
  return n;
}

void orc_print_pair(FILE *fh, const orc_params *p, const orc_pair *r);
long orc_format_pair(char *buf, size_t cap, const orc_params *p, const orc_pair *r) {
  FILE *fh = fmemopen(buf, cap, "w");
  if (fh == NULL) return -1;
  orc_print_pair(fh, p, r);
  long n = ftell(fh);
  fclose(fh);
  return n;
}

void orc_print_pair(FILE *fh, const orc_params *p, const orc_pair *r) {
  /* labels == NULL reproduces glibc's "(null)" for the reference's NULL labels without --pos
     (ngsLD.cpp:135, gen_func.cpp:729-731) */
  const char *l1 = p->labels ? p->labels[r->s1] : "(null)";
  const char *l2 = p->labels ? p->labels[r->s2] : "(null)";
  fprintf(fh, "%s\t%s\t%.0f\t%f\t%f\t%f\t%f", l1, l2, r->dist, r->r2pear, r->D, r->Dp, r->r2);
  if (p->extend_out)
    fprintf(fh, "\t%lu\t%f\t%f\t%f\t%f\t%f\t%f\t%f\t%f\t%f\t%f\t%lu", (unsigned long)r->n_ind_data, p->maf[r->s1],
            p->maf[r->s2], r->hap[0], r->hap[1], r->hap[2], r->hap[3], r->hap_maf[0], r->hap_maf[1], r->chi2, 0.0,
            (unsigned long)r->n_iter);
  fprintf(fh, "\n");
}
