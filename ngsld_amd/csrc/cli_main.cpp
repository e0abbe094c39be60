// cli_main.cpp -- the `ngsLD` command line of the MI355X-native engine: same flags, defaults, validation
// messages, stderr chatter and TSV as the reference binary (parse_args.cpp:6-184, ngsLD.cpp:27-223), with
// the thread-pool section (ngsLD.cpp:153-198) replaced by the C-ABI in include/ngsld.h.
//
// Differences a user can see (all listed in DESIGN.md):
//   * rows are written in (site1, site2) order (the reference's order is arbitrary for --n_threads > 1);
//   * --n_threads sets the number of host threads that format the TSV; --device N (new) picks the GPU, --max_gpu_mem GB (new)
//     caps the device memory: a windowed run on binary input that does not fit is streamed slab by slab;
//   * --devices 0-7 (or 0,2,5; new) runs the job on several GPUs of the node from this one process (ngsld_run_multi):
//     the rows are cut into one part per device, the output is the single-device output;
#include <getopt.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <string>
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/ngsld.h"
#include "../../include/ngsld_host.h"
#include "knobs.h"
#include "host_buf.h"

namespace {

const char *kVersion = "1.2.1-mi355x";

struct Params {  // ngsLD.hpp:11-44
  char *in_geno = nullptr;
  bool in_bin = false, in_probs = false, in_logscale = false;
  uint64_t n_ind = 0, n_sites = 0;
  char *in_pos = nullptr;
  bool in_pos_header = false;
  uint64_t max_kb_dist = 100, max_snp_dist = 0;
  double min_maf = 0;
  bool ignore_miss_data = false, call_geno = false;
  double N_thresh = 0, call_thresh = 0, rnd_sample = 1;
  uint64_t seed = 0;
  bool extend_out = false;
  char *out = nullptr;
  FILE *out_fh = stdout;
  unsigned n_threads = 1, verbose = 1;
  int device = 0;
  double max_gpu_mem = 0;  // GB of device memory the run may use; 0 = what is free on the device
  std::vector<int> devices;  // --devices: more than one entry = one part of the rows per device
  bool keep_parts = false;   // --keep_parts: with --devices and --out, leave parts 1.. in <out>.part<k> instead of appending them
};

// "0-3", "0,2,5", "1" -> device indices; empty on a malformed list
std::vector<int> parse_devices(const char *txt) {
  std::vector<int> out;
  const char *p = txt;
  while (*p) {
    char *end = nullptr;
    const long a = strtol(p, &end, 10);
    if (end == p || a < 0) return {};
    long b = a;
    p = end;
    if (*p == '-') {
      b = strtol(p + 1, &end, 10);
      if (end == p + 1 || b < a) return {};
      p = end;
    }
    for (long d = a; d <= b; ++d) out.push_back((int)d);
    if (*p == ',') ++p;
    else if (*p) return {};
  }
  return out;
}

[[noreturn]] void error(const char *func, const char *msg) {  // gen_func.cpp:12-18
  fflush(stdout);
  fprintf(stderr, "\n=====\nERROR: [%s] %s\n=====\n\n", func, msg);
  perror("\t");
  fflush(stderr);
  exit(-1);
}

// The function the reference names in front of an error of its positions reader: the file is opened by read_file
// (gen_func.cpp:244-246), its fields are counted by read_split (read_data.cpp:134-147), everything else is read_dist's own.
const char *pos_error_function(const char *msg, const char *path) {
  // (a file that opens but holds no usable line is "cannot open file!" too -- from read_split, read_data.cpp:134-136)
  if (std::strcmp(msg, "cannot open file!") == 0) return path != nullptr && access(path, R_OK) == 0 ? "read_split" : "read_file";
  if (std::strcmp(msg, "invalid number of fields in file!") == 0) return "read_split";
  return "read_dist";
}

void parse_cmd_args(Params *pars, int argc, char **argv) {
  static struct option long_options[] = {{"geno", required_argument, NULL, 'g'},
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
                                         {"device", required_argument, NULL, 1001},
                                         {"max_gpu_mem", required_argument, NULL, 1002},
                                         {"devices", required_argument, NULL, 1003},
                                         {"keep_parts", no_argument, NULL, 1004},
                                         {0, 0, 0, 0}};
  pars->seed = (uint64_t)(time(NULL) + rand() % 1000);  // parse_args.cpp:23
  int c = 0;
  while ((c = getopt_long_only(argc, argv, "g:pln:s:Z:d:D:f:mcN:C:r:S:xo:t:V:", long_options, NULL)) != -1)
    switch (c) {
      case 'g': pars->in_geno = optarg; break;
      case 'p': pars->in_probs = true; break;
      case 'l': pars->in_logscale = true; pars->in_probs = true; break;
      case 'n': pars->n_ind = (uint64_t)atoi(optarg); break;
      case 's': pars->n_sites = (uint64_t)atoi(optarg); break;
      case 'a': pars->in_pos = optarg; pars->in_pos_header = false; break;
      case 'A': pars->in_pos = optarg; pars->in_pos_header = true; break;
      case 'd': pars->max_kb_dist = (uint64_t)atoi(optarg); break;
      case 'D': pars->max_snp_dist = (uint64_t)atoi(optarg); break;
      case 'f': pars->min_maf = atof(optarg); break;
      case 'm': pars->ignore_miss_data = true; break;
      case 'c': pars->call_geno = true; break;
      case 'N': pars->N_thresh = atof(optarg); pars->call_geno = true; break;
      case 'C': pars->call_thresh = atof(optarg); pars->call_geno = true; break;
      case 'r': pars->rnd_sample = atof(optarg); break;
      case 'S': pars->seed = (uint64_t)atoi(optarg); break;
      case 'x': pars->extend_out = true; break;
      case 'o': pars->out = optarg; break;
      case 't': pars->n_threads = (unsigned)atoi(optarg); break;
      case 'V': pars->verbose = (unsigned)atoi(optarg); break;
      case 1001: pars->device = atoi(optarg); break;
      case 1002: pars->max_gpu_mem = atof(optarg); break;
      case 1004: pars->keep_parts = true; break;
      case 1003:
        pars->devices = parse_devices(optarg);
        if (pars->devices.empty()) error(__FUNCTION__, "--devices takes a list like 0-7 or 0,2,5");
        pars->device = pars->devices[0];
        break;
      default: exit(-1);  // unknown flags and --outH (declared, no case: parse_args.cpp:55,130)
    }

  if (pars->verbose >= 1) {  // parse_args.cpp:135-159
    fprintf(stderr, "==> Input Arguments:\n");
    fprintf(stderr,
            "\tgeno: %s\n\tprobs: %s\n\tlog_scale: %s\n\tn_ind: %lu\n\tn_sites: %lu\n\tpos: %s (%s header)\n\t"
            "max_kb_dist (kb): %lu\n\tmax_snp_dist: %lu\n\tmin_maf: %f\n\tignore_miss_data: %s\n\tcall_geno: %s\n\t"
            "N_thresh: %f\n\tcall_thresh: %f\n\trnd_sample: %f\n\tseed: %lu\n\textend_out: %s\n\tout: %s\n\t"
            "n_threads: %d\n\tverbose: %d\n\tversion: %s (%s @ %s)\n\n",
            pars->in_geno, pars->in_probs ? "true" : "false", pars->in_logscale ? "true" : "false",
            (unsigned long)pars->n_ind, (unsigned long)pars->n_sites, pars->in_pos,
            pars->in_pos_header ? "WITH" : "WITHOUT", (unsigned long)pars->max_kb_dist,
            (unsigned long)pars->max_snp_dist, pars->min_maf, pars->ignore_miss_data ? "true" : "false",
            pars->call_geno ? "true" : "false", pars->N_thresh, pars->call_thresh, pars->rnd_sample,
            (unsigned long)pars->seed, pars->extend_out ? "true" : "false", pars->out, pars->n_threads,
            pars->verbose, kVersion, __DATE__, __TIME__);
  }
  if (pars->verbose > 4)
    fprintf(stderr,
            "==> Verbose values greater than 4 for debugging purpose only. Expect large amounts of info on screen\n");

  // parse_args.cpp:168-183
  if (pars->in_geno == NULL) error(__FUNCTION__, "genotype input file (--geno) missing!");
  if (pars->n_ind == 0) error(__FUNCTION__, "number of individuals (--n_ind) missing!");
  if (pars->n_sites == 0) error(__FUNCTION__, "number of sites (--n_sites) missing!");
  if (pars->in_pos == NULL && pars->max_kb_dist > 0)
    error(__FUNCTION__, "position file necessary in order to filter by maximum distance!");
  if (pars->min_maf < 0 || pars->min_maf > 1) error(__FUNCTION__, "minimum allele frequency must be in [0,1]!");
  if (pars->call_geno && !pars->in_probs)
    error(__FUNCTION__, "can only call genotypes from likelihoods/probabilities!");
  if (pars->rnd_sample <= 0 || pars->rnd_sample > 1)
    error(__FUNCTION__, "proportion of comparisons to sample must be in ]0,1]!");
  if (pars->n_threads < 1) error(__FUNCTION__, "number of threads cannot be less than 1!");
  // (n_threads is unsigned in the reference too, ngsLD.hpp:33: a negative value passes the test above as 4 billion and
  // its thread pool then fails to start; here it would only ask for that many formatter threads)
  if (pars->n_threads > 4096) pars->n_threads = 4096;
  // --keep_parts (new here): <out>, <out>.part1, ... in this order are the table.  Part 0 of a compressed output would be a gzip
  // member and the others plain text -- their concatenation neither a gzip stream nor a table -- so the two do not combine;
  // and without --out or with a single device there are no part files to keep: said, not silently ignored
  if (pars->keep_parts) {
    const char *dot = pars->out ? strrchr(pars->out, '.') : NULL;
    if (dot != NULL && strcmp(dot, ".gz") == 0) error(__FUNCTION__, "--keep_parts cannot be combined with a compressed output (--out *.gz)!");
    if (pars->out == NULL || pars->devices.size() < 2)
      fprintf(stderr, "WARN: --keep_parts has no effect without --out and --devices with two or more devices\n");
  }
}

ngsld_gz *g_gz = nullptr;     // --out *.gz: the compressor behind pars.out_fh
FILE *g_gz_fh = nullptr;      // ... and the stream over the compressor's pipe (owns the pipe's write descriptor)
// Ends the compressed file: the stream first (what stdio still buffers goes into the pipe, and the pipe's write end is
// closed exactly once, by its owner), then the compressor.  Registered with atexit, so that error() exits end the file
// too -- atexit handlers run BEFORE stdio flushes its streams, hence the explicit fclose here.
void finish_gz() {
  if (g_gz_fh != nullptr) {
    FILE *fh = g_gz_fh;
    g_gz_fh = nullptr;
    fclose(fh);
  }
  if (g_gz != nullptr) {
    ngsld_gz *g = g_gz;
    g_gz = nullptr;
    if (ngsld_host_gz_close(g) != NGSLD_OK) fprintf(stderr, "\nERROR: the compressed output is incomplete\n");
  }
}
// fclose(pars.out_fh) for every path out of main
void close_output(FILE *fh) {
  if (fh != nullptr && fh == g_gz_fh)
    finish_gz();
  else if (fh != nullptr)
    fclose(fh);
}

struct SinkState {
  const Params *pars;
  const ngsld_pos *pos;    // may be NULL (no --pos)
  const double *pos_dist;  // NULL => all INFINITY
  const std::vector<double> *maf;
};

// One batch, in (s1, s2) order: --n_threads threads format (the reference's threads each fprintf under a mutex,
// ngsLD.cpp:310-352), one ordered write.
double g_sink_seconds = 0.0;  // NGSLD_TIMING=1: time spent formatting + writing, reported at exit
uint64_t g_sink_batches = 0;

int write_batch(void *user, const ngsld_batch *b) {
  SinkState *st = static_cast<SinkState *>(user);
  const auto t0 = std::chrono::steady_clock::now();
  fflush(st->pars->out_fh);
  int rc = 0;
  if (b->text != nullptr) {  // rows formatted on the device (ngsld_set_text_output): only bytes to write
    const char *q = b->text;
    uint64_t left = b->text_len;
    const int fd = fileno(st->pars->out_fh);
    while (left && rc == 0) {
      const ssize_t w = ::write(fd, q, left > (1ull << 30) ? (size_t)(1ull << 30) : (size_t)left);
      if (w <= 0) rc = 1;
      else { q += w; left -= (uint64_t)w; }
    }
  } else {
    rc = ngsld_host_write_batch(b, st->pos, st->pos_dist, st->maf->data(), (int)st->pars->n_threads,
                                fileno(st->pars->out_fh)) == NGSLD_OK ? 0 : 1;
  }
  g_sink_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  ++g_sink_batches;
  return rc;
}

struct TimingReport {  // NGSLD_TIMING=1: wall time since start, per phase, and the sink's share, on stderr
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
  bool on = getenv("NGSLD_TIMING") && strcmp(getenv("NGSLD_TIMING"), "1") == 0;
  void mark(const char *what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[timing] %-28s %.3f s\n", what, std::chrono::duration<double>(now - last).count());
    last = now;
  }
  bool printed = false;
  void print() {
    if (on && !printed)
      fprintf(stderr, "[timing] total %.3f s, TSV formatting + write %.3f s in %lu batches\n",
              std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), g_sink_seconds,
              (unsigned long)g_sink_batches);
    printed = true;
  }
  ~TimingReport() { print(); }
};


// (the genotype matrix in host memory: host_buf.h -- malloc semantics, huge pages)
double *alloc_matrix(size_t bytes) { return ngsld::alloc_host_matrix(bytes / sizeof(double)); }

struct ReadState {
  const Params *pars;
  char err[512];
};

int read_slab(void *user, uint64_t site_begin, uint64_t n_sites, double *dst) {
  ReadState *r = static_cast<ReadState *>(user);
  return ngsld_host_read_geno_bin_range(r->pars->in_geno, r->pars->n_ind, site_begin, n_sites, dst, r->err,
                                        sizeof(r->err)) == NGSLD_OK ? 0 : 1;
}

void fill_run_params(const Params &pars, ngsld_params *lp, ngsld_geno_opts *go) {
  lp->max_kb_dist = pars.max_kb_dist;
  lp->max_snp_dist = pars.max_snp_dist;
  lp->min_maf = pars.min_maf;
  lp->ignore_miss_data = pars.ignore_miss_data ? 1 : 0;
  lp->extend_out = pars.extend_out ? 1 : 0;
  lp->rnd_sample = pars.rnd_sample;
  lp->seed = pars.seed;
  lp->first_row = 0;
  memset(go, 0, sizeof(*go));
  go->log_scale = pars.in_logscale ? 1 : 0;
  go->ignore_miss_data = pars.ignore_miss_data ? 1 : 0;
  go->call_geno = pars.call_geno ? 1 : 0;
  go->N_thresh = pars.N_thresh;
  go->call_thresh = pars.call_thresh;
}

// The out-of-core path (no counterpart in the reference): the same TSV, the matrix read slab by slab.
// Returns false (nothing written yet) when may_fall_back and the slabs cannot hold a window.
// NGSLD_HOST_TEXT=1: every row through the host formatter (the fallback of the device-side TSV; A/B, tests)
static bool host_text_only() {
  static const bool on = getenv("NGSLD_HOST_TEXT") && strcmp(getenv("NGSLD_HOST_TEXT"), "1") == 0;
  return on;
}

bool run_streamed(Params &pars, uint64_t slab_sites, bool may_fall_back) {
  char err[512];
  ngsld_pos *pos = nullptr;
  if (pars.verbose >= 1) fprintf(stderr, "==> Getting sites coordinates\n");
  if (pars.in_pos &&
      ngsld_host_read_pos(pars.in_pos, pars.in_pos_header ? 1 : 0, pars.n_sites, &pos, err, sizeof(err)) != NGSLD_OK)
    error(pos_error_function(err, pars.in_pos), err);
  ngsld_params lp;
  ngsld_geno_opts go;
  fill_run_params(pars, &lp, &go);
  std::vector<double> maf(pars.n_sites);
  SinkState sink;
  sink.pars = &pars;
  sink.pos = pos;
  sink.pos_dist = pos ? ngsld_host_pos_dist(pos) : nullptr;
  sink.maf = &maf;
  ReadState rs;
  rs.pars = &pars;
  rs.err[0] = 0;
  uint64_t n_pairs = 0, n_slabs = 0;
  if (may_fall_back) {
    std::vector<ngsld_slab> probe(pars.n_sites);
    uint64_t n = 0;
    if (ngsld_plan_slabs(sink.pos_dist, pars.n_sites, &lp, slab_sites, probe.data(), probe.size(), &n) != NGSLD_OK) {
      ngsld_host_free_pos(pos);
      return false;
    }
  }
  if (pars.verbose >= 1)
    fprintf(stderr, "==> Streaming the genotype matrix in slabs of up to %lu sites\n", (unsigned long)slab_sites);
  const bool dev_text = !host_text_only();
  std::vector<const char *> lab;
  if (pos && dev_text) {
    lab.resize(pars.n_sites);
    for (uint64_t s = 0; s < pars.n_sites; s++) lab[s] = ngsld_host_label(pos, s);
  }
  const int rc = ngsld_run_streamed_text(pars.device, pars.n_sites, pars.n_ind, sink.pos_dist, &lp, &go, slab_sites,
                                         read_slab, &rs, maf.data(), write_batch, &sink, &n_pairs, &n_slabs, err,
                                         sizeof(err), pos && dev_text ? lab.data() : nullptr, dev_text ? 1 : 0);
  if (rc == NGSLD_ERR_NAN) error("read_geno", err);
  if (rc == NGSLD_ERR_MAF_RANGE) error("haplo_freq", err);
  if (rc != NGSLD_OK) error("ngsld_run_streamed", rs.err[0] ? rs.err : err);
  if (pars.verbose >= 2) {  // (as the resident run's line, + where: a large host share means no room for the slabs' exact stores)
    ngsld_replay_stats_t st;
    if (ngsld_streamed_replay_info(&st) == NGSLD_OK)
      fprintf(stderr, "==> %lu of %lu pairs replayed in the reference's operation order (%lu on the device, %lu on host threads)\n",
              (unsigned long)st.pairs_replayed, (unsigned long)n_pairs, (unsigned long)st.pairs_on_device, (unsigned long)st.pairs_on_host);
  }
  if (pars.verbose >= 1) fprintf(stderr, "==> Freeing memory...\n");
  close_output(pars.out_fh);
  ngsld_host_free_pos(pos);
  if (pars.verbose >= 1) fprintf(stderr, "Done!\n");
  return true;
}

// ---- several devices, one process (ngsld_run_multi): part 0 writes straight to the output, the other parts spool their
// rows to <out>.part<k> (a temporary file when the output is stdout) and are appended in order at the end ----
struct MultiSink {
  const Params *pars;
  const ngsld_pos *pos;
  const double *pos_dist;
  const double *maf;
  std::vector<FILE *> fh;  // one per part
  int n_threads;           // formatter threads per part
};

int write_part_batch(void *user, int part, const ngsld_batch *b) {
  MultiSink *m = static_cast<MultiSink *>(user);
  FILE *fh = m->fh[(size_t)part];
  fflush(fh);
  const int fd = fileno(fh);
  if (b->text != nullptr) {
    const char *q = b->text;
    uint64_t left = b->text_len;
    while (left) {
      const ssize_t w = ::write(fd, q, left > (1ull << 30) ? (size_t)(1ull << 30) : (size_t)left);
      if (w <= 0) return 1;
      q += w;
      left -= (uint64_t)w;
    }
    return 0;
  }
  return ngsld_host_write_batch(b, m->pos, m->pos_dist, m->maf, m->n_threads, fd) == NGSLD_OK ? 0 : 1;
}

void run_multi(Params &pars, const double *raw, int text_semantics, int log_scale) {
  char err[512];
  const int n = (int)pars.devices.size();
  ngsld_pos *pos = nullptr;
  if (pars.verbose >= 1) fprintf(stderr, "==> Getting sites coordinates\n");
  if (pars.in_pos &&
      ngsld_host_read_pos(pars.in_pos, pars.in_pos_header ? 1 : 0, pars.n_sites, &pos, err, sizeof(err)) != NGSLD_OK)
    error(pos_error_function(err, pars.in_pos), err);
  ngsld_params lp;
  ngsld_geno_opts go;
  fill_run_params(pars, &lp, &go);
  go.text_semantics = text_semantics;
  go.log_scale = log_scale;
  std::vector<double> maf(pars.n_sites);
  MultiSink ms;
  ms.pars = &pars;
  ms.pos = pos;
  ms.pos_dist = pos ? ngsld_host_pos_dist(pos) : nullptr;
  ms.maf = maf.data();
  ms.n_threads = (int)std::max<unsigned>(1u, pars.n_threads / (unsigned)n);
  std::vector<std::string> part_names((size_t)n);
  ms.fh.assign((size_t)n, nullptr);
  ms.fh[0] = pars.out_fh;
  for (int k = 1; k < n; ++k) {
    if (pars.out != NULL) {
      part_names[(size_t)k] = std::string(pars.out) + ".part" + std::to_string(k);
      ms.fh[(size_t)k] = fopen(part_names[(size_t)k].c_str(), "w+");
    } else {
      ms.fh[(size_t)k] = tmpfile();
    }
    if (ms.fh[(size_t)k] == nullptr) error(__FUNCTION__, "cannot open a part file next to the output");
  }
  ReadState rs;
  rs.pars = &pars;
  rs.err[0] = 0;
  const bool dev_text = !host_text_only();
  std::vector<const char *> lab;
  if (pos && dev_text) {
    lab.resize(pars.n_sites);
    for (uint64_t s = 0; s < pars.n_sites; s++) lab[s] = ngsld_host_label(pos, s);
  }
  if (pars.verbose >= 1) fprintf(stderr, "==> Launching %d device threads...\n", n);
  std::vector<uint64_t> per((size_t)n, 0);
  const int rc = ngsld_run_multi(pars.devices.data(), n, pars.n_sites, pars.n_ind, ms.pos_dist, &lp, &go, raw,
                                 raw ? nullptr : read_slab, raw ? nullptr : &rs, maf.data(), write_part_batch, &ms,
                                 pos && dev_text ? lab.data() : nullptr, dev_text ? 1 : 0, per.data(), err, sizeof(err));
  if (rc == NGSLD_ERR_NAN) error("read_geno", err);
  if (rc == NGSLD_ERR_MAF_RANGE) error("haplo_freq", err);
  if (rc != NGSLD_OK) error("ngsld_run_multi", rs.err[0] ? rs.err : err);
  if (pars.verbose >= 2)
    for (int k = 0; k < n; ++k)
      fprintf(stderr, "\tdevice %d: %lu pairs\n", pars.devices[(size_t)k], (unsigned long)per[(size_t)k]);
  // append the spooled parts in order -- inside the kernel where the file systems allow it (copy_file_range: no trip of the
  // bytes through this process), through a buffer otherwise.  --keep_parts (with --out): the parts stay where they are,
  // <out> + <out>.part1 + ... in this order is the table (a shard per device is as good as a merged file, and at
  // 10^9 rows the merge is a second pass over seven eighths of the output)
  fflush(pars.out_fh);
  std::vector<char> buf;
  for (int k = 1; k < n; ++k) {
    FILE *fh = ms.fh[(size_t)k];
    fflush(fh);
    if (pars.keep_parts && !part_names[(size_t)k].empty()) {
      if (fclose(fh) != 0) error(__FUNCTION__, "cannot finish a part file");
      continue;
    }
    const off_t size = lseek(fileno(fh), 0, SEEK_END);
    off_t done = 0;
    while (size > 0 && done < size) {  // (the output's own position is at its end: it has only ever been appended to)
      off_t in_off = done;
      const ssize_t w = copy_file_range(fileno(fh), &in_off, fileno(pars.out_fh), nullptr, (size_t)std::min<off_t>(size - done, (off_t)1 << 30), 0);
      if (w <= 0) break;  // (a pipe, /dev/null, another file system on an old kernel: the buffered copy below takes over)
      done += w;
    }
    if (done < size) {
      if (buf.empty()) buf.resize(8u << 20);
      if (fseeko(fh, done, SEEK_SET) != 0) error(__FUNCTION__, "cannot re-read a part file");
      size_t got;
      while ((got = fread(buf.data(), 1, buf.size(), fh)) > 0)
        if (fwrite(buf.data(), 1, got, pars.out_fh) != got) error(__FUNCTION__, "cannot append a part to the output");
      fflush(pars.out_fh);
    }
    fclose(fh);
    if (!part_names[(size_t)k].empty()) unlink(part_names[(size_t)k].c_str());
  }
  if (pars.verbose >= 1) fprintf(stderr, "==> Freeing memory...\n");
  close_output(pars.out_fh);
  ngsld_host_free_pos(pos);
  if (pars.verbose >= 1) fprintf(stderr, "Done!\n");
}

}  // namespace

int main(int argc, char **argv) {
  TimingReport timing_report;
  // This program's pinned host buffers are registered anonymous memory rather than hipHostMalloc memory (engine.hip, PinBuf):
  // 4x cheaper to get and to give back -- 0.07 s of a one-second run of configs[2], 3 s of configs[4] at full size.  The
  // library leaves that off by default (in a long-lived process that forks children it cost a GPU memory fault before the
  // mapping became fork-proof); this process is short-lived, single-purpose and never forks.  NGSLD_PIN_REGISTER=0 turns it off.
  setenv("NGSLD_PIN_REGISTER", "1", /*overwrite=*/0);
  Params pars;
  parse_cmd_args(&pars, argc, argv);

  // ---- check input files (ngsLD.cpp:41-57) ----
  struct stat st;
  if (stat(pars.in_geno, &st) != 0) error(__FUNCTION__, "cannot check GENO file size!");
  const char *dot = strrchr(pars.in_geno, '.');  // the reference dereferences NULL for a name without '.' (ngsLD.cpp:45)
  if (dot != NULL && strcmp(dot, ".gz") == 0) {
    if (pars.verbose >= 1) fprintf(stderr, "==> GZIP input file (not BINARY)\n");
    pars.in_bin = false;
  } else {
    if (pars.verbose >= 1) fprintf(stderr, "==> BINARY input file (always lkl)\n");
    pars.in_bin = true;
    pars.in_probs = true;
    if (!ngsld_host_geno_size_ok((uint64_t)st.st_size, pars.n_ind, pars.n_sites))
      error(__FUNCTION__, "invalid/corrupt genotype input file!");
  }
  // ---- prepare output (ngsLD.cpp:73-77): the header is always written ----
  // (new: an --out name ending in .gz is written gzip-compressed, --n_threads deflate workers; the reference has no
  // compressed output)
  const char *odot = pars.out ? strrchr(pars.out, '.') : NULL;
  if (odot != NULL && strcmp(odot, ".gz") == 0) {
    int wfd = -1;
    if (ngsld_host_gz_open(pars.out, (int)pars.n_threads, &g_gz, &wfd) != NGSLD_OK)
      error(__FUNCTION__, "cannot open output file!");
    pars.out_fh = g_gz_fh = fdopen(wfd, "w");
    atexit(finish_gz);  // every path out of main (the streamed and the multi-device runs return early) ends the file
  } else if (pars.out != NULL) {
    pars.out_fh = fopen(pars.out, "w");
  }
  if (pars.out_fh == NULL) error(__FUNCTION__, "cannot open output file!");
  char hdr[512];
  const size_t hn = ngsld_host_format_header(hdr, sizeof(hdr), pars.extend_out);
  fwrite(hdr, 1, hn, pars.out_fh);
  fflush(pars.out_fh);

  ngsld_host_set_threads((int)pars.n_threads);
  timing_report.mark("arguments, output header");

  // A binary matrix that will plainly run resident (below the 4 GiB from which a windowed run is pipelined in slabs, and
  // well inside any --max_gpu_mem) is read -- and the positions after it -- by a thread of its own while the device context
  // comes up: the two take 0.15-0.35 s each and used to run one after the other.  Errors wait for the join and come out
  // where and as they always did.
  struct FreeDeleter {
    void operator()(double *q) const { free(q); }
  };
  struct EarlyRead {
    std::thread th;
    bool started = false, pos_done = false;
    int geno_rc = NGSLD_OK, pos_rc = NGSLD_OK;
    char geno_err[512] = "", pos_err[512] = "";
    std::unique_ptr<double, FreeDeleter> raw;
    ngsld_pos *pos = nullptr;
  } early;
  const uint64_t geno_bytes = (uint64_t)pars.n_sites * pars.n_ind * 3 * sizeof(double);
  if (pars.in_bin && geno_bytes < (4ull << 30) && ngsld::test_knob("SLAB_SITES") == nullptr &&
      (pars.max_gpu_mem <= 0 || pars.max_gpu_mem * 1e9 > 3.0 * (double)geno_bytes)) {
    early.raw.reset(alloc_matrix((size_t)geno_bytes));
    if (early.raw) {
      early.started = true;
      early.th = std::thread([&early, &pars]() {
        early.geno_rc = ngsld_host_read_geno_bin(pars.in_geno, pars.n_ind, pars.n_sites, early.raw.get(), early.geno_err,
                                                 sizeof(early.geno_err));
        if (early.geno_rc == NGSLD_OK && pars.in_pos) {
          early.pos_rc = ngsld_host_read_pos(pars.in_pos, pars.in_pos_header ? 1 : 0, pars.n_sites, &early.pos,
                                             early.pos_err, sizeof(early.pos_err));
          early.pos_done = true;
        }
      });
    }
  }
  auto join_early = [&early]() {
    if (early.th.joinable()) early.th.join();
  };

  // ---- several devices: a windowed run on binary input lets every part read its own slab from the file ----
  if (pars.devices.size() > 1 && pars.in_bin && (pars.max_kb_dist > 0 || pars.max_snp_dist > 0)) {
    join_early();
    early.raw.reset();  // (every part reads its own slab)
    ngsld_host_free_pos(early.pos);
    if (pars.verbose >= 1) fprintf(stderr, "> Reading data from file...\n");
    run_multi(pars, nullptr, 0, pars.in_logscale ? 1 : 0);
    return 0;
  }

  // ---- device ----
  ngsld_ctx *ctx = nullptr;
  if (ngsld_create(pars.device, &ctx) != NGSLD_OK) {
    join_early();
    error("ngsld_create", ngsld_last_error(nullptr));
  }

  timing_report.mark("ngsld_create");
  char err[512];
  // ---- does the matrix fit the device?  If not, a windowed run on binary input is streamed slab by slab ----
  uint64_t budget = 0;
  {
    uint64_t free_b = 0, total_b = 0;
    if (ngsld_device_memory(pars.device, &free_b, &total_b) != NGSLD_OK)
      error("ngsld_device_memory", "cannot query the device memory");
    budget = (uint64_t)(0.9 * (double)free_b);
    if (pars.max_gpu_mem > 0 && pars.max_gpu_mem * 1e9 < (double)budget) budget = (uint64_t)(pars.max_gpu_mem * 1e9);
    // (--max_gpu_mem holds for what the library allocates by itself too: the exact store of the device-side replay)
    if (pars.max_gpu_mem > 0) (void)ngsld_set_memory_budget(pars.device, budget);
  }
  // resident = one context holding every site; the budget helpers price two, so ask with twice the budget.  `fits`: with room
  // for what the device-side replay of un-called input builds beside the planes -- the exact store and, for the lane-per-pair
  // replay kernel, its individual-major copy; this program's text batches of up to 512 individuals are the wavefront-per-pair
  // kernel's, which reads the store's own layout: the matrix twice, three times beyond.  `fits_bare`: the planes alone -- what a
  // run needs; without room for the store its flagged pairs are replayed on host threads (the same bytes, the reference's speed).
  const int copies = pars.n_ind > 512 ? 3 : 2;
  const bool fits = ngsld_sites_for_budget(pars.n_ind, 2 * budget, copies) >= pars.n_sites;
  const bool fits_bare = fits || ngsld_sites_for_budget(pars.n_ind, 2 * budget, 1) >= pars.n_sites;
  const bool streamable = pars.in_bin && (pars.max_kb_dist > 0 || pars.max_snp_dist > 0);
  uint64_t slab_sites = 0;  // > 0: run slab by slab
  if (const char *e = ngsld::test_knob("SLAB_SITES")) {  // tests: stream a small file in slabs of n sites
    if (pars.in_bin) slab_sites = strtoull(e, nullptr, 10);
  } else if (!fits) {
    // A windowed run on binary input that fits only without the store is streamed: every slab then has room for its own store.
    // Where slabs cannot be had (text input, no window, a budget below two contexts) the matrix stays resident if its planes
    // fit, without room for the store.
    if (streamable) slab_sites = ngsld_sites_for_budget(pars.n_ind, budget, copies);
    if (slab_sites < 2) slab_sites = 0;
    if (slab_sites == 0 && !fits_bare) {
      if (!pars.in_bin)
        error(__FUNCTION__, "the genotype matrix does not fit the device memory budget (only binary input is streamed)");
      slab_sites = ngsld_sites_for_budget(pars.n_ind, budget, 1);  // (slabs without room for the store before none at all)
      if (slab_sites < 2) error(__FUNCTION__, "the device memory budget is too small for this number of individuals");
    }
  } else if (pars.in_bin && (pars.max_kb_dist > 0 || pars.max_snp_dist > 0) && (uint64_t)st.st_size >= (4ull << 30) &&
             !(getenv("NGSLD_PIPELINE") && strcmp(getenv("NGSLD_PIPELINE"), "0") == 0)) {
    // a large windowed job that fits is still cut into about six slabs, only to overlap the file read and the
    // upload of one part with the pair kernels of the previous one (same output; falls back when a window is too wide).
    // From 4 GiB: a 1.2 GB file ran 0.5 s faster resident (2.2 s) than in slabs, a 5.8 GB one 0.5 s slower.
    slab_sites = std::min<uint64_t>(ngsld_sites_for_budget(pars.n_ind, budget, copies), (pars.n_sites + 5) / 6);
  }
  if (slab_sites > 0) {
    join_early();  // (an early read is only started for matrices far below these thresholds: normally nothing to wait for)
    early.raw.reset();
    ngsld_host_free_pos(early.pos);
    early.pos = nullptr;
    early.started = early.pos_done = false;
    ngsld_destroy(ctx);
    ctx = nullptr;
    if (run_streamed(pars, slab_sites, /*may_fall_back=*/fits_bare)) return 0;
    if (ngsld_create(pars.device, &ctx) != NGSLD_OK) error("ngsld_create", ngsld_last_error(nullptr));
  }

  // the pinned host buffers the text batches will land in: allocated by a library thread while the matrix is read, uploaded
  // and prepped (pinning ~0.7 GB costs ~0.1 s, which the first batches of the run used to wait for).  Only now that the run is
  // known to be resident: a streamed run uses contexts of its own, and destroying this one had to wait for the pinning to
  // finish only to undo it
  if (!host_text_only())
    (void)ngsld_reserve_text_buffers(ctx, pars.extend_out ? 190 : 95);

  // ---- read input data (ngsLD.cpp:85-114; the arithmetic runs on the device) ----
  if (pars.verbose >= 1) fprintf(stderr, "> Reading data from file...\n");
  join_early();
  // (malloc: 1.2 GB of zero-filling a std::vector before overwriting it was a third of the read time)
  std::unique_ptr<double, FreeDeleter> raw;
  if (early.started)
    raw = std::move(early.raw);
  else
    raw.reset(alloc_matrix((size_t)pars.n_sites * pars.n_ind * 3 * sizeof(double)));
  if (!raw) error(__FUNCTION__, "cannot allocate the genotype matrix");
  ngsld_geno_opts go;
  memset(&go, 0, sizeof(go));
  go.log_scale = pars.in_logscale ? 1 : 0;
  go.ignore_miss_data = pars.ignore_miss_data ? 1 : 0;
  go.call_geno = pars.call_geno ? 1 : 0;
  go.N_thresh = pars.N_thresh;
  go.call_thresh = pars.call_thresh;
  if (early.started) {
    if (early.geno_rc != NGSLD_OK) error("read_geno", early.geno_err);
  } else if (pars.in_bin) {
    if (ngsld_host_read_geno_bin(pars.in_geno, pars.n_ind, pars.n_sites, raw.get(), err, sizeof(err)) != NGSLD_OK)
      error("read_geno", err);
  } else {
    int is_log = 0;
    if (ngsld_host_read_geno_text(pars.in_geno, pars.in_probs ? 1 : 0, pars.in_logscale ? 1 : 0, pars.n_ind,
                                  pars.n_sites, raw.get(), &is_log, err, sizeof(err)) != NGSLD_OK)
      error("read_geno", err);
    go.log_scale = is_log;
    go.text_semantics = 1;
  }
  timing_report.mark("read genotype file");
  if (pars.devices.size() > 1) {  // all pairs (or text input) on several devices: the matrix is in host memory once
    ngsld_destroy(ctx);
    if (pars.call_geno && pars.verbose >= 1) fprintf(stderr, "> Calling genotypes...\n");
    if (pars.verbose >= 1) fprintf(stderr, "==> Calculating MAF for all sites...\n");
    run_multi(pars, raw.get(), go.text_semantics, go.log_scale);
    return 0;
  }
  if (pars.call_geno && pars.verbose >= 1) fprintf(stderr, "> Calling genotypes...\n");
  if (pars.verbose >= 1) fprintf(stderr, "==> Calculating MAF for all sites...\n");
  int rc = ngsld_set_geno_raw_opts(ctx, raw.get(), pars.n_sites, pars.n_ind, &go);
  if (rc == NGSLD_ERR_NAN) error("read_geno", ngsld_last_error(ctx));
  if (rc == NGSLD_ERR_INVALID && pars.call_geno) error("call_geno", ngsld_last_error(ctx));
  if (rc != NGSLD_OK) error("ngsld_set_geno_raw", ngsld_last_error(ctx));
  timing_report.mark("upload + per-site prep");
  // The matrix stays in host memory for the run (the reference holds it throughout, ngsLD.cpp:86-89): the pairs whose
  // outcome the reference's own rounding decides are replayed from it in the reference's operation order.
  struct RawSource {
    const double *raw;
    uint64_t n_sites, n_ind;
  } raw_src{raw.get(), pars.n_sites, pars.n_ind};
  auto read_raw = [](void *user, uint64_t site_begin, uint64_t n, double *dst) -> int {
    const RawSource *r = static_cast<const RawSource *>(user);
    if (site_begin + n > r->n_sites) return 1;
    memcpy(dst, r->raw + site_begin * r->n_ind * 3, n * r->n_ind * 3 * sizeof(double));
    return 0;
  };
  if (ngsld::test_knob_is("REPLAY_SOURCE", "0"))
    raw.reset();  // (tests: replay from the device's own planes)
  else if (ngsld::test_knob_is("REPLAY_SOURCE", "callback")) {  // (tests: the callback form)
    if (ngsld_set_replay_source(ctx, read_raw, &raw_src) != NGSLD_OK) error("ngsld_set_replay_source", ngsld_last_error(ctx));
  } else if (ngsld_set_replay_matrix(ctx, raw.get()) != NGSLD_OK)
    error("ngsld_set_replay_matrix", ngsld_last_error(ctx));

  if (pars.verbose >= 1) fprintf(stderr, "==> Getting sites coordinates\n");
  ngsld_pos *pos = nullptr;
  if (pars.in_pos) {
    if (early.pos_done) {
      if (early.pos_rc != NGSLD_OK) error(pos_error_function(early.pos_err, pars.in_pos), early.pos_err);
      pos = early.pos;
    } else if (ngsld_host_read_pos(pars.in_pos, pars.in_pos_header ? 1 : 0, pars.n_sites, &pos, err, sizeof(err)) != NGSLD_OK)
      error(pos_error_function(err, pars.in_pos), err);
    if (pars.verbose >= 6)
      for (uint64_t s = 0; s < (pars.n_sites < 10 ? pars.n_sites : 10); s++)
        fprintf(stderr, "%lu\t%f\n", (unsigned long)s, ngsld_host_pos_dist(pos)[s]);
  }
  if (ngsld_set_pos_dist(ctx, pos ? ngsld_host_pos_dist(pos) : nullptr) != NGSLD_OK)
    error("ngsld_set_pos_dist", ngsld_last_error(ctx));

  // ---- analyze data (replaces ngsLD.cpp:150-198) ----
  if (pars.verbose >= 1) fprintf(stderr, "==> Launching threads...\n");
  ngsld_params lp;
  lp.max_kb_dist = pars.max_kb_dist;
  lp.max_snp_dist = pars.max_snp_dist;
  lp.min_maf = pars.min_maf;
  lp.ignore_miss_data = pars.ignore_miss_data ? 1 : 0;
  lp.extend_out = pars.extend_out ? 1 : 0;
  lp.rnd_sample = pars.rnd_sample;
  lp.seed = pars.seed;
  lp.first_row = 0;
  uint64_t n_pairs = 0;
  timing_report.mark("positions");
  if (ngsld_plan(ctx, &lp, &n_pairs) != NGSLD_OK) error("ngsld_plan", ngsld_last_error(ctx));
  std::vector<double> maf(pars.n_sites);  // (after the plan: a frequency that ties --min_maf has been settled by then)
  if (ngsld_get_maf(ctx, maf.data()) != NGSLD_OK) error("ngsld_get_maf", ngsld_last_error(ctx));
  timing_report.mark("plan");
  // The rows are formatted on the device (the fprintf block of calc_pair_LD, ngsLD.cpp:310-352, at kernel rates); a
  // batch the device formatter cannot take arrives as records and goes through the --n_threads host formatter as
  // before.  NGSLD_HOST_TEXT=1 keeps everything on the host formatter (A/B, tests).
  if (!host_text_only()) {
    std::vector<const char *> lab;
    if (pos) {
      lab.resize(pars.n_sites);
      for (uint64_t s = 0; s < pars.n_sites; s++) lab[s] = ngsld_host_label(pos, s);
    }
    if (ngsld_set_text_output(ctx, pos ? lab.data() : nullptr, 1) != NGSLD_OK)
      error("ngsld_set_text_output", ngsld_last_error(ctx));
  }
  if (pars.verbose >= 1) fprintf(stderr, "==> Waiting for all threads to finish...\n");
  SinkState sink;
  sink.pars = &pars;
  sink.pos = pos;
  sink.pos_dist = pos ? ngsld_host_pos_dist(pos) : nullptr;
  sink.maf = &maf;
  timing_report.mark("labels to the device");
  rc = ngsld_run(ctx, 0, pars.n_sites, write_batch, &sink);
  timing_report.mark("pair kernels + text + write");
  if (rc == NGSLD_ERR_MAF_RANGE) error("haplo_freq", ngsld_last_error(ctx));
  if (rc != NGSLD_OK) error("ngsld_run", ngsld_last_error(ctx));
  if (pars.verbose >= 2) {  // (level 1 is the reference's default: its stderr stays what the reference prints.  A large share
                            // here means pairs computed at the host's speed: two nearly monomorphic sites each)
    ngsld_replay_stats_t st{};
    (void)ngsld_replay_info(ctx, &st);
    fprintf(stderr, "==> %lu of %lu pairs replayed in the reference's operation order (%lu on the device, %lu on host threads)\n",
            (unsigned long)st.pairs_replayed, (unsigned long)n_pairs, (unsigned long)st.pairs_on_device, (unsigned long)st.pairs_on_host);
  }

  // ---- free memory (ngsLD.cpp:205-222) ----
  if (pars.verbose >= 1) fprintf(stderr, "==> Freeing memory...\n");
  close_output(pars.out_fh);
  ngsld_host_free_pos(pos);
  ngsld_destroy(ctx);
  timing_report.mark("free");
  if (pars.verbose >= 1) fprintf(stderr, "Done!\n");
  return 0;
}
