/*
 * ngsld.h -- C-ABI of the MI355X-native pairwise-LD engine (drop-in for the pair-LD path of
 * fgvieira/ngsLD v1.2.1).  Plain C: opaque handle, pointers and sizes, int return codes, no C++ or
 * torch types, no exceptions across the boundary.
 *
 * What this boundary replaces in the reference (paths relative to the reference tree):
 *   ngsLD.cpp:153-198   threadpool_create -> per-s1 threadpool_add(calc_pair_LD) -> threadpool_wait
 *                       -> threadpool_destroy            => ngsld_create / ngsld_plan / ngsld_run / ngsld_destroy
 *   ngsLD.cpp:229-359   calc_pair_LD (window walk, filters, haplo_freq, pearson_r, D/D'/r2)
 *                                                         => the HIP pair kernel behind ngsld_run
 *   ngsLD.cpp:103-114   est_maf + exp() + expected genotypes => ngsld_set_geno_raw (device prep kernel)
 *   ngsLD.hpp:11-44     `params`: geno_lkl, maf, pos_dist, max_kb_dist, max_snp_dist, min_maf,
 *                       ignore_miss_data, extend_out      => ngsld_set_geno_*, ngsld_set_pos_dist, ngsld_params
 * The text side (labels, dist column, %f formatting, chi2 in float) stays with the caller; helpers
 * that mirror it live in ngsld_host.h.
 *
 * Errors: the reference is fatal-on-error (error(), gen_func.cpp:12-18).  This library never exits:
 * every entry point returns NGSLD_OK or a negative code and ngsld_last_error() gives the text; the
 * ngsLD command-line wrapper turns them back into the reference's stderr format and exit(-1).
 *
 * Threading: one thread per ngsld_ctx at a time.  ngsld_run blocks; the sink is called on the calling
 * thread, batches arrive in increasing (s1, s2) order.
 */
#ifndef NGSLD_H
#define NGSLD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ngsld_ctx ngsld_ctx;

enum {
  NGSLD_OK = 0,
  NGSLD_ERR_INVALID = -1,    /* bad argument or call order (cf. threadpool_invalid, threadpool.h:39) */
  NGSLD_ERR_DEVICE = -2,     /* HIP runtime failure, or no usable gfx950 device */
  NGSLD_ERR_NOMEM = -3,      /* host or device allocation failed */
  NGSLD_ERR_NAN = -4,        /* "NaN found! Is the file format correct?" (read_data.cpp:42-45) */
  NGSLD_ERR_MAF_RANGE = -5,  /* "invalid allele frequencies" (gen_func.cpp:1030-1031) */
  NGSLD_ERR_SINK = -6,       /* the sink callback returned non-zero */
  NGSLD_ERR_UNSUPPORTED = -7 /* problem shape outside the supported range (n_sites or n_ind >= 2^32 - 64) */
};

/* The fields of `params` (ngsLD.hpp:11-44) that calc_pair_LD reads. */
typedef struct {
  uint64_t max_kb_dist;      /* 0 = no distance limit (ngsLD.cpp:252) */
  uint64_t max_snp_dist;     /* 0 = no limit (ngsLD.cpp:258) */
  double min_maf;            /* ngsLD.cpp:264-275: s1 below -> row empty, s2 below -> pair skipped */
  int32_t ignore_miss_data;  /* gen_func.cpp:1089 */
  int32_t extend_out;        /* also produce ngsld_rec_ext */
  double rnd_sample;         /* ngsLD.cpp:277: keep a pair iff its Tausworthe draw <= rnd_sample; 0 or 1 = keep all */
  uint64_t seed;             /* --seed of the master gsl_rng_taus stream (ngsLD.cpp:69-70); used iff rnd_sample < 1 */
  uint64_t first_row;        /* global index of this context's site 0 when it holds a slab of a larger matrix (multi-GPU
                                row sharding): that many draws of the master stream are skipped so that every row gets
                                the seed it has in the single-process run (ngsLD.cpp:165-166).  0 otherwise. */
} ngsld_params;

/* Standard columns computed per pair (ngsLD.cpp:290-306): 32 bytes. */
typedef struct {
  double r2_ExpG; /* pearson_r(expected_geno[s1], expected_geno[s2])^2 */
  double D;
  double Dp;
  double r2;
} ngsld_rec_std;

/* What the extended columns (ngsLD.cpp:336-349) need beyond per-site data: 40 bytes. */
typedef struct {
  double hap[4];       /* hap_freq[0..3] after the EM */
  uint32_t n_ind_data; /* sample_size: individuals used by the EM (bit-exact) */
  uint32_t n_iter;     /* return value of haplo_freq */
} ngsld_rec_ext;

/* One unit of the pair space: the pairs (s1, s2_begin + c) for the bits c set in `mask` (c < count <= 64), i.e.
 * the candidates of row s1 that survived the maf[s2] skip (ngsLD.cpp:270) and the random sub-sampling
 * (ngsLD.cpp:277).  Their records are consecutive, starting at `first_record`, in increasing c. */
typedef struct {
  uint32_t s1, s2_begin, count, reserved;
  uint64_t mask;
  uint64_t first_record;
} ngsld_item;

/* One batch of results: all pairs of rows [s1_begin, s1_end), in (s1, s2) order.  `items` lists them row by row
 * (increasing s1, then increasing s2_begin), first_record relative to this batch.  Pointers are valid only
 * during the callback. */
typedef struct {
  uint64_t s1_begin, s1_end;
  uint64_t n_pairs;
  uint64_t n_items;
  const ngsld_item *items;   /* [n_items] */
  const ngsld_rec_std *std;  /* [n_pairs] */
  const ngsld_rec_ext *ext;  /* [n_pairs] or NULL when extend_out == 0 */
  /* Only after ngsld_set_text_output(..., 1): the batch's TSV rows, formatted on the device -- byte for byte what
   * ngsld_host_write_batch writes for this batch (ngsLD.cpp:314-351).  When text != NULL, items / std / ext are NULL
   * (the records were not copied to the host).  A batch the device formatter cannot take (a value beyond its
   * fast path) arrives as records, text == NULL, as without text output. */
  const char *text;
  uint64_t text_len;
} ngsld_batch;

typedef int (*ngsld_sink_fn)(void *user, const ngsld_batch *batch);

/* Fill dst with the raw values ([site][ind][3] doubles, as ngsld_set_geno_raw_opts takes them) of sites
 * [site_begin, site_begin + n_sites).  Called on a library thread, never concurrently.  Non-zero = failure. */
typedef int (*ngsld_read_sites_fn)(void *user, uint64_t site_begin, uint64_t n_sites, double *dst);

const char *ngsld_version(void);

/* Bind to HIP device `device` (must be gfx950).  Fails with NGSLD_ERR_DEVICE when there is no GPU:
 * there is no CPU fallback behind this API. */
int ngsld_create(int device, ngsld_ctx **ctx);
void ngsld_destroy(ngsld_ctx *ctx);
/* Text of the last failure on ctx (ctx may be NULL: last ngsld_create failure of this thread). */
const char *ngsld_last_error(const ngsld_ctx *ctx);

/* Genotype likelihoods exactly as the reference's binary input holds them: n_sites*n_ind*3 doubles,
 * [site][ind][geno] (read_data.cpp:28-47), natural scale or logs (log_scale).  The device does what
 * read_geno + main do to them: log, -inf -> -1e15, log-normalise, NaN check, est_maf, exp, expected
 * genotypes.  `on_device` != 0: gl_raw is a device pointer on this ctx's device (not modified).
 * `ignore_miss_data` here also picks the device layout of the matrix (for some cohort sizes the kernels measured best
 * differ with the flag, and their plane padding with them): give the value ngsld_params.ignore_miss_data will have.
 * A plan with the other value is computed all the same -- same records, possibly on the slower of two kernels. */
int ngsld_set_geno_raw(ngsld_ctx *ctx, const double *gl_raw, uint64_t n_sites, uint64_t n_ind, int log_scale,
                       int ignore_miss_data, int on_device);
/* The same with every input-side switch of the reference's main():
 *   text_semantics  the values came from a text (.gz) genotype file: plain log() with no -inf -> -1e15
 *                   replacement and no NaN check, as the text branch of read_geno (read_data.cpp:83-99)
 *   call_geno       harden the likelihoods first (ngsLD.cpp:92-98 -> call_geno, gen_func.cpp:886-914):
 *                   best genotype below N_thresh -> missing, at or above call_thresh -> called
 *   per_individual_only  never switch to the genotype-combination kernel (ngsld_pair_kernel): a caller that computes
 *                   parts of one matrix in several contexts sets it unless ALL parts qualify, so that every part
 *                   runs the same arithmetic */
typedef struct {
  int32_t log_scale;
  int32_t ignore_miss_data;
  int32_t on_device;
  int32_t text_semantics;
  int32_t call_geno;
  int32_t per_individual_only;
  double N_thresh;
  double call_thresh;
} ngsld_geno_opts;
int ngsld_set_geno_raw_opts(ngsld_ctx *ctx, const double *gl_raw, uint64_t n_sites, uint64_t n_ind,
                            const ngsld_geno_opts *opts);
/* The reference's own data contract at calc_pair_LD: normalised normal-space geno_lkl
 * [site][ind][3] and maf[site] already computed by the caller (ngsLD.hpp:36-37). */
int ngsld_set_geno_lkl(ngsld_ctx *ctx, const double *geno_lkl, const double *maf, uint64_t n_sites, uint64_t n_ind,
                       int on_device);
/* Copy the per-site allele frequencies (est_maf) to host memory, n_sites doubles. */
int ngsld_get_maf(ngsld_ctx *ctx, double *maf_out);
/* pos_dist[s] = bp gap to the previous site, INFINITY at a chromosome change (read_data.cpp:165-218).
 * NULL = all INFINITY (the reference's no --pos case, ngsLD.cpp:134).  Host pointer, n_sites doubles. */
int ngsld_set_pos_dist(ngsld_ctx *ctx, const double *pos_dist);

/* Fix the filters and enumerate the pair space (the s2 walk of ngsLD.cpp:240-282 for every s1).
 * Returns the total number of pairs that reach haplo_freq. */
int ngsld_plan(ngsld_ctx *ctx, const ngsld_params *params, uint64_t *n_pairs);
/* Host views of the plan, valid until the next ngsld_plan/ngsld_set_*: row_off [n_sites+1] (pairs
 * before each row), row_end [n_sites].  Used to shard rows across GPUs by pair count. */
int ngsld_plan_rows(ngsld_ctx *ctx, const uint64_t **row_off, const uint32_t **row_end);

/* Compute every pair of rows [s1_begin, s1_end) and hand the records to `sink`, batch by batch.  The records of a batch are
 * in pinned host memory the pair kernels wrote directly (no device copy, no D2H behind the last kernel): the host-resident
 * rate is the kernels' rate to within 1 %. */
int ngsld_run(ngsld_ctx *ctx, uint64_t s1_begin, uint64_t s1_end, ngsld_sink_fn sink, void *user);

/* Device-side TSV for ngsld_run (replaces the fprintf block of calc_pair_LD, ngsLD.cpp:310-352, at kernel rates):
 * labels = n_sites C strings as they appear in the first two columns (ngsLD.cpp:127-132), or NULL for the
 * reference's no --pos output "(null)".  enable != 0: ngsld_run hands over text (ngsld_batch.text) instead of records.
 * Call after ngsld_set_geno_* (the labels belong to that matrix); the dist column comes from ngsld_set_pos_dist and
 * needs its finite gaps to be integers (as read_dist produces them), otherwise batches arrive as records. */
int ngsld_set_text_output(ngsld_ctx *ctx, const char *const *labels, int enable);

/* Optional: start allocating the pinned host buffers of the text batches (three, each for one batch of rows of about
 * bytes_per_row) on a library thread, so that pinning them overlaps whatever the caller does next (reading,
 * ngsld_set_geno_*, ngsld_plan) instead of the first batches of ngsld_run.  A batch that needs more gets a larger buffer
 * then.  Callable any time after ngsld_create. */
int ngsld_reserve_text_buffers(ngsld_ctx *ctx, uint64_t bytes_per_row);

/* ---- Exact-order replay ------------------------------------------------------------------------------------
 * The kernels evaluate every pair with reordered arithmetic (tree sums, fused multiply-adds), which agrees with the
 * reference to ~1e-15 wherever the outcome is well conditioned.  A few outcomes are decided by the reference's own
 * rounding noise: D' and r2 of a pair with a site (nearly) monomorphic in the estimated haplotypes (0/0-type
 * quotients, ngsLD.cpp:296-306: -nan, 0.000000 or inf), nIter when eps lands within 1e-12 of EPSILON
 * (gen_func.cpp:1054), the maf < min_maf tests when a frequency ties --min_maf (ngsLD.cpp:264-275), r2_ExpG of a
 * pair of sites whose expected genotypes are both nearly constant (1 / (std1 std2) > 2^13; ngsLD.cpp:365-367).  The kernels flag those pairs and
 * the engine re-evaluates them on the host in the reference's operation order (sequential sums over individuals,
 * the sequential renormalisation, no fused multiply-add: ngsld_host_replay_pair in ngsld_host.h) and overwrites
 * their records -- before a batch reaches the sink, before it is formatted on the device, before ngsld_run_device
 * returns.  On by default.
 *
 * The replay needs the two sites' values again.  ngsld_set_replay_source registers where to get them: `read` fills
 * dst with the RAW values of sites [site_begin, site_begin + n) of this context ([site][ind][3], exactly what
 * ngsld_set_geno_raw_opts / ngsld_set_geno_lkl was given) and may be called from several library threads, one call
 * at a time.  With a source the replayed records are the reference's own bits (same libm).  Without one the
 * library reads its prepped planes back from the device: same operation order, inputs that differ from the
 * reference's by the device's exp/log rounding (exact for ngsld_set_geno_lkl), and --min_maf ties are left to the
 * device's est_maf.  Call after ngsld_set_geno_* (the source belongs to that matrix), before ngsld_plan. */
int ngsld_set_replay_source(ngsld_ctx *ctx, ngsld_read_sites_fn read, void *user);
/* The common case of a source -- the caller still HOLDS the matrix it gave ngsld_set_geno_raw_opts / ngsld_set_geno_lkl, as
 * the reference's main does throughout (ngsLD.cpp:86-89): `values` = that same host array, [site][ind][3], which must stay
 * valid and unchanged until the context is destroyed or given other data.  Read in place by the replay threads: no callback,
 * no copy, no lock (the callback form serialises its calls).  NULL removes it.  Replaces any registered callback. */
int ngsld_set_replay_matrix(ngsld_ctx *ctx, const double *values);
/* enable == 0: no replay, every record is the kernels' own value. */
int ngsld_set_replay(ngsld_ctx *ctx, int enable);
/* Pairs replayed by the last ngsld_run / ngsld_run_device (+ ngsld_finish_device) and sites re-evaluated by the last
 * ngsld_plan and run.  Either pointer may be NULL. */
int ngsld_replay_stats(ngsld_ctx *ctx, uint64_t *pairs, uint64_t *sites);

/* Where the flagged pairs are replayed.  Called-genotype matrices: on the device (their values are the same bits there).
 * LIKELIHOOD matrices: a few dozen pairs per 10^8 on SNP-called input, which host threads replay -- but on matrices that are
 * NOT SNP-called (the reference's README.md:73: "comparisons will show up as nan or inf") every pair with a (nearly)
 * monomorphic site is flagged, a third of all pairs at 20 % monomorphic sites.  Those are replayed on the DEVICE too, a
 * wavefront per pair, in the reference's operation order (ld_replay_lkl.hip) -- on an "exact store": the matrix once more in
 * device memory, normal-space likelihoods and est_maf as the reference holds them when calc_pair_LD runs, i.e. through the
 * HOST's libm.  Data given through ngsld_set_geno_lkl is that already (nothing is built, the replay is on the device from the
 * first pair on); without a replay source the device's own prepped values serve (the replay then runs on the device's
 * exp / log rounding, as the host's did); with a source the store is built from the caller's raw values -- 17 libm calls per
 * triple, ~0.25 us per individual and site on one thread, once per matrix -- the first time a run has flagged more pairs than
 * the host should replay (more than half as many as the matrix has sites).  The build runs on threads of its own BESIDE the
 * ngsld_run that asked for it, in site order: a batch's replay waits only until the builder has passed the batch's last
 * window; the run returns when the build has ended (the source is read during runs only).  ngsld_finish_device builds it
 * before it replays.
 * mode: 0 never (host replay only), 1 as described (default), 2 build at the first flagged pair.
 * (0.4.0) Such a matrix is recognised when it is set: ngsld_set_geno_* marks the DEGENERATE sites -- the one-locus EM of the site,
 * from its est_maf frequency, ends below 3e-6: every pair of such a site (nearly) is one the replay settles.  Where the device can
 * replay (mode != 0) the pair kernels leave the EM of those pairs out -- the replay, exact for any pair, is their one evaluation --
 * and ngsld_run with text output computes in groups of up to 2^25 pairs (one launch of pair kernels + one replay per group) instead
 * of batch by batch; the sink sees the same batches.  ngsld_replay_stats_t.sites_degenerate; NGSLD_REPLAY_SKIP=0 turns the marks off. */
int ngsld_set_exact_store(ngsld_ctx *ctx, int mode);
typedef struct {
  uint64_t pairs_flagged;      /* pairs the kernels of the last run flagged */
  uint64_t pairs_replayed;     /* ... of them replayed (all, unless the replay is off) */
  uint64_t pairs_on_device;    /* ... on the device */
  uint64_t pairs_on_host;      /* ... on host threads */
  uint64_t sites_reevaluated;  /* sites the host re-evaluated for the last plan + run */
  int32_t exact_store;         /* 0 none, 1 the planes themselves serve as the store, 2 built from the replay source */
  int32_t text_rows_patched;   /* text runs: rows of pairs replayed on the host whose value columns were overwritten in the host's copy of the text */
  double exact_store_build_s;  /* host seconds the build took (once per matrix) */
  uint64_t sites_degenerate;   /* (0.4.0) sites of the matrix marked degenerate -- their one-locus EM ends below 3e-6: the pair kernels leave
                                  the EM of such a site's pairs to the exact-order replay, which was going to start them over (0 on SNP-called
                                  input, or where the device cannot replay: nothing is skipped then) */
} ngsld_replay_stats_t;
int ngsld_replay_info(ngsld_ctx *ctx, ngsld_replay_stats_t *out);

/* Same computation with the records left in caller-owned DEVICE memory (no host transfer):
 * d_std holds ngsld_rec_std[n], d_ext ngsld_rec_ext[n] (may be NULL), n = row_off[s1_end] -
 * row_off[s1_begin], record k = global pair index - row_off[s1_begin].  `hip_stream` is a hipStream_t
 * (NULL = the ctx's own stream); the call returns after enqueueing when a stream is given. */
int ngsld_run_device(ngsld_ctx *ctx, uint64_t s1_begin, uint64_t s1_end, void *d_std, void *d_ext, void *hip_stream);
/* After ngsld_run_device on a caller's stream: waits for that stream, replays the flagged pairs and patches the device
 * records (exact-order replay, above).  A context has ONE pending device run: the next ngsld_run_device, ngsld_run,
 * ngsld_plan or ngsld_set_geno_* finishes it first (as this call would), so a flagged record is never left with the
 * kernels' own value.  A no-op after a run on the ctx's own stream (hip_stream == NULL does all of it before returning).
 * BUFFER LIFETIME: the replay of a pending run reads the matrix it was started with -- through the registered replay source
 * or matrix (ngsld_set_replay_source / ngsld_set_replay_matrix) -- at the moment the run is finished, whoever finishes it.
 * Keep that host array (or what the callback reads) valid and UNCHANGED until ngsld_finish_device, or the call that
 * finishes the run implicitly, has returned: refilling the buffer with the next matrix before ngsld_set_geno_* would have
 * the old run's flagged records recomputed from the new values, silently. */
int ngsld_finish_device(ngsld_ctx *ctx);

/* Timing of the pair kernel launches issued by the last ngsld_run / ngsld_run_device, measured with
 * HIP events on the stream they ran on (synchronises that stream).  Any pointer may be NULL. */
int ngsld_last_kernel_time(ngsld_ctx *ctx, double *total_ms, uint64_t *n_launches, uint64_t *n_pairs);

/* Which pair kernel family the genotype data set last will run on: "group" (8 / 16 / 32 lanes per pair), "run" (one
 * wavefront per pair, up to 640 individuals), "ab" (one wavefront per pair, a/b form of the EM step: 641..960 individuals), "multi" / "multi-ab" (several wavefronts per pair, P form / a/b form, up to 7,680 individuals), "stream" (beyond: one vector in registers, the other re-read in every EM iteration), or "hard" -- every
 * likelihood triple is a called genotype or "no data" (text genotypes, --call_geno: ngsLD.cpp:92-98,
 * read_data.cpp:83-99), the pairs run on their 16 genotype-combination counts.  "" before any data is set. */
const char *ngsld_pair_kernel(const ngsld_ctx *ctx);
/* The same decision without a device or a context: which kernel family and shape a likelihood matrix of n_ind individuals set
 * with this ignore_miss_data runs on, as text -- "<family> <wavefronts per pair>x<individuals per lane> lanes=<lanes per pair>
 * np=<padded individuals per genotype plane>", e.g. "run 1x8 lanes=64 np=512" for 500.  (Called-genotype matrices take "hard"
 * whatever the cohort, up to 4,096 individuals.)  Returns NGSLD_ERR_UNSUPPORTED for a cohort size outside the supported range,
 * NGSLD_ERR_INVALID when buf is too short.  tests/golden/dispatch_table.txt is this function over 1..12,000. */
int ngsld_describe_dispatch(uint64_t n_ind, int ignore_miss_data, char *buf, size_t buf_len);

/* Tuning knobs (optional): pairs per work item, max pairs per batch of ngsld_run. 0 keeps the default -- record batches of
 * 2^24 pairs (the pair kernels write them straight into two pinned host buffers of that many records; halved, down to 2^16,
 * on a host that cannot pin them), text batches of 2^19 rows.  A value given here is taken as it is for record batches and as
 * an upper bound for text batches. */
int ngsld_set_tuning(ngsld_ctx *ctx, uint32_t pairs_per_item, uint64_t batch_pairs);

/* On-device self test of the wavefront primitives the pair kernel relies on (cross-lane fold
 * reduction, refined reciprocal).  Returns NGSLD_OK or NGSLD_ERR_DEVICE with a message. */
int ngsld_selftest(ngsld_ctx *ctx);

/* ---- Streaming: matrices larger than the device budget (windowed runs only) ------------------------------
 * The reference keeps the whole matrix in host memory (twice during the transpose, ngsLD.cpp:87-89).  Here a
 * windowed run (max_kb_dist and/or max_snp_dist > 0) can be cut into slabs of rows: slab k holds the sites
 * [row_begin, site_end) = its rows plus the halo their windows reach into, and computes the rows
 * [row_begin, row_end).  Two contexts on the same device alternate, so the file read + upload + prep of slab
 * k+1 overlap the pair kernels of slab k; neither the host nor the device ever holds more than two slabs. */
typedef struct {
  uint64_t row_begin, row_end; /* rows (s1) this slab computes */
  uint64_t site_end;           /* one past the last site any of those rows pairs with */
} ngsld_slab;

/* Upper end of the s2 walk of every row from the distance / SNP-count limits alone (ngsLD.cpp:240-262):
 * row s1 pairs with sites (s1, row_end[s1]).  Host only, no device needed; pos_dist NULL = all INFINITY. */
int ngsld_window_ends(const double *pos_dist, uint64_t n_sites, const ngsld_params *params, uint32_t *row_end);

/* Cut rows [0, n_sites) into slabs of at most max_slab_sites sites (rows + halo).  slabs has room for `cap`
 * entries (n_sites always suffices).  NGSLD_ERR_NOMEM when a single row's window does not fit. Host only. */
int ngsld_plan_slabs(const double *pos_dist, uint64_t n_sites, const ngsld_params *params, uint64_t max_slab_sites,
                     ngsld_slab *slabs, uint64_t cap, uint64_t *n_slabs);

/* How many sites of n_ind individuals fit a slab when `budget_bytes` of device memory may be used in total
 * (both contexts, their record buffers included, the matrix priced THREE times per context: the planes, and the exact store of
 * the device-side replay with its individual-major copy, which input that is not SNP-called makes the library build); 0 when
 * the budget is too small for any. */
uint64_t ngsld_slab_sites_for_budget(uint64_t n_ind, uint64_t budget_bytes);
/* (0.4.0) The same with the matrix priced `matrix_copies` times per context: 1 = the planes only -- what a run NEEDS (SNP-called
 * input never builds the store; un-called input without room for it has its flagged pairs replayed on host threads, slower and
 * the same bytes), 3 = ngsld_slab_sites_for_budget.  A caller that cannot stream (text input, no window) asks with 1 before it
 * gives up. */
uint64_t ngsld_sites_for_budget(uint64_t n_ind, uint64_t budget_bytes, int matrix_copies);

/* (0.4.0) ngsld_replay_info of the last ngsld_run_streamed / ngsld_run_streamed_text of this process, added up over its slabs
 * (exact_store: the largest of the slabs' values; sites_degenerate counts a halo site once per slab that holds it). */
int ngsld_streamed_replay_info(ngsld_replay_stats_t *out);

/* (0.4.0) A cap, in bytes, on the device memory this PROCESS takes on `device` from now on (0: none) -- what the drop-in binary's
 * --max_gpu_mem sets.  Slab sizes are the caller's to plan (ngsld_slab_sites_for_budget); the cap is looked at where the library
 * allocates what it can do without: the exact store of the device-side replay (without room: flagged pairs on host threads) and
 * its individual-major copy (without room: the wavefront-per-pair replay kernel, which reads the store's own layout). */
int ngsld_set_memory_budget(int device, uint64_t bytes);

/* Free and total memory of HIP device `device`, in bytes. */
int ngsld_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);


/* The whole job, slab by slab: create two contexts on `device`, and for every slab read -> ngsld_set_geno_raw_opts
 * -> ngsld_set_pos_dist -> ngsld_plan (first_row = row_begin) -> ngsld_run.  The sink sees the batches of all
 * slabs in global (s1, s2) order with GLOBAL site indices in ngsld_batch and ngsld_item.  maf_out (n_sites
 * doubles, may be NULL) receives est_maf of every site; entries are final before the first batch that refers to
 * them is handed to the sink.  pos_dist: n_sites doubles or NULL.  err (may be NULL) receives the message. */
int ngsld_run_streamed(int device, uint64_t n_sites, uint64_t n_ind, const double *pos_dist,
                       const ngsld_params *params, const ngsld_geno_opts *opts, uint64_t max_slab_sites,
                       ngsld_read_sites_fn read, void *read_user, double *maf_out, ngsld_sink_fn sink,
                       void *sink_user, uint64_t *n_pairs, uint64_t *n_slabs, char *err, size_t errlen);

/* The same with device-side TSV (ngsld_set_text_output on every slab): labels = n_sites C strings or NULL ("(null)"),
 * text_output != 0 makes the sink receive text batches (ngsld_batch.text), in global (s1, s2) order. */
int ngsld_run_streamed_text(int device, uint64_t n_sites, uint64_t n_ind, const double *pos_dist,
                            const ngsld_params *params, const ngsld_geno_opts *opts, uint64_t max_slab_sites,
                            ngsld_read_sites_fn read, void *read_user, double *maf_out, ngsld_sink_fn sink,
                            void *sink_user, uint64_t *n_pairs, uint64_t *n_slabs, char *err, size_t errlen,
                            const char *const *labels, int text_output);

/* ---- Several GPUs of one node, one process (replaces the thread-pool section of the reference's main(),
 * ngsLD.cpp:153-198, at the scale of devices) ----------------------------------------------------------------------
 * The rows are cut into n_devices contiguous parts with equal candidate-pair counts; part k runs on devices[k] from its
 * own host thread and holds only the sites its rows pair with.  Pairs are independent, so nothing is exchanged while
 * computing; the records of the parts, in part order, are the single-device run bit for bit. */

/* Cut rows [0, n_sites) into n_parts parts: parts[k] = {row_begin, row_end, site_end} -- part k computes the rows
 * [row_begin, row_end) and needs the sites [row_begin, site_end).  Host only. */
int ngsld_plan_parts(const double *pos_dist, uint64_t n_sites, const ngsld_params *params, int n_parts, ngsld_slab *parts);

/* Sink of a multi-device run: called on part `part`'s own thread -- the batches of one part arrive in (s1, s2) order,
 * different parts call concurrently.  Site indices in the batch are global. */
typedef int (*ngsld_multi_sink_fn)(void *user, int part, const ngsld_batch *batch);

/* The whole job: gl_raw = the matrix in host memory ([site][ind][3], as ngsld_set_geno_raw_opts takes it), or NULL and
 * `read` delivers the raw values of any site range (it is called from several threads at once).  A windowed run uploads
 * every part's slab from host memory over the part's own PCIe link; an all-pairs run on >= 2 distinct devices puts the
 * matrix on the first device once and hands it to the others with ONE ncclBroadcast (RCCL over xGMI, loaded on demand).
 * maf_out (n_sites doubles, may be NULL) receives est_maf; it is complete before the first batch reaches the sink.
 * labels / text_output as in ngsld_run_streamed_text.  pairs_per_part (n_devices entries, may be NULL) receives the
 * pairs each part computed.  Every part must fit its device (a job that does not is for ngsld_run_streamed). */
int ngsld_run_multi(const int *devices, int n_devices, uint64_t n_sites, uint64_t n_ind, const double *pos_dist,
                    const ngsld_params *params, const ngsld_geno_opts *opts, const double *gl_raw,
                    ngsld_read_sites_fn read, void *read_user, double *maf_out, ngsld_multi_sink_fn sink, void *sink_user,
                    const char *const *labels, int text_output, uint64_t *pairs_per_part, char *err, size_t errlen);

/* How the last ngsld_run_multi of this process handed the matrix to its devices (the one collective of the
 * design is this broadcast): uploaded slab by slab over each part's own PCIe link, copied device to device from the first
 * device, or ONE ncclBroadcast over RCCL / xGMI. */
enum { NGSLD_DIST_NONE = 0, NGSLD_DIST_UPLOAD = 1, NGSLD_DIST_PEER_COPY = 2, NGSLD_DIST_RCCL = 3 };
int ngsld_multi_last_distribution(void);

/* The RCCL calls of ngsld_run_multi's broadcast -- dlopen of librccl, ncclCommInitAll, ncclGroupStart / ncclBroadcast /
 * ncclGroupEnd, ncclCommDestroy -- on a communicator of ONE device: `bytes` of a known pattern go to the device, through an
 * in-place broadcast, and back.  What a one-GPU box can prove about that path; NGSLD_OK, or NGSLD_ERR_DEVICE with the
 * reason in err. */
int ngsld_rccl_selftest(int device, uint64_t bytes, char *err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
