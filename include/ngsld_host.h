/*
 * ngsld_host.h -- host-side helpers of the drop-in (no device needed): the input readers and the TSV
 * writer, with the reference's semantics.  Plain C ABI so they can be bound and tested like ngsld.h.
 *
 * Reference interfaces mirrored here (paths relative to the reference tree):
 *   shared/read_data.cpp:165-218  read_dist       => ngsld_host_read_pos (pos_dist)
 *   ngsLD.cpp:124-132             labels          => ngsld_host_read_pos (labels, first TAB -> ':')
 *   shared/gen_func.cpp:238-282   read_file       => line rules (skip empty and '#' lines, header offset)
 *   shared/read_data.cpp:28-47    read_geno (bin) => ngsld_host_read_geno_bin (raw doubles; the arithmetic
 *                                                    of that loop runs on the device, ngsld_set_geno_raw)
 *   shared/read_data.cpp:48-104   read_geno (text)=> ngsld_host_read_geno_text
 *   ngsLD.cpp:55-56               size check      => ngsld_host_geno_size_ok
 *   ngsLD.cpp:77,314-351          TSV header/rows => ngsld_host_format_header / ngsld_host_format_pair
 *   ngsLD.cpp:296-298,328-333     hap_maf, chi2   => inside ngsld_host_format_pair (float chi2)
 *   ngsLD.cpp:310-352             fprintf under the mutex => ngsld_host_write_batch (threads format, one ordered write)
 *   ngsLD.cpp:290-306 + gen_func.cpp:1027-1119   one pair in the reference's own operation order
 *                                                 => ngsld_host_replay_pair (the engine's exact-order replay, see ngsld.h)
 */
#ifndef NGSLD_HOST_H
#define NGSLD_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "ngsld.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ngsld_pos ngsld_pos;

/* Number of host threads the readers below may use (text parsing); the TSV writer takes its own argument.
 * Process-wide, default 1 (the reference's --n_threads default, parse_args.cpp:27). */
void ngsld_host_set_threads(int n_threads);

/* Read a position file (chr TAB pos [TAB ...]); `header` != 0 skips one line (--posH).
 * Returns NGSLD_OK or NGSLD_ERR_INVALID with the reference's message text in err. */
int ngsld_host_read_pos(const char *path, int header, uint64_t n_sites, ngsld_pos **out, char *err, size_t errlen);
const double *ngsld_host_pos_dist(const ngsld_pos *p);
const char *ngsld_host_label(const ngsld_pos *p, uint64_t site);
void ngsld_host_free_pos(ngsld_pos *p);
/* Copy of sites [begin, end) (a rank's slab in a multi-GPU run): labels and pos_dist re-indexed from 0. */
ngsld_pos *ngsld_host_pos_slice(const ngsld_pos *p, uint64_t begin, uint64_t end);

/* n_sites == file_size / 8 / n_ind / 3 with the reference's integer divisions (ngsLD.cpp:55). */
int ngsld_host_geno_size_ok(uint64_t file_size, uint64_t n_ind, uint64_t n_sites);
/* Read n_sites*n_ind*3 raw doubles (plain or gzip-compressed file, like gzread) and require EOF after them. */
int ngsld_host_read_geno_bin(const char *path, uint64_t n_ind, uint64_t n_sites, double *out_raw, char *err,
                             size_t errlen);
/* The raw doubles of sites [site_begin, site_begin + n_sites) of the same file (a slab of a streamed run,
 * ngsld_run_streamed): pread for a plain file, gzseek for a compressed one.  No EOF requirement. */
int ngsld_host_read_geno_bin_range(const char *path, uint64_t n_ind, uint64_t site_begin, uint64_t n_sites,
                                   double *out_raw, char *err, size_t errlen);

/* Text genotype input (plain or .gz), the text branch of read_geno (read_data.cpp:48-104): one line per site,
 * fields split on blanks and TABs, only fully numeric fields count and the LAST n_ind*3 (in_probs: GL or
 * posterior triples) or n_ind (called genotypes {-1,0,1,2}) of them are the data; a first line with fewer
 * numeric fields is a header.  out_raw receives n_sites*n_ind*3 doubles for ngsld_set_geno_raw_opts with
 * text_semantics = 1 and log_scale = *out_log_scale (called genotypes are handed over as log triples). */
int ngsld_host_read_geno_text(const char *path, int in_probs, int log_scale, uint64_t n_ind, uint64_t n_sites,
                              double *out_raw, int *out_log_scale, char *err, size_t errlen);
/* What ngsld_host_read_geno_text stores, three times, for a missing call (-1) of a genotype file: log(1/3), read_data.cpp:94.
 * A matrix whose individuals without data are exactly this triple (text semantics, log scale, no --call_geno) has the pairs of
 * such sites replayed on the device like any other (ngsld.h, "Exact-order replay"). */
double ngsld_host_missing_call_log(void);

/* TSV text.  Both return the number of bytes written (no NUL needed), 0 if cap is too small.
 * NaN is printed as "-nan": every NaN the reference prints comes from an x86 invalid operation
 * (0/0 ...), whose default NaN has the sign bit set, and glibc's %f shows that sign. */
size_t ngsld_host_format_header(char *buf, size_t cap, int extend_out);
size_t ngsld_host_format_pair(char *buf, size_t cap, const char *label1, const char *label2, double dist,
                              const ngsld_rec_std *std_rec, const ngsld_rec_ext *ext_rec, double maf1, double maf2);

/* "%f" (decimals = 6) or "%.0f" (decimals = 0) of one double, byte-identical to glibc's printf for finite
 * values (exact binary value, round-half-even), with "-nan" / "inf" / "-inf".  cap >= 400.  Returns bytes written. */
size_t ngsld_host_format_double(char *buf, size_t cap, double v, int decimals);

/* Format a whole batch (what calc_pair_LD's fprintf block does for every pair of the batch, ngsLD.cpp:310-352)
 * with n_threads threads and write it to file descriptor fd in (s1, s2) order.  pos may be NULL (labels print as
 * "(null)"), pos_dist NULL = all INFINITY; maf = per-site allele frequencies (ngsld_get_maf). */
int ngsld_host_write_batch(const ngsld_batch *b, const ngsld_pos *pos, const double *pos_dist, const double *maf,
                           int n_threads, int fd);

/* gzip-compressed output (optional; the reference writes plain text only): whatever is written to
 * *fd_to_write is cut into 4 MiB blocks, each deflated by one of n_threads workers into a gzip member of its own, and
 * written to `path` in order -- a valid .gz file (multi-member), content identical to the plain output.
 * NGSLD_GZ_LEVEL=1..9 (default 1).  *fd_to_write belongs to the CALLER: close (or fclose) it, then call
 * ngsld_host_gz_close, which waits for the end of the stream and returns NGSLD_OK when every block reached the file.  A
 * write or deflate error never blocks the producer: the rest of the stream is read and dropped, and close reports it. */
typedef struct ngsld_gz ngsld_gz;
int ngsld_host_gz_open(const char *path, int n_threads, ngsld_gz **out, int *fd_to_write);
int ngsld_host_gz_close(ngsld_gz *gz);

/* One pair evaluated on the host in the reference's own operation order: read_geno's arithmetic on the two sites' raw
 * values ([n_ind][3] each, as ngsld_set_geno_raw_opts takes them; opts->on_device is ignored), call_geno, est_maf
 * (sequential, gen_func.cpp:974-1009), exp, pearson_r, haplo_freq (gen_func.cpp:1027-1119: sequential sums, the
 * sequential renormalisation, no fused multiply-add) and D / D' / r2 (ngsLD.cpp:296-306).  This is the arithmetic the
 * engine replays for the pairs whose outcome the reference's own rounding decides (ngsld_set_replay_source).
 * ext_rec and maf_out (2 doubles) may be NULL.  Returns NGSLD_OK, or NGSLD_ERR_MAF_RANGE where haplo_freq calls error(). */
int ngsld_host_replay_pair(const double *raw1, const double *raw2, uint64_t n_ind, const ngsld_geno_opts *opts,
                           ngsld_rec_std *std_rec, ngsld_rec_ext *ext_rec, double *maf_out);

#ifdef __cplusplus
}
#endif
#endif
