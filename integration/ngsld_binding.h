// ngsld_binding.h -- what a maintainer of ngsLD adds to run its pair loop on an MI355X through libngsld.so.
//
// Reference side:
//   * ngsLD.cpp, above main:           #include "ngsld_binding.h"
//   * ngsLD.cpp:153-198 (thread pool creation, one calc_pair_LD job per site, wait, destroy) become ONE line:
//                                      ngsld_compute_all(pars);
//   * calc_pair_LD's print block (ngsLD.cpp:311-351, with the two hap-derived frequencies of :296-298 it prints) moves, as
//     it stands, into                  void print_pair(params*, s1, s2, dist, r2pear, D, Dp, r2, hap_freq, n_ind_data, n_iter);
//     which the record sink below calls row by row (with device-side text -- the default -- the rows arrive formatted and
//     the sink only writes bytes; a batch the device formatter cannot take still arrives as records).
//   * Makefile: -lngsld (and nothing of GSL's statistics any more: pearson_r is computed on the device).
// Everything else of main() -- argument parsing, read_geno, call_geno, est_maf, the exp() loop, read_dist, labels, the
// output file and header, the frees -- stays as it is: the library takes the reference's own normal-space geno_lkl and maf.
//
// This file is compiled for real: oracle/build_ref.sh streams the reference's main() from where it lies, replaces the
// thread-pool section by the one line above, generates print_pair from the reference's own lines, and links libngsld.so
// (oracle/_ref/libngsld_ref_hip.so, entry ref_main_hip).  tests/test_gpu_ref_main_patched.py holds that program's TSV to the
// unpatched reference program's over the same argv and files.
//
// Environment (tests): NGSLD_BINDING_TEXT=0 -> records + print_pair for every batch instead of device-side text.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ngsld.h"

// calc_pair_LD's own print block (ngsLD.cpp:311-351), moved into a function by the maintainer
void print_pair(params *pars, uint64_t s1, uint64_t s2, double dist, double r2pear, double D, double Dp, double r2,
                double *hap_freq, uint64_t n_ind_data, uint64_t n_iter);

// One batch of finished pairs: rows s1_begin..s1_end-1, row by row, partners in increasing order (calc_pair_LD's order for
// one thread).  Text batches are written as they are; record batches go through the reference's own print block.
static int ngsld_binding_sink(void *user, const ngsld_batch *b) {
  params *pars = (params *)user;
  if (b->text != NULL) return fwrite(b->text, 1, b->text_len, pars->out_fh) == b->text_len ? 0 : 1;
  uint64_t cur_s1 = UINT64_MAX, cur_s2 = 0;
  double dist = 0;
  for (uint64_t i = 0; i < b->n_items; i++) {
    const ngsld_item *it = &b->items[i];
    if (it->s1 != cur_s1) {  // a new row: dist restarts at the row's own site (ngsLD.cpp:236-241)
      cur_s1 = it->s1;
      cur_s2 = it->s1;
      dist = 0;
    }
    uint64_t k = it->first_record;
    for (uint32_t c = 0; c < it->count; c++) {
      const uint64_t s2 = (uint64_t)it->s2_begin + c;
      while (cur_s2 < s2) dist += pars->pos_dist[++cur_s2];  // the running sum of ngsLD.cpp:241
      if (!((it->mask >> c) & 1)) continue;                  // a partner calc_pair_LD skips (maf, --rnd_sample)
      const ngsld_rec_std *r = &b->std[k];
      double hap[4] = {0, 0, 0, 0};
      uint64_t n_ind_data = 0, n_iter = 0;
      if (b->ext != NULL) {  // (--extend_out: the haplotype frequencies, sample size and iteration count of the pair)
        memcpy(hap, b->ext[k].hap, sizeof hap);
        n_ind_data = b->ext[k].n_ind_data;
        n_iter = b->ext[k].n_iter;
      }
      print_pair(pars, cur_s1, s2, dist, r->r2_ExpG, r->D, r->Dp, r->r2, hap, n_ind_data, n_iter);
      k++;
    }
  }
  return 0;
}

// In place of ngsLD.cpp:153-198.
static void ngsld_compute_all(params *pars) {
  ngsld_ctx *ctx = NULL;
  if (ngsld_create(0, &ctx) != NGSLD_OK) error(__FUNCTION__, ngsld_last_error(NULL));
  // geno_lkl is a jagged double*** in normal space at this point (ngsLD.cpp:107-114): handed over flat, copied once
  std::vector<double> flat((size_t)pars->n_sites * pars->n_ind * 3);
  for (uint64_t s = 0; s < pars->n_sites; s++)
    for (uint64_t i = 0; i < pars->n_ind; i++) memcpy(&flat[(s * pars->n_ind + i) * 3], pars->geno_lkl[s][i], 3 * sizeof(double));
  if (ngsld_set_geno_lkl(ctx, flat.data(), pars->maf, pars->n_sites, pars->n_ind, 0) != NGSLD_OK ||
      ngsld_set_replay_matrix(ctx, flat.data()) != NGSLD_OK ||  // pairs the reference's rounding decides: re-evaluated from these values
      ngsld_set_pos_dist(ctx, pars->pos_dist) != NGSLD_OK)
    error(__FUNCTION__, ngsld_last_error(ctx));
  ngsld_params lp;
  memset(&lp, 0, sizeof lp);
  lp.max_kb_dist = pars->max_kb_dist;
  lp.max_snp_dist = pars->max_snp_dist;
  lp.min_maf = pars->min_maf;
  lp.ignore_miss_data = pars->ignore_miss_data;
  lp.extend_out = pars->extend_out;
  lp.rnd_sample = pars->rnd_sample;
  lp.seed = pars->seed;
  uint64_t n_pairs = 0;
  if (ngsld_plan(ctx, &lp, &n_pairs) != NGSLD_OK) error(__FUNCTION__, ngsld_last_error(ctx));
  const char *t = getenv("NGSLD_BINDING_TEXT");
  if (!(t != NULL && strcmp(t, "0") == 0) &&
      ngsld_set_text_output(ctx, (const char *const *)pars->labels, 1) != NGSLD_OK)
    error(__FUNCTION__, ngsld_last_error(ctx));
  if (ngsld_run(ctx, 0, pars->n_sites, ngsld_binding_sink, pars) != NGSLD_OK) error(__FUNCTION__, ngsld_last_error(ctx));
  if (pars->verbose >= 1) fprintf(stderr, "==> %lu pairs computed on the device (%s)\n", (unsigned long)n_pairs, ngsld_pair_kernel(ctx));
  ngsld_destroy(ctx);
}
