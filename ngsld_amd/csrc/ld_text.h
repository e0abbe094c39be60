// ld_text.h -- launch interface of the device-side TSV formatter (see ld_text.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ngsld.h"

namespace ngsld {

struct TextArgs {
  const ngsld_item *items;  // the batch's items (row by row); first_record is a global record index
  uint64_t n_items;
  uint64_t out_base;        // global index of the batch's record 0
  uint64_t n_pairs;         // records in the batch
  const ngsld_rec_std *std_rec;
  const ngsld_rec_ext *ext_rec;  // null without --extend_out
  const double *maf;        // [n_sites]
  const double *cum;        // [n_sites] running sum of the finite pos_dist entries up to and including the site
  const uint32_t *infc;     // [n_sites] number of INFINITY entries (chromosome changes) up to and including the site
  const char *labels;       // label bytes, back to back (no terminators); null = every label is "(null)"
  const uint64_t *label_off;  // [n_sites + 1]
  uint64_t *lens;           // [n_pairs] out (length pass) / in (write pass)
  const uint64_t *offs;     // [n_pairs] exclusive prefix sums of lens (write pass)
  char *text;               // write pass: the batch's text
  uint64_t text_cap;        // ... and its capacity in bytes (0: not checked); a row that would end beyond it is not written and
  uint64_t *overflow;       // *overflow is set (the write pass issued before the host knows the batch's length: engine_run.hip)
  int *needs_host;          // set to 1 when a value is outside the device formatter's range
};

// pass 1: the byte length of every row; pass 2 (after the prefix sums): the rows themselves
hipError_t launch_text_lengths(const TextArgs &a, hipStream_t stream);
hipError_t launch_text_write(const TextArgs &a, hipStream_t stream);
// the lengths of n rows again (batch-relative record indices rec[], their sites s1[] / s2[], all device arrays): lens[] is
// updated, *changed (device) set to 1 when any differs from what the length pass found
hipError_t launch_text_relength(const TextArgs &a, const uint64_t *rec, const uint32_t *s1, const uint32_t *s2, uint64_t n,
                                uint64_t *changed, hipStream_t stream);
// exclusive prefix sums of lens -> offs; total (one uint64, device) receives the batch's text length
size_t text_scan_temp_bytes(uint64_t n);
hipError_t text_scan(void *temp, size_t temp_bytes, const uint64_t *lens, uint64_t *offs, uint64_t n, uint64_t *total,
                     hipStream_t stream);

}  // namespace ngsld
