// ld_common.h -- gfx950 (MI355X, CDNA4) device code of the pair-LD path: what every kernel family shares --
// the constants of the reference, the flag buffer's layout, PairArgs, the cross-lane and LDS primitives.
// (ld_device.h includes all of the pair-LD device headers in order: ld_common.h, ld_em.h, ld_kernel_multi.h,
// ld_run_pipeline.h, ld_kernel_run.h, ld_kernel_group.h, ld_kernel_stream.h, ld_dispatch.h.)
//
// One wavefront (64 lanes) owns one SNP pair; for n_ind > 512 a workgroup of 2..8 wavefronts
// shares one pair.  Replaces calc_pair_LD / haplo_freq / pair_freq_iter / pearson_r of the reference
// (ngsLD.cpp:229-367, shared/gen_func.cpp:1027-1119); see DESIGN.md for the derivation.
//
// EM step, restated for the hardware.  With a = site-1 GL triple and b = site-2 GL triple of an
// individual, P[g1][g2] = a[g1]*b[g2] (9 products, invariant over EM iterations, held in VGPRs for
// the whole pair).  The reference's 16-term `sum` (gen_func.cpp:1093-1096) is the bilinear form
// s = sum_G W[G]*P[G] with the 3x3 two-locus genotype weights W(f) (uniform per iteration), and
// its four `tmp/sum` accumulations (gen_func.cpp:1098-1104) are linear in R[G] = sum_i P_i[G]/s_i:
//   ff_k/(2x) = f_k * sum_h f_h * R[G(k,h)] / x.
// Per individual and iteration that is 9 FMA (s) + one refined reciprocal + 9 FMA (R) instead of the
// reference's ~168 flops; f64 throughout, no MFMA (nothing is shared across pairs to contract over).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ngsld.h"

namespace ngsld {

// Build-time tuning knobs (tools/build_variant.sh); everything that was measured and lost is gone from this file, its
// numbers are in DESIGN.md section 5.
#ifndef NGSLD_PRIO_S  // issue priority per stretch of an EM iteration (swept on the bench: differences of +-0.5 %)
#define NGSLD_PRIO_S 0
#define NGSLD_PRIO_TREE 3
#define NGSLD_PRIO_R 1
#define NGSLD_PRIO_SERIAL 3
#endif
#ifndef NGSLD_SETPRIO
#define NGSLD_SETPRIO 1  // issue priority raised through the serial phases of an EM iteration (round 3, same box: off = -3.5 %)
#endif
#ifndef NGSLD_MASK_SLOTS
#define NGSLD_MASK_SLOTS 6  // lockstep kernels: converged groups are masked off from this many individuals per lane on
                            // (measured: 3, 4, 5 slots lose 2.5 %, 6 gains 2 %, 7-8 gain 5.5-7 %)
#endif
constexpr int kIterMax = 100;      // ITER_MAX, gen_func.hpp:18
constexpr double kEpsilon = 1e-5;  // EPSILON,  gen_func.hpp:16

// Exact-order replay (replay.h): the kernels FLAG the pairs whose outcome the reference's own rounding decides and the
// engine re-evaluates those in the reference's operation order (on the device where the reference's input bits are to be had
// there: ld_replay.hip, ld_replay_lkl.hip; on the host otherwise).  A pair is flagged when
//   * D' or r2 is not reproducible to kRecordTol: the hap-derived allele frequencies 1 - (f0 + f1) / 1 - (f0 + f2)
//     (ngsLD.cpp:297-298) carry ~1e-16 of ABSOLUTE rounding noise in the reference and here alike, D' and r2 are quotients by
//     products of these margins q, so the two evaluations differ by ~ noise * (1 / q0 + 1 / q1) * the value itself.  The
//     noise is taken as kHapNoise = 2^-49 (1.8e-15: four times what a 39,000-case soak showed -- differences up to 1.1e-10
//     right above a then fixed threshold q >= 2^-18, i.e. 4.2e-16; with 2^-50 the round-5 soak over 10,000 un-called cases saw
//     3.1e-10 on a pair just under the bound), the tolerance as a quarter of the 1e-9 bar.  Below
//     kReplayFloor the margins themselves may be exact zeros on one side and not on the other (0/0-type quotients: nan,
//     0 or inf by the noise alone): every such pair is flagged whatever its values;
//     (rounds 2-4 flagged every pair with q < 2^-16 / 2^-18: on matrices that are not SNP-called that is 40 % of the pairs,
//     the derived bound 35 %, profiles/r05)
//   * any frequency is NaN;
//   * eps came within kTieMargin of EPSILON in some iteration (gen_func.cpp:1054: nIter could differ by one);
//   * the Pearson cross moment is ill conditioned for THIS pair (kPearsonCond): sites whose expected genotypes are nearly
//     constant -- at the extreme gsl_stats_correlation is a 0/0-type quotient of its own accumulation noise
//     (ngsLD.cpp:365-367).
constexpr double kHapNoise = 0x1p-49;
constexpr double kRecordTol = 2.5e-10;
constexpr double kReplayFloor = 0x1p-30;
constexpr double kTieMargin = 1e-12;
// r = sxy * rsx1 * rsx2 with sxy = sum e1 e2 - n mean1 mean2: the cancellation leaves ~20 ulp * n * size1 * size2 of noise
// in sxy (size = the expected genotypes' magnitude, <= 2), i.e. |delta r2| <~ 1.8e-14 * n * rsx1 * rsx2.  Pairs with
// n * rsx1 * rsx2 = 1 / (std1 * std2) above 2^13 are replayed: the bound is then 1.4e-10, a seventh of the 1e-9 bar.
// (Round 2 marked SITES -- std below 1/500 of the size -- and replayed all their pairs: low-information sites of low-depth
// data ran at the host's speed although next to an ordinary partner, std ~ 0.5, their r2_ExpG is good to 1e-11.)
constexpr double kPearsonCond = 0x1p13;
constexpr double kEpsilonTie = kEpsilon + kTieMargin;
constexpr uint32_t kTieBit = 0x80000000u;  // rides on n_iter (<= 100) from the EM loop to write_pair
// Layout of a launch's flag buffer (uint32 words): [0] count of flagged pairs, [1] count of those that are kFlagHostOnly,
// [2] pairs the device-side replay of likelihood matrices (ld_replay_lkl.hip) has settled, [3] the work counter of its
// wavefront-per-pair kernel, [4] entries of its pair list, [5] the work counter of its lane-per-pair kernel, [6] the flagged pairs its expansion left in the
// bitmap, [7] set by the called-genotype replay when it took a launch that overflowed its list (ld_replay.hip),
// [8 .. 8 + 2 cap) the record indices (uint64) of the first `cap` flagged pairs in the order their atomics landed, then the
// first kFlagHostCap kFlagHostOnly pairs once more, by themselves (what is left for the host after a device-side replay: read
// from the head that travels with the batch -- fetching a bitmap for them cost a text batch 7 ms, beside the next batch's
// pair kernel); behind this head one bit per record, and behind that bitmap (PairArgs::flags_host) a second one: the
// kFlagHostOnly pairs;
// cap = PairArgs::flag_cap, set by the engine from the launch's size (flag_cap_for).  A launch of 10^8 likelihood pairs flags
// a few dozen, one of called genotypes 26,000 (exact ties of eps with EPSILON): the host reads the head and never the bitmap.
// A list entry's top bits: kFlagHostOnly -- the pair was flagged for a reason only the host's replay settles (its r2_ExpG:
// GSL's long double recurrence) --, kFlagDone -- the device-side replay (ld_replay.hip) has already rewritten the record.
constexpr uint64_t kFlagHostOnly = 1ull << 63, kFlagDone = 1ull << 62, kFlagIndexMask = (1ull << 62) - 1;
constexpr uint32_t kFlagListAt = 8;  // first word of the list
constexpr uint32_t kFlagHostCap = 1024;  // entries of the host-only list
__host__ __device__ inline uint32_t flag_head_words(uint32_t cap) { return kFlagListAt + 2u * cap + 2u * kFlagHostCap; }

// One unit of work = ngsld_item: pairs (s1, s2_begin + c) for the bits c set in mask, records from first_record.
typedef ngsld_item Item;

// A run = up to kRunItems consecutive work items of ONE row: what one workgroup of the run kernel works through.
struct Run {
  uint32_t first_item, n_items;
};
#ifndef NGSLD_RUN_ITEMS
#define NGSLD_RUN_ITEMS 16  // build-time tuning knob: 4 / 8 / 16 / 32 measured 503 / 507 / 498 / 498 ms on the bench
#endif
constexpr uint32_t kRunItems = NGSLD_RUN_ITEMS;

struct PairArgs {
  const double *planes;  // [n_sites][3][np] normal-space normalised GLs, zero padded to np
  uint64_t site_stride;  // 3 * np
  uint32_t np;
  uint32_t n_ind;
  double inv_n;          // 1.0 / n_ind (the EM's 1/x when every individual has data)
  const double *maf;     // [n_sites] est_maf
  const double *mean_e;  // [n_sites] mean expected genotype
  const double *rsx;     // [n_sites] 1 / sqrt(sum (e - mean)^2)  (inf for a constant site)
  const Item *items;
  uint64_t n_items;
  const struct Run *runs;  // run kernel: this launch's runs (consecutive items of one row each), indices into items_all
  uint64_t n_runs;
  const Item *items_all;   // the whole plan's item array
  const double *sc4;       // [n_sites][4] packed per-site scalars {maf, mean_e, rsx, 0}: one 32-byte copy per site
  uint64_t out_base;  // global index of record 0 of the output buffers
  ngsld_rec_std *out_std;
  ngsld_rec_ext *out_ext;  // may be null
  int *status;             // set to NGSLD_ERR_MAF_RANGE when haplo_freq would error()
  // hard-called matrices (pair_ld_hard_kernel): per site four bit sets over the individuals -- genotype 0, 1, 2, no data
  const uint64_t *hard_masks;  // [n_sites][4][mask_words]
  const double *hard_u;        // [n_sites] the value of the three equal likelihoods of an individual without data
  uint32_t mask_words;         // ceil(n_ind / 64)
  // exact-order replay: flags[0] counts the flagged pairs, bit r of flags[flag_head_words(flag_cap) + r / 32] marks record r
  // of the output buffers (null: no flagging); the first flag_cap of them are also listed by record index right behind the counter
  // (flag_list(): what the host reads back is the counter and that list -- 32 KB whatever the launch's size -- and the
  // bitmap only when more pairs were flagged than the list holds)
  uint32_t *flags;
  uint32_t *flags_host;  // second bitmap (one bit per record): the flagged pairs only the host's replay settles (may be null)
  uint32_t flag_cap;   // entries of the list in flags
  uint32_t flag_text;  // also flag the pairs whose printed digits (six decimals) rounding noise could change
  uint32_t pearson_on_device;  // an ill-conditioned Pearson moment (kPearsonCond) is settled by the device-side replay of likelihood
                               // matrices (ld_replay_lkl.hip: two passes over the exact values); 0: such pairs are the host's
  uint32_t skip_degenerate;    // the pairs of a degenerate site (sc4[.][3] != 0; ld_prep.hip, site_skip_kernel) are flagged without their EM
                               // (sits in what was padding: the other members keep their offsets)
  // tiled workgroup order of the multi-wavefront kernel (launch_pair_kernel; tile_nk == 0: workgroup i takes item i):
  // rows [row0, row1) of the plan, tiles of tile_rows rows x 8 items, tile_nk tiles per row block; workgroup ids without an
  // item (row beyond row1, item index beyond the row's count) leave at once
  const uint64_t *item_off;    // device: [n_sites + 1] first item of each row (index into items_all)
  const uint64_t *h_item_off;  // the same on the host (for the launcher; never dereferenced on the device)
  uint32_t row0, row1, tile_rows, tile_nk;
  uint64_t planes_bytes;       // size of the whole planes array (launcher: is the matrix larger than the caches?)
};

// ---------------------------------------------------------------------------------------------
// cross-lane primitives
// ---------------------------------------------------------------------------------------------
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double mk_double(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }

// x86 writes the default NaN of an invalid operation (0/0, inf - inf, the x87's "real indefinite" of gsl_stats_correlation)
// with its sign bit SET -- glibc prints it "-nan", the reference's TSV is full of them -- and hands an operand's NaN on as it
// is; gfx950 generates NaNs with the bit clear.  Records carry the reference's pattern, whoever computed them (the product's
// own formatters print every NaN "-nan"; the reference's fprintf, given these records by the binding, prints the sign).
__device__ __forceinline__ double ref_nan(double v) { return v != v ? mk_double(0u, 0xfff80000u) : v; }

__device__ __forceinline__ double uniform(double v) {  // value is wave-uniform: move it to SGPRs
  return mk_double((unsigned)__builtin_amdgcn_readfirstlane(__double2loint(v)),
                   (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(v)));
}

__device__ __forceinline__ double read_lane(double v, int lane) {
  return mk_double((unsigned)__builtin_amdgcn_readlane(__double2loint(v), lane),
                   (unsigned)__builtin_amdgcn_readlane(__double2hiint(v), lane));
}

// v_permlane32_swap: lanes 32..63 of x trade places with lanes 0..31 of y.  The sum then holds
// x[l] + x[l+32] in lanes 0..31 and y[l-32] + y[l] in lanes 32..63: two values folded into one register.
__device__ __forceinline__ double fold32(double x, double y) {
  u32x2 l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  u32x2 h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return mk_double(l[0], h[0]) + mk_double(l[1], h[1]);
}

// v_permlane16_swap: odd 16-lane rows of x trade places with even rows of y.
__device__ __forceinline__ double fold16(double x, double y) {
  u32x2 l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  u32x2 h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return mk_double(l[0], h[0]) + mk_double(l[1], h[1]);
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // old = 0 with bound_ctrl: every lane has a source for the controls used here (row_ror, quad_perm), and this form
  // lets the compiler write a fresh register instead of first copying `old` into the destination
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// Sum four per-lane values over the 64 lanes with a FIXED order (deterministic per pair):
// 2 fold steps (64 -> 16 lanes, four values packed into one register, one per 16-lane row),
// 4 DPP steps inside each row, then one readlane per value.  7 f64 adds instead of 24.
__device__ __forceinline__ void wave_sum4(double &t0, double &t1, double &t2, double &t3) {
  double z01 = fold32(t0, t1);  // lanes <32: t0, lanes >=32: t1
  double z23 = fold32(t2, t3);
  double w = fold16(z01, z23);  // row0: t0, row1: t2, row2: t1, row3: t3
  w += dpp_mov<0x128>(w);       // row_ror:8
  w += dpp_mov<0x124>(w);       // row_ror:4
  w += dpp_mov<0x4E>(w);        // quad_perm:[2,3,0,1]
  w += dpp_mov<0xB1>(w);        // quad_perm:[1,0,3,2]
  t0 = read_lane(w, 0);
  t2 = read_lane(w, 16);
  t1 = read_lane(w, 32);
  t3 = read_lane(w, 48);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_rows(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);  // rows outside ROW_MASK keep old = 0
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Three values: same scheme, and the third goes into the second fold UNFOLDED -- its rows 0+1 end up in row 1, its rows
// 2+3 in row 3, and once the rows are summed one GFX9 row broadcast (lane 31 into row 3: two DPP moves and an add) joins
// the halves (folding the third value with itself first took two more v_permlane32_swap at ~14 cycles of issue each:
// -1.1 % on configs[2]).  Every lane of row 0 then holds the sum of t1, of row 2 that of t2, of row 3 that of t3.
// (The same reduction on the MATRIX pipe -- v_mfma_f64_4x4x4 with a ones / selector operand as a cross-lane adder -- was
// built and measured at -1.7 ... -2.3 %: an f64 MFMA is not free issue beside an f64 VALU stream.  DESIGN.md section 5.)
__device__ __forceinline__ double wave_sum3_rows(double t1, double t2, double t3) {
  double z12 = fold32(t1, t2);
  double w = fold16(z12, t3);  // row0: t1, row2: t2, rows 1 / 3: t3 by halves
  w += dpp_mov<0x128>(w);
  w += dpp_mov<0x124>(w);
  w += dpp_mov<0x4E>(w);
  w += dpp_mov<0xB1>(w);
  w += dpp_mov_rows<0x143, 0x8>(w);  // row_bcast:31 into row 3
  return w;
}
__device__ __forceinline__ void wave_sum3(double &t1, double &t2, double &t3) {
  // as wave_sum3_rows, but only lane 48 of the last step is ever read: the broadcast needs no defined value (no zeroing
  // moves) in the rows it does not write
  double z12 = fold32(t1, t2);
  double w = fold16(z12, t3);
  w += dpp_mov<0x128>(w);
  w += dpp_mov<0x124>(w);
  w += dpp_mov<0x4E>(w);
  w += dpp_mov<0xB1>(w);
  t1 = read_lane(w, 0);
  t2 = read_lane(w, 32);
  int ulo, uhi;
  asm("; undefined" : "=v"(ulo), "=v"(uhi));
  const int lo = __builtin_amdgcn_update_dpp(ulo, __double2loint(w), 0x143, 0x8, 0xf, false);  // row_bcast:31 into row 3
  const int hi = __builtin_amdgcn_update_dpp(uhi, __double2hiint(w), 0x143, 0x8, 0xf, false);
  t3 = read_lane(w + __hiloint2double(hi, lo), 48);
}

// One value, no permlane swaps (each costs ~14 cycles of issue): four DPP levels inside the rows, then the GFX9 row
// broadcasts -- lane 15 of rows 0 / 2 into rows 1 / 3, lane 31 into rows 2 and 3 -- leave the total in row 3.
__device__ __forceinline__ double wave_sum1_bcast(double v) {
  v += dpp_mov<0xB1>(v);   // quad_perm:[1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm:[2,3,0,1]
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x128>(v);  // row_ror:8
  v += dpp_mov_rows<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_mov_rows<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return read_lane(v, 63);
}

__device__ __forceinline__ double wave_sum1(double v) {
  double a = v, b = 0.0, c = 0.0, d = 0.0;
  wave_sum4(a, b, c, d);
  return a;
}

// 1/s to 0.5 ulp: v_rcp_f64 seed (measured 2^-24.4 on gfx950, tools/probe_rcp.hip) + ONE cubic step
// r0*(1 + e + e^2), e = 1 - s*r0, which leaves e^3 ~ 2^-73: same accuracy as two Newton steps for one FMA
// less.  s == 0 gives NaN (inf * 0), which is what the caller wants: the reference's tmp/sum is 0/0 there
// (gen_func.cpp:1103).
__device__ __forceinline__ double rcp_refined(double s) {
  const double r0 = __builtin_amdgcn_rcp(s);
  const double e = fma(-s, r0, 1.0);
  const double t = fma(e, e, e);
  return fma(r0, t, r0);
}

// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() would also
// wait vmcnt(0), i.e. drain an asynchronous global->LDS site copy that is meant to fly through the whole EM loop.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// LDS accesses of the exchange between the wavefronts of a pair, written as assembly.  The compiler orders every LDS access
// it can see behind an asynchronous global->LDS copy in flight (s_waitcnt vmcnt(0) in front of the first ds instruction
// after a global_load_lds: it cannot tell that the exchange buffer and the copy's target are different bytes) -- and the
// slice of the NEXT pair is meant to fly through the whole EM loop of this one.  These it does not see; the waiting is done
// here: lds_barrier() drains the stores, lds_gather ends with its own s_waitcnt.
typedef double dbl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_addr(const void *p) {  // LDS byte address of a pointer into __shared__ memory
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}
__device__ __forceinline__ void lds_post(uint32_t addr, double v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_post2(uint32_t addr, double a, double b) {
  dbl2 v = {a, b};
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// N consecutive 16-byte pieces from addr (the same address in every lane: broadcast reads), all in flight together, ONE
// wait -- written out by the compiler the reads of the partial sums came one LDS round trip after the other, each behind
// the add that consumed the previous one.
template <int N>
__device__ __forceinline__ void lds_gather(uint32_t addr, dbl2 (&q)[N]) {
  static_assert(N == 2 || N == 3 || N == 4 || N == 6 || N == 8 || N == 12, "lds_gather: unsupported count");
  if constexpr (N == 2)
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(q[0]), "=&v"(q[1]) : "v"(addr) : "memory");
  else if constexpr (N == 3)
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]) : "v"(addr) : "memory");
  else if constexpr (N == 4)
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                 "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(addr) : "memory");
  else if constexpr (N == 6)
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\t"
                 "ds_read_b128 %3, %6 offset:48\n\tds_read_b128 %4, %6 offset:64\n\tds_read_b128 %5, %6 offset:80\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]) : "v"(addr) : "memory");
  else if constexpr (N == 8)
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\t"
                 "ds_read_b128 %3, %8 offset:48\n\tds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\t"
                 "ds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
                 : "v"(addr) : "memory");
  else
    asm volatile("ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:16\n\tds_read_b128 %2, %12 offset:32\n\t"
                 "ds_read_b128 %3, %12 offset:48\n\tds_read_b128 %4, %12 offset:64\n\tds_read_b128 %5, %12 offset:80\n\t"
                 "ds_read_b128 %6, %12 offset:96\n\tds_read_b128 %7, %12 offset:112\n\tds_read_b128 %8, %12 offset:128\n\t"
                 "ds_read_b128 %9, %12 offset:144\n\tds_read_b128 %10, %12 offset:160\n\tds_read_b128 %11, %12 offset:176\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7]),
                   "=&v"(q[8]), "=&v"(q[9]), "=&v"(q[10]), "=&v"(q[11])
                 : "v"(addr) : "memory");
}

// gen_func.cpp:862-868 miss_data with the reference's abs() macro semantics
__device__ __forceinline__ bool miss_data(double g0, double g1, double g2) {
  double d01 = g0 - g1, d12 = g1 - g2;
  d01 = d01 >= 0 ? d01 : -d01;
  d12 = d12 >= 0 ? d12 : -d12;
  return d01 < kEpsilon && d12 < kEpsilon;
}

}  // namespace ngsld
