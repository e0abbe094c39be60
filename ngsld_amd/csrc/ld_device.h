// ld_device.h -- gfx950 (MI355X, CDNA4) device code of the pair-LD path.
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

// ---------------------------------------------------------------------------------------------
// Building blocks of the pair kernels
// ---------------------------------------------------------------------------------------------
// Reciprocals of N positive numbers from ONE reciprocal: products up a binary tree (N - 1 multiplies), 1/root, then
// down again -- the inverse of a node is the parent's inverse times the sibling's product (2 multiplies per inner
// node).  N = 8: 21 multiplies + one refined v_rcp_f64 instead of 8 (or 4, taken in pairs) of the 16-cycle kind.
template <int N>
struct RcpTree {
  static constexpr int L = N / 2;
  static __device__ __forceinline__ double prod(const double *s) {
    return RcpTree<L>::prod(s) * RcpTree<N - L>::prod(s + L);
  }
  static __device__ __forceinline__ void down(const double *s, double inv, double *r) {
    const double pl = RcpTree<L>::prod(s), pr = RcpTree<N - L>::prod(s + L);  // same expressions as in prod(): CSE'd
    RcpTree<L>::down(s, inv * pr, r);
    RcpTree<N - L>::down(s + L, inv * pl, r + L);
  }
};
template <>
struct RcpTree<1> {
  static __device__ __forceinline__ double prod(const double *s) { return s[0]; }
  static __device__ __forceinline__ void down(const double *, double inv, double *r) { r[0] = inv; }
};

struct PairedTag { static constexpr bool value = true; };   // compile-time selectors of em_pair's reciprocal scheme
struct SingleTag { static constexpr bool value = false; };
typedef __attribute__((address_space(3))) void lds_void_t;        // operands of __builtin_amdgcn_global_load_lds
typedef const __attribute__((address_space(1))) void glb_void_t;

// Allele relabelling.  The frequency recovered from the other three carries an ABSOLUTE error of ~1e-16.
// That is harmless for the largest of the four and ruinous for a tiny one: with both sites nearly monomorphic the
// denominators of D' and r2 are products of two small margins (1e-14, say), and 1e-16 in a hap00 of 1e-15 moved D' in
// the third decimal.  So each site's alleles are labelled such that its estimated frequency is <= 1/2 -- a site with
// maf > 1/2 has its genotype planes 0 and 2 read in each other's place -- which makes hap 0 (initially (1-m1)(1-m2) >=
// 1/4) the common-common haplotype, the one that tends to 1 exactly where the conditioning is bad.  The EM is equivariant
// under the relabelling; the frequencies are put back in the caller's order afterwards (k = 2 * allele1 + allele2).
struct Relabel {
  bool flip1, flip2;
  double m1, m2, mean1, mean2;  // frequencies and mean expected genotypes under the new labels
};
__device__ __forceinline__ Relabel relabel(double m1, double m2, double mean1, double mean2) {
  Relabel r;
  r.flip1 = m1 > 0.5;
  r.flip2 = m2 > 0.5;
  r.m1 = r.flip1 ? 1.0 - m1 : m1;
  r.m2 = r.flip2 ? 1.0 - m2 : m2;
  r.mean1 = r.flip1 ? 2.0 - mean1 : mean1;  // expected genotype p1 + 2 p2 of a normalised triple becomes 2 - e
  r.mean2 = r.flip2 ? 2.0 - mean2 : mean2;
  return r;
}
__device__ __forceinline__ void unrelabel(bool flip1, bool flip2, double &f0, double &f1, double &f2, double &f3) {
  if (flip1) {  // allele at site 1: haplotypes k <-> k ^ 2
    double t = f0; f0 = f2; f2 = t;
    t = f1; f1 = f3; f3 = t;
  }
  if (flip2) {  // allele at site 2: k <-> k ^ 1
    double t = f0; f0 = f1; f1 = t;
    t = f2; f2 = f3; f3 = t;
  }
}

// Stage both sites of one pair: P = a (x) b for this lane's SLOTS individuals, their validity bits and the
// Pearson cross moment.  pa / pb point at a site's three planes [3][np] -- in HBM/L2 (direct kernel) or in
// LDS (prefetch kernel); after inlining the compiler knows which and emits global_load or ds_read.
//   UNCENTRED: sxy comes back as the uncentred cross moment sum e1 e2 -- padding lanes hold a == b == 0, so no bounds
//   test -- and the caller subtracts n * mean1 * mean2 once per pair
//   GHOSTS.  A slot that holds no individual -- a padding lane, or under --ignore_miss_data an individual without data at
//   either site -- is staged as P = (1, 0, ..., 0).  In the hot EM step (shared reciprocal, three-value form) such a slot has
//   s = f0^2 -- positive, at least 2^-20 while the pair is in that loop -- so it neither zeroes the lane's product tree nor
//   overflows its reciprocal, and it adds nothing to R[1..8]: r * 0.  R[0] is never accumulated in that form.  The steps
//   that do accumulate R[0] take one reciprocal per individual and skip the slot by its validity bit.  (Rounds 1-2 kept
//   P = 0 and added a per-slot `pad` of 0 / 1 to s: two registers per slot in every kernel that may hold empty slots
//   anywhere -- all of --ignore_miss_data -- which is what spilled there.)
__device__ __forceinline__ const double *uniform_ptr(const double *p) {  // a wavefront-uniform pointer, said so: SGPRs
  const uint64_t v = (uint64_t)(uintptr_t)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return reinterpret_cast<const double *>((uintptr_t)(((uint64_t)hi << 32) | lo));
}

//   A_GLOBAL: pa points into global memory (the multi-wavefront kernels read their slice of the row vector from L2 for
//   every pair): the three plane bases are handed to the loads as SGPR pairs + one 32-bit lane offset.  Left to itself the
//   compiler kept ~10 64-bit VGPR addresses for them, and under --ignore_miss_data, where registers are tightest, SPILLED
//   them -- seven scratch reloads per pair, one after the other into the same register pair, each an L2 round trip in
//   front of the load it feeds: 4.7 us of a 15 us pair at n_ind 2,000
template <int SLOTS, bool MASKED, bool ONLY_LAST = false, bool UNCENTRED = false, bool A_GLOBAL = false>  // ONLY_LAST: only the last slot can hold padding lanes
__device__ __forceinline__ void stage_pair(const double *pa, uint32_t npa, uint32_t ia0, const double *pb, uint32_t npb,
                                           uint32_t ib0, uint32_t ind0, uint32_t n_ind, double mean1, double mean2,
                                           double (&P)[SLOTS][9], uint32_t &vbits, double &sxy, bool flip_a = false,
                                           bool flip_b = false, const double (*a_regs)[3] = nullptr, int n_a_regs = 0) {
  // a_regs (may be null): this lane's first n_a_regs triples of site 1, already relabelled, held in registers by the caller
  // for all the pairs of an item (the row vector is the same for every one of them) -- pa is not read for those slots
  // pa[g * npa + ia0 + 64 j] / pb[g * npb + ib0 + 64 j] hold genotype g of individual ind0 + 64 j (this lane, slot j);
  // flip_a / flip_b (wavefront-uniform) relabel the alleles of a site: genotype planes 0 and 2 trade places (see Relabel)
  vbits = 0;
  sxy = 0.0;
  const double *pa0 = pa + (flip_a ? 2 * npa : 0u), *pa1 = pa + npa, *pa2 = pa + (flip_a ? 0u : 2 * npa);
  const double *pb0 = pb + (flip_b ? 2 * npb : 0u), *pb2 = pb + (flip_b ? 0u : 2 * npb);
  if (A_GLOBAL) {
    pa0 = uniform_ptr(pa0); pa1 = uniform_ptr(pa1); pa2 = uniform_ptr(pa2);
  }
  // kByCount (several wavefronts per pair, every individual counts): a wavefront's slots are full up to a wavefront-
  // uniform slot n_full, at most ONE slot is partly filled (lanes below rem), the rest are empty -- validity bits and ghosts
  // come from those two numbers instead of a compare and a select per slot: inside the loop below these held nine compare
  // masks and select temporaries beside the loads in flight and cost 64 bytes of scratch per lane (2 x 9 slots: -14 % pairs/s)
  constexpr bool kByCount = !MASKED && !ONLY_LAST;
  const uint32_t lane_in_wave = ind0 & 63u;
  uint32_t n_full = 0, rem = 0;
  if (kByCount) {
    const uint32_t first = ind0 - lane_in_wave;  // this wavefront's first individual
    const uint32_t have = n_ind > first ? n_ind - first : 0u;
    n_full = (uint32_t)__builtin_amdgcn_readfirstlane((int)(have >> 6 < (uint32_t)SLOTS ? have >> 6 : (uint32_t)SLOTS));
    rem = (uint32_t)__builtin_amdgcn_readfirstlane((int)(have >> 6 < (uint32_t)SLOTS ? have & 63u : 0u));
    vbits = ((1u << n_full) - 1u) | ((lane_in_wave < rem ? 1u : 0u) << n_full);
  }
#pragma unroll
  for (int j = 0; j < SLOTS; ++j) {
    const uint32_t ia = ia0 + (uint32_t)j * 64, ib = ib0 + (uint32_t)j * 64;
    typedef const __attribute__((address_space(1))) double gdouble_t;  // (said to be global memory: global_load, not flat_load)
    const bool in_regs = a_regs != nullptr && j < n_a_regs;
    const double a0 = in_regs ? a_regs[j < n_a_regs ? j : 0][0] : (A_GLOBAL ? ((gdouble_t *)pa0)[ia] : pa0[ia]),
                 a1 = in_regs ? a_regs[j < n_a_regs ? j : 0][1] : (A_GLOBAL ? ((gdouble_t *)pa1)[ia] : pa1[ia]),
                 a2 = in_regs ? a_regs[j < n_a_regs ? j : 0][2] : (A_GLOBAL ? ((gdouble_t *)pa2)[ia] : pa2[ia]);
    const double b0 = pb0[ib], b1 = pb[npb + ib], b2 = pb2[ib];
    const bool inb = kByCount ? ((vbits >> j) & 1u) != 0
                              : ((ONLY_LAST && j < SLOTS - 1) ? true : ind0 + (uint32_t)j * 64 < n_ind);
    if (!kByCount) {
      bool ok = inb;
      if (MASKED) ok = inb && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);  // gen_func.cpp:1089
      vbits |= (ok ? 1u : 0u) << j;
    }
    double z0 = a0, z1 = a1, z2 = a2;
    if (MASKED) {  // an individual without data: P = (1, 0, ..., 0)
      const double keep = ((vbits >> j) & 1u) ? 1.0 : 0.0;
      z0 = a0 * keep; z1 = a1 * keep; z2 = a2 * keep;
      P[j][0] = fma(z0, b0, 1.0 - keep);
    } else if (ONLY_LAST && j == SLOTS - 1) {  // padding lanes hold zeros in the planes already
      P[j][0] = fma(a0, b0, inb ? 0.0 : 1.0);
    } else {                                   // (kByCount: the ghosts are put in after the loop)
      P[j][0] = a0 * b0;
    }
    P[j][1] = z0 * b1; P[j][2] = z0 * b2;
    P[j][3] = z1 * b0; P[j][4] = z1 * b1; P[j][5] = z1 * b2;
    P[j][6] = z2 * b0; P[j][7] = z2 * b1; P[j][8] = z2 * b2;
    // expected genotypes p1 + 2*p2 (ngsLD.cpp:113); pearson_r runs over ALL individuals (ngsLD.cpp:290)
    if (UNCENTRED && !MASKED) {  // (a1 + 2 a2)(b1 + 2 b2) = P4 + 2 P5 + 2 P7 + 4 P8 (measured 0.3 % faster than from a, b)
      sxy += fma(4.0, P[j][8], fma(2.0, P[j][5] + P[j][7], P[j][4]));
    } else if (UNCENTRED) {      // P of individuals without data is zeroed: take the moment from a and b
      sxy = fma(fma(2.0, a2, a1), fma(2.0, b2, b1), sxy);
    } else {
      const double c1 = inb ? fma(2.0, a2, a1) - mean1 : 0.0;
      const double c2 = inb ? fma(2.0, b2, b1) - mean2 : 0.0;
      sxy = fma(c1, c2, sxy);
    }
  }
  if (kByCount) {  // ghosts behind scalar branches: full wavefronts -- all but a pair's last -- skip every one of them
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      if (n_full <= (uint32_t)j) {
        const bool keep = n_full == (uint32_t)j && lane_in_wave < rem;
        P[j][0] = keep ? P[j][0] : 1.0;
      }
    }
  }
}

// x = individuals with data (gen_func.cpp:1091): popcount of ballots, integer exact
template <int SLOTS>
__device__ __forceinline__ uint32_t count_valid(uint32_t vbits) {
  uint32_t x = 0;
#pragma unroll
  for (int j = 0; j < SLOTS; ++j) x += (uint32_t)__popcll(__ballot((vbits >> j) & 1u));
  return x;
}

// haplo_freq (gen_func.cpp:1027-1059) on the staged pair.  Returns n_iter; f0..f3 hold hap_freq on exit.
//   vbits:     bit j = slot j of this lane holds an individual that counts (not a ghost, see stage_pair)
//   WAVES > 1: the pair is spread over WAVES wavefronts, partial sums meet in xch (LDS, double buffered)
//   xpar:      (WAVES > 1) the caller's count of exchanges so far: its parity picks the half of xch an exchange uses.  Carried
//              from pair to pair, consecutive exchanges alternate whatever the iteration counts were -- no barrier is needed
//              between the last exchange of one pair and the first of the next
// Reciprocals.  ALL slots of a lane share one v_rcp_f64 (RcpTree); ghost slots take part with s = f0^2.  s lies in (0, 1];
// the product of SLOTS values can underflow (all below ~1e-38 for eight slots), and that -- like any other non-finite
// outcome -- is caught by the sanity test on the new frequencies, after which the iteration is redone with one reciprocal
// per individual before anything is concluded from it.  (One reciprocal per individual, and one per two individuals, were
// the earlier forms: -9 % and -4 % against the tree at eight slots.)
template <int SLOTS, int WAVES>
__device__ __forceinline__ uint32_t em_pair(const double (&P)[SLOTS][9], uint32_t vbits, double inv_x, double m1,
                                            double m2, double &f0, double &f1, double &f2, double &f3,
                                            double (*xch)[WAVES][4], int sub, int lane, int *status,
                                            uint32_t *xpar = nullptr) {
  static_assert(WAVES == 1 || WAVES == 2 || WAVES == 4 || WAVES == 8, "em_pair: 1, 2, 4 or 8 wavefronts per pair");
  f0 = (1 - m1) * (1 - m2); f1 = (1 - m1) * m2; f2 = m1 * (1 - m2); f3 = m1 * m2;  // gen_func.cpp:1034-1037
  if (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1) {  // error() in the reference (:1030); reported through status
    if (lane == 0 && sub == 0) atomicExch(status, (int)NGSLD_ERR_MAF_RANGE);
    f0 = f1 = f2 = f3 = __builtin_nan("");
  }
  // f = ff/(2x) (gen_func.cpp:1108-1109).  The renormalisation that follows there (:1112-1113) divides
  // by sum_k ff_k/(2x) = (1/x) sum_i s_i/s_i = 1 up to rounding, and the EM map does not depend on the
  // scale of f, so it is not repeated per iteration.  inv_x = 1/x; x == 0 gives 0 * inf = NaN like the reference's 0/0.
  // (held in a VGPR: the four products t_k * inv_x below take t_k from SGPRs, and a VALU op reads one SGPR at most)
  asm("" : "+v"(inv_x));
  bool bad = false, tie = false;
  uint32_t n_iter = 0;
  constexpr bool kTree = SLOTS > 1;
  constexpr bool kScaled = WAVES == 1;  // (several wavefronts per pair: partial sums are scaled after they met)
  // tree_tag: the step with the shared reciprocal, or with one reciprocal per individual.  drop_tag: the step in its
  // three-value form (hap 0 recovered from the sum) or in the full four-value form.  The shared-reciprocal step only exists
  // in the three-value form; the step with one reciprocal per individual, which only ever runs outside the hot loop, in
  // the full form -- and, where several wavefronts share a pair, in the three-value form too (all of them have to
  // exchange the same values, and some take this step in every iteration).
  auto em_step = [&](auto tree_tag, auto drop_tag, double &n0, double &n1, double &n2, double &n3) {
    constexpr bool kShared = decltype(tree_tag)::value;
    constexpr bool kDrop = decltype(drop_tag)::value;
    static_assert(!kShared || (kTree && kDrop), "the shared-reciprocal step: several slots, three-value form");
    // products f_k f_h: they build the two-locus genotype weights W (s = sum_G W[G] P[G] is the
    // reference's 16-term `sum`, gen_func.cpp:1093-1096) and are reused by the t_k contraction below
    const double p00 = f0 * f0, p01 = f0 * f1, p02 = f0 * f2, p03 = f0 * f3, p11 = f1 * f1;
    const double p12 = f1 * f2, p13 = f1 * f3, p22 = f2 * f2, p23 = f2 * f3, p33 = f3 * f3;
    const double w1 = p01 + p01, w3 = p02 + p02, w4 = 2.0 * (p03 + p12), w5 = p13 + p13, w7 = p23 + p23;
    double R0 = 0, R1 = 0, R2 = 0, R3 = 0, R4 = 0, R5 = 0, R6 = 0, R7 = 0, R8 = 0;
    if (NGSLD_SETPRIO && kShared) __builtin_amdgcn_s_setprio(NGSLD_PRIO_S);  // the dense s sums start here
    auto slot_s = [&](int j) -> double {
      double s = p00 * P[j][0];
      s = fma(w1, P[j][1], s); s = fma(p11, P[j][2], s);
      s = fma(w3, P[j][3], s); s = fma(w4, P[j][4], s); s = fma(w5, P[j][5], s);
      s = fma(p22, P[j][6], s); s = fma(w7, P[j][7], s); s = fma(p33, P[j][8], s);
      return s;
    };
    auto slot_acc = [&](int j, double r) {
      if (!kDrop) R0 = fma(P[j][0], r, R0);
      R1 = fma(P[j][1], r, R1); R2 = fma(P[j][2], r, R2);
      R3 = fma(P[j][3], r, R3); R4 = fma(P[j][4], r, R4); R5 = fma(P[j][5], r, R5);
      R6 = fma(P[j][6], r, R6); R7 = fma(P[j][7], r, R7); R8 = fma(P[j][8], r, R8);
    };
    if constexpr (kShared) {
      double sv[SLOTS], rv[SLOTS];
#pragma unroll
      for (int j = 0; j < SLOTS; ++j) sv[j] = slot_s(j);
      // Two wavefronts share a SIMD.  The one inside a serial stretch of its iteration (reciprocal tree; contraction,
      // reduction, convergence test and the next f products) has one instruction ready at a time and every cycle it
      // waits for the issue slot lengthens its critical path; the one inside a dense stretch (the s and R sums) has
      // dozens ready.  Priority goes to the former.
      if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(NGSLD_PRIO_TREE);
      // kScaled: 1/x rides on the root inverse, so every R -- and with them the three t_k -- come out divided by x
      double inv = rcp_refined(RcpTree<SLOTS>::prod(sv));
      if (kScaled) inv *= inv_x;
      RcpTree<SLOTS>::down(sv, inv, rv);
      if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(NGSLD_PRIO_R);
#pragma unroll
      for (int j = 0; j < SLOTS; ++j) slot_acc(j, rv[j]);
      if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(NGSLD_PRIO_SERIAL);
    } else {
#pragma unroll
      for (int j = 0; j < SLOTS; ++j) {
        if ((vbits >> j) & 1u) slot_acc(j, rcp_refined(slot_s(j)));  // (ghost slots are skipped: this form accumulates R[0])
      }
    }
    // t_k = sum_h f_k f_h R[G(k,h)]  (= this lane's share of ff_k / 2, gen_func.cpp:1098-1104)
    double t0 = kDrop ? 0.0 : fma(p03, R4, fma(p02, R3, fma(p01, R1, p00 * R0)));
    double t1 = fma(p13, R5, fma(p12, R4, fma(p11, R2, p01 * R1)));
    double t2 = fma(p23, R7, fma(p22, R6, fma(p12, R4, p02 * R3)));
    double t3 = fma(p33, R8, fma(p23, R7, fma(p13, R5, p03 * R4)));
    if (WAVES > 1 && kDrop) {
      // several wavefronts per pair, three-value form: the row totals go to the exchange buffer from the lanes that hold
      // them (no v_readlane, no copies back to VGPRs), and every wavefront adds the partials up in the same order -- the
      // new frequencies must be the same bit pattern in all of them, they decide together when to leave the loop.
      // The LDS accesses are assembly (lds_post / lds_gather): the compiler must not order them behind the slice copy in flight.
      const double w = wave_sum3_rows(t1, t2, t3);
      const int par = (int)((*xpar)++ & 1u);
      const int row = lane >> 4;
      // layout of one parity's buffer (WAVES * 32 bytes): [value k = 0..2][wavefront] -- a value's partials side
      // by side, WAVES / 2 reads of 16 bytes each; added up in the order of the wavefronts
      const uint32_t base = lds_addr(&xch[par][0][0]);
      if ((lane & 15) == 0 && row != 1)  // rows 0 / 2 / 3 hold t1 / t2 / t3 (wave_sum3_rows)
        lds_post(base + (uint32_t)((row == 0 ? 0 : row - 1) * WAVES + sub) * 8u, w);
      lds_barrier();
      if constexpr (WAVES == 8) {
        // 24 partials: lane l < 24 reads partial l (value l / 8 of wavefront l % 8), three DPP steps add the eight of a value
        // inside their eight lanes -- a fixed tree, the same in every wavefront -- and lanes 0 / 8 / 16 hand the totals out
        // (12 reads of 16 bytes per lane -- 48 registers of partials in flight -- lost 4.5 % at n_ind 4000)
        double v;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(base + (uint32_t)(lane & 31) * 8u) : "memory");
        v += dpp_mov<0xB1>(v);   // quad_perm:[1,0,3,2]
        v += dpp_mov<0x4E>(v);   // quad_perm:[2,3,0,1]
        v += dpp_mov<0x141>(v);  // row_half_mirror: lane l <-> 7 - l inside each 8 lanes
        t1 = read_lane(v, 0); t2 = read_lane(v, 8); t3 = read_lane(v, 16);
      } else {
        constexpr int kHalf = WAVES > 1 ? WAVES / 2 : 1;  // (one wavefront per pair: instantiated, never run)
        dbl2 q[3 * kHalf];
        lds_gather<3 * kHalf>(base, q);
        t1 = q[0][0] + q[0][1]; t2 = q[kHalf][0] + q[kHalf][1]; t3 = q[2 * kHalf][0] + q[2 * kHalf][1];
#pragma unroll
        for (int v = 1; v < kHalf; ++v) {
          t1 += q[v][0]; t2 += q[kHalf + v][0]; t3 += q[2 * kHalf + v][0];
          t1 += q[v][1]; t2 += q[kHalf + v][1]; t3 += q[2 * kHalf + v][1];
        }
      }
    } else {
      if (kDrop)
        wave_sum3(t1, t2, t3);
      else
        wave_sum4(t0, t1, t2, t3);
      if (WAVES > 1) {  // (the full four-value form of a pair spread over several wavefronts: rare, plain LDS accesses)
        const int par = (int)((*xpar)++ & 1u);
        if (lane == 0) {
          xch[par][sub][0] = t0; xch[par][sub][1] = t1; xch[par][sub][2] = t2; xch[par][sub][3] = t3;
        }
        lds_barrier();
        t0 = t1 = t2 = t3 = 0.0;
        for (int w = 0; w < WAVES; ++w) {
          t0 += xch[par][w][0]; t1 += xch[par][w][1]; t2 += xch[par][w][2]; t3 += xch[par][w][3];
        }
      }
    }
    const bool scaled = kShared && kScaled;
    n1 = scaled ? t1 : t1 * inv_x; n2 = scaled ? t2 : t2 * inv_x; n3 = scaled ? t3 : t3 * inv_x;
    // sum_k ff_k / (2x) = 1 (every individual's four posterior weights add up to one): in the three-value form the first
    // frequency is what the other three leave, R[0] is never accumulated and three values go through the reduction
    // instead of four
    n0 = kDrop ? 1.0 - ((n1 + n2) + n3) : t0 * inv_x;
  };
  // Any individual with s == 0 makes every tmp/sum NaN in the reference, hence all four f NaN and, as a
  // NaN difference never raises eps (gen_func.cpp:1049-1053), "convergence" at this iteration.  Here
  // s == 0 poisons every R with inf/NaN -- fma(P, NaN, R) -- so any one accumulated frequency that is not a sane
  // value below 2 <=> the reference is all NaN.
  // The hot loops hold the one-reciprocal step and nothing else: a step that does not look sane leaves its loop, is
  // redone with one reciprocal per individual (an underflowed product has to be ruled out before anything is
  // concluded), and the loop is entered again -- with a single definition of the new frequencies per trip the compiler
  // carries them from one iteration to the next without register copies.
  // Two forms of the step.  The three-value form leaves hap 0 with an ABSOLUTE error of ~1e-16, which is nothing while
  // hap 0 is a sizeable frequency (the allele relabelling makes it the common-common haplotype) and too much once it is
  // tiny: its update is multiplicative, so a relative error stays for good, and the denominators of D' and r2 can be
  // products of two small margins.  Below kFullBelow the pair therefore leaves the hot loop for good and finishes in
  // the full four-value form (one reciprocal per individual: slower, and rare): hap 0 then keeps the relative accuracy
  // it had at the switch (1e-16 / kFullBelow ~ 1e-13).
  constexpr double kFullBelow = 0x1p-10;
  bool full = __builtin_amdgcn_ballot_w64(f0 < kFullBelow) != 0;  // wave-uniform (f is)
  bool done = false;
  while (!done && n_iter < (uint32_t)kIterMax) {
    if constexpr (kTree) {
      if (!full) {
        for (; n_iter < (uint32_t)kIterMax; ++n_iter) {
          double n0, n1, n2, n3;
          em_step(PairedTag(), PairedTag(), n0, n1, n2, n3);  // (the tags double as true / false)
          if (__builtin_amdgcn_ballot_w64(!(n1 < 2.0))) break;  // an odd step (wave-uniform values: all-or-nothing)
          // eps = the largest of the four changes (gen_func.cpp:1049-1053) is at least the change of hap 1: while that one
          // alone is above EPSILON -- nine iterations in ten -- the other three differences are not formed (+1.2 %)
          bool conv = false;
          if (__builtin_amdgcn_ballot_w64(fabs(n1 - f1) < kEpsilonTie)) {
            const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
            conv = __builtin_amdgcn_ballot_w64(eps < kEpsilon) != 0;  // gen_func.cpp:1054-1055
            tie |= __builtin_amdgcn_ballot_w64(fabs(eps - kEpsilon) < kTieMargin) != 0;  // too close to call: replayed
          }
          f0 = n0; f1 = n1; f2 = n2; f3 = n3;
          if (conv) {
            done = true;
            break;
          }
          if (__builtin_amdgcn_ballot_w64(n0 < kFullBelow)) {
            full = true;
            ++n_iter;  // this iteration is complete
            break;
          }
        }
        if (done || n_iter >= (uint32_t)kIterMax) break;
        if (full) continue;
        // an odd step: on to the second opinion
        if (WAVES > 1) lds_barrier();  // n1 is the same in every wavefront: all redo, none still reads the exchange buffer
      }
    }
    // one iteration with one reciprocal per individual, four-value form: the kernels' only path where any slot may be
    // empty, the second opinion on an odd step, and how a pair with a tiny hap 0 finishes
    double n0, n1, n2, n3;
    if (WAVES == 1 || full)
      em_step(SingleTag(), SingleTag(), n0, n1, n2, n3);
    else
      em_step(SingleTag(), PairedTag(), n0, n1, n2, n3);
    if (__builtin_amdgcn_ballot_w64(!(n1 < 2.0))) {
      bad = true;
      break;
    }
    const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
    f0 = n0; f1 = n1; f2 = n2; f3 = n3;
    tie |= __builtin_amdgcn_ballot_w64(fabs(eps - kEpsilon) < kTieMargin) != 0;
    if (__builtin_amdgcn_ballot_w64(eps < kEpsilon)) break;
    ++n_iter;
    if (!full && __builtin_amdgcn_ballot_w64(f0 < kFullBelow)) full = true;  // (wavefronts without a hot loop)
  }
  if (bad) f0 = f1 = f2 = f3 = __builtin_nan("");
  return n_iter | (tie ? kTieBit : 0u);
}

// Is v closer than d to a point where "%f" (six decimals) rounds the other way?  NaN / inf: no.
__device__ __forceinline__ bool near_rounding(double v, double d) {
  const double t = fabs(v) * 1e6;
  return fabs((t - floor(t)) - 0.5) < d * 1e6;
}

// ngsLD.cpp:296-306 (hap-derived maf, D, D', r2) + pearson_r, one record per pair; pairs whose outcome the reference's
// rounding decides are flagged for the exact-order replay (see kHapNoise / kReplayFloor above).
__device__ __forceinline__ void write_pair(const PairArgs &A, uint64_t slot, double f0, double f1, double f2,
                                           double f3, double sxy, double rsx1, double rsx2, uint32_t x,
                                           uint32_t n_iter) {
  const bool tie = (n_iter & kTieBit) != 0;
  n_iter &= ~kTieBit;
  const double hm0 = 1 - (f0 + f1);
  const double hm1 = 1 - (f0 + f2);
  const double D = f0 * f3 - f1 * f2;
  const double q00 = hm0 * hm1, q11 = (1 - hm0) * (1 - hm1);
  const double q01 = hm0 * (1 - hm1), q10 = (1 - hm0) * hm1;
  const double den = D < 0 ? -(q00 <= q11 ? q00 : q11) : (q01 <= q10 ? q01 : q10);
  const double Dp = D / den;
  const double rr = D / sqrt(hm0 * hm1 * (1 - hm0) * (1 - hm1));
  // a constant site (rsx = 1/sqrt(0) = inf) is 0/0 = NaN in gsl_stats_correlation; said explicitly because a cross
  // moment centred after the fact (run kernel) is only ~0 there, not exactly 0.
  // (a negative rsx: a site with a triple that does not sum to 1 whose alleles the kernels relabelled -- their Pearson moment
  // assumes e' = 2 - e there: ld_prep.hip, signed_rsx; such pairs are replayed)
  const double a1 = fabs(rsx1), a2 = fabs(rsx2);
  const bool odd_site = rsx1 < 0 || rsx2 < 0;
  const bool constant_site = a1 == __builtin_inf() || a2 == __builtin_inf();
  const double r = constant_site ? __builtin_nan("") : sxy * a1 * a2;
  ngsld_rec_std o;
  o.r2_ExpG = ref_nan(r * r);
  o.D = ref_nan(D);
  o.Dp = ref_nan(Dp);
  o.r2 = ref_nan(rr * rr);
  A.out_std[slot] = o;
  if (A.out_ext != nullptr) {
    ngsld_rec_ext e;
    e.hap[0] = ref_nan(f0); e.hap[1] = ref_nan(f1); e.hap[2] = ref_nan(f2); e.hap[3] = ref_nan(f3);
    e.n_ind_data = x;
    e.n_iter = n_iter;
    A.out_ext[slot] = e;
  }
  if (A.flags != nullptr) {
    const double q0 = fabs(hm0) <= fabs(1 - hm0) ? fabs(hm0) : fabs(1 - hm0);
    const double q1 = fabs(hm1) <= fabs(1 - hm1) ? fabs(hm1) : fabs(1 - hm1);
    // (NaN frequencies fail both comparisons)
    // host_only: reasons that concern r2_ExpG -- what the device-side replay of called genotypes (ld_replay.hip) leaves alone
    const bool pearson_bad = !constant_site && (double)A.n_ind * a1 * a2 > kPearsonCond;  // (a constant site: NaN on every path)
    bool host_only = odd_site || (pearson_bad && !A.pearson_on_device);
    // (written so that a NaN anywhere -- frequencies, D', r2 -- flags the pair)
    const double amp_q = 1.0 / q0 + 1.0 / q1, big = fabs(Dp) >= o.r2 ? fabs(Dp) : o.r2;
    bool flag = tie || host_only || pearson_bad || !(q0 >= kReplayFloor) || !(q1 >= kReplayFloor) || !(kHapNoise * amp_q * big <= kRecordTol);
    // The TSV prints six decimals (ngsLD.cpp:314-349).  A value that sits on a rounding point of the sixth decimal --
    // closer to it than this kernel and the reference can differ -- would print a different last digit, and a D within
    // rounding noise of zero a different sign ("-0.000000"): those pairs are replayed too, so that the text is the
    // reference's byte for byte.  Error bounds: hap, hap_maf and D are absolute (a few ulp of 1); D' and r2 divide by
    // products of the margins q (relative error ~ulp / q); r2_ExpG carries ~ulp * n * rsx1 * rsx2 of cancellation.
    // (flag_text: only where the records may become text -- ngsld_run; ngsld_run_device leaves them on the device)
    constexpr double kUlp = 0x1p-52;
    const double d_abs = A.flag_text ? 32 * kUlp : -1.0;  // (negative: near_rounding is never true)
    const double amp = A.flag_text ? amp_q : 0.0;
    // (a pair whose moment is ill conditioned: this r2_ExpG is not the number to look at -- whoever replays the pair does)
    host_only = host_only || (!pearson_bad && near_rounding(o.r2_ExpG, 0.5 * d_abs * (1.0 + 4.0 * (double)A.n_ind * a1 * a2)));
    flag = flag || host_only || fabs(D) < 2 * d_abs || near_rounding(D, d_abs) ||
           near_rounding(Dp, 2 * d_abs * (1.0 + fabs(Dp) * amp)) || near_rounding(o.r2, 2 * d_abs * (1.0 + o.r2 * amp));
    if (A.out_ext != nullptr)
      flag = flag || near_rounding(f0, d_abs) || near_rounding(f1, d_abs) || near_rounding(f2, d_abs) ||
             near_rounding(f3, d_abs) || near_rounding(hm0, d_abs) || near_rounding(hm1, d_abs);
    if (flag) {
      atomicOr(&A.flags[flag_head_words(A.flag_cap) + (slot >> 5)], 1u << (slot & 31u));
      if (host_only && A.flags_host != nullptr) {
        atomicOr(&A.flags_host[slot >> 5], 1u << (slot & 31u));
        const uint32_t kh = atomicAdd(&A.flags[1], 1u);
        if (kh < kFlagHostCap) reinterpret_cast<uint64_t *>(A.flags + kFlagListAt + 2u * A.flag_cap)[kh] = slot;
      }
      const uint32_t k = atomicAdd(&A.flags[0], 1u);
      if (k < A.flag_cap) reinterpret_cast<uint64_t *>(A.flags + kFlagListAt)[k] = slot | (host_only ? kFlagHostOnly : 0ull);
    }
  }
}

// What the EM leaves behind for one pair.  The derived statistics (write_pair: ~100 wavefront-uniform f64
// instructions with two divisions and a square root) are not computed by the wavefront that ran the EM -- there they
// would cost a full instruction issue each for ONE pair -- but once per work item, one LANE per pair.
struct PairResult {
  double f[4], sxy, rsx2;
  uint32_t x, n_iter;
};

// ---------------------------------------------------------------------------------------------
// Multi-wavefront kernel: WAVES = 2, 4 or 8 wavefronts share one pair (n_ind > 512, or whatever the one-wavefront kernels
// do not take).
//   SLOTS  individuals per lane (compile time, P lives in 18*SLOTS VGPRs)
//   MASKED --ignore_miss_data: individuals missing at either site are left out (gen_func.cpp:1089)
// Every wavefront only ever reads ITS slice of a site vector (individuals sub*SLOTS*64 ...), so the slice of the NEXT
// pair is copied global->LDS asynchronously into a wave-private 1536*SLOTS-byte buffer while the EM loop of the current
// pair runs; the row vector (same for the whole item, L2-hot) is read directly.  No extra barrier is needed for the
// prefetch.  (Without the prefetch -- every pair starting with an L2 / HBM round trip -- the kernel measured 12 % slower.)
// A cohort that does not fill all slots but the last (513 individuals on 2 x 5 slots: the second wavefront's fourth slot
// holds ONE individual, its fifth none) needs nothing special: empty slots are ghosts (stage_pair).
template <int SLOTS, int WAVES, bool MASKED>
__global__ __launch_bounds__(WAVES * 64, 2) void pair_ld_kernel(PairArgs A) {
  static_assert(WAVES == 2 || WAVES == 4 || WAVES == 8, "pair_ld_kernel: 2, 4 or 8 wavefronts per pair");
  constexpr int kSliceBytes = SLOTS * 64 * 3 * 8;
  constexpr int kXchBase = WAVES * kSliceBytes;
  // kParked (every individual counts): the Pearson cross moment needs no meeting of the wavefronts before the EM loop --
  // each parks its partial sum per candidate, thread t adds them up when it writes the record -- and x is n_ind: one
  // barrier, one LDS round trip and one f64 division less per pair
  constexpr bool kParked = !MASKED;
  __shared__ __attribute__((aligned(16))) char smem[kXchBase + WAVES * 96 + 64 * sizeof(PairResult) +
                                                    (kParked ? 64 * WAVES * sizeof(double) : 0)];
  PairResult *res = reinterpret_cast<PairResult *>(smem + kXchBase + WAVES * 96);  // one per candidate
  double (*parked)[WAVES] = reinterpret_cast<double (*)[WAVES]>(smem + kXchBase + WAVES * 96 + 64 * sizeof(PairResult));
  double (*xch)[WAVES][4] = reinterpret_cast<double (*)[WAVES][4]>(smem + kXchBase);
  double (*xch0)[2] = reinterpret_cast<double (*)[2]>(smem + kXchBase + WAVES * 64);

  const int lane = threadIdx.x & 63;
  const int sub = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#ifdef NGSLD_PHASE_DELAY
  // Experiment (round 5, tools/ab_phase.sh; not in the product build): the review's idea for configs[4] -- a SIMD holds one
  // wavefront of each of the CU's two workgroups, and VALU sits idle when both are in the serial stretch of their iteration --
  // start one of the two half an iteration late.  Which of the two: the wavefront slot's parity (HW_ID bits 3:0).
  {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((4 - 1) << 11));
    if (__syncthreads_or((int)(hw & 1u))) __builtin_amdgcn_s_sleep(NGSLD_PHASE_DELAY);  // (x 64 cycles)
  }
#endif
  const Item *item_ptr;
  if (A.tile_nk != 0) {
    // Tiled order.  Workgroup ids go round the eight XCDs, so with tiles of tile_rows rows x 8 items, laid out row by row,
    // XCD x works on item column x of every row of the tile: the same ~64 + tile_rows candidate sites for tile_rows rows,
    // out of its own L2 -- in plain item order the workgroups in flight together are one row's whole candidate range, no
    // site is used twice while it is anywhere on the chip, and an all-pairs run streams the matrix from HBM once per row
    // (50,000 x 1,000: 2.1 TB/s, paid for in clock: the device is at its power limit).
    const uint32_t per = A.tile_rows * 8u;
    const uint32_t t = blockIdx.x / per, w = blockIdx.x % per;
    const uint32_t row = A.row0 + (t / A.tile_nk) * A.tile_rows + (w >> 3);
    // (the column an XCD takes rotates from tile to tile: the last tile of a row block is only partly filled, and with a fixed
    // assignment the XCDs of its first columns would carry all of it -- the dispatcher deals workgroup ids round robin, an
    // XCD cannot take over another's share: measured -18 % on rows of 7-9 items)
    const uint32_t k = (t % A.tile_nk) * 8u + ((w + t) & 7u);
    if (row >= A.row1) return;
    const uint64_t lo = A.item_off[row], hi = A.item_off[row + 1];
    if ((uint64_t)k >= hi - lo) return;
    item_ptr = A.items_all + lo + k;
  } else {
    if ((uint64_t)blockIdx.x >= A.n_items) return;
    item_ptr = A.items + blockIdx.x;
  }

  const Item it = *item_ptr;
  const uint32_t s1 = it.s1;
  const double m1 = A.maf[s1];
  const double mean1 = A.mean_e[s1];
  const double rsx1 = A.rsx[s1];
  const uint64_t rec0 = it.first_record - A.out_base;
  const double *pa = A.planes + (uint64_t)s1 * A.site_stride;
  const uint32_t i0 = (uint32_t)sub * (SLOTS * 64) + (uint32_t)lane;
  char *lds_b = smem + sub * kSliceBytes;
  // the scalars of the item's candidate sites come into LDS once, by one coalesced load per array, so the pair loop waits
  // for no ordinary global load (a ~2 us round trip per pair, and it would drain the slice copy in flight)
  __shared__ double site_sc[3][64];
  if (threadIdx.x < it.count) {
    const uint32_t s2 = it.s2_begin + threadIdx.x;
    site_sc[0][threadIdx.x] = A.maf[s2];
    site_sc[1][threadIdx.x] = A.mean_e[s2];
    site_sc[2][threadIdx.x] = A.rsx[s2];
  }
  __syncthreads();

  // copy this wavefront's slice of site s2 (three runs of SLOTS*512 B, one per genotype plane) into lds_b
  auto dma_slice = [&](uint32_t s2) {
    const char *g = reinterpret_cast<const char *>(A.planes + (uint64_t)s2 * A.site_stride + (uint32_t)sub * (SLOTS * 64)) +
                    lane * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int k = 0; k < (SLOTS * 512 + 1023) / 1024; ++k)
        if ((k + 1) * 1024 <= SLOTS * 512 || lane * 16 < SLOTS * 512 - k * 1024)
          __builtin_amdgcn_global_load_lds((glb_void_t *)(g + (size_t)pl * A.np * 8 + k * 1024),
                                           (lds_void_t *)(lds_b + pl * SLOTS * 512 + k * 1024), 16, 0, 0);
  };
  auto next_kept = [&](uint32_t c) -> uint32_t {  // first computed pair at or after c (ngsLD.cpp:270-282 filters)
    while (c < it.count && !((it.mask >> c) & 1ull)) ++c;
    return c;
  };

  // The wavefront's slice of the ROW vector is the same for all 64 candidates of the item: what fits beside P is loaded once,
  // relabelled (that depends on the row's frequency only) and kept in registers, and every pair is spared those loads from L2
  // -- and part of their round trip -- at its start.  Up to six slots per lane all of it fits: +6 % at 1,281..1,536 and
  // 2,561..3,072 individuals, +8..10 % under --ignore_miss_data; seven slots take three, eight slots two (three on two
  // wavefronts): configs[3] +1.2 % (+2.5 % masked), configs[4] +1.6 % (+3.9 %), eight wavefronts +1 %, same record bits
  // (profiles/r03/sweep_multi_aregs.txt, sweep_multi_aregs_8w.txt, ab_aregs78.txt, ab_aregs_final.txt).  Nine / ten slots:
  // none (they spill as it is).
  constexpr int kNA = SLOTS <= 6   ? SLOTS
                      : SLOTS == 7 ? (WAVES == 8 ? 2 : 3)
                      : SLOTS == 8 ? (WAVES == 2 ? 3 : (WAVES == 8 && MASKED ? 0 : 2))   // 8 x 8 masked would spill 72 B
                                   : 0;
  constexpr bool kARegs = kNA > 0;
  double a_regs[kARegs ? kNA : 1][3];
  if (kARegs) {
    const bool flip1 = m1 > 0.5;  // (relabel())
    const double *q0 = pa + (flip1 ? 2 * A.np : 0u), *q1 = pa + A.np, *q2 = pa + (flip1 ? 0u : 2 * A.np);
#pragma unroll
    for (int j = 0; j < kNA; ++j) {
      a_regs[j][0] = q0[i0 + 64u * (uint32_t)j]; a_regs[j][1] = q1[i0 + 64u * (uint32_t)j]; a_regs[j][2] = q2[i0 + 64u * (uint32_t)j];
    }
  }
  uint32_t c = next_kept(0);
  if (c < it.count) dma_slice(it.s2_begin + c);
  uint32_t xpar = 0;  // exchanges of this workgroup so far (see em_pair)
  while (c < it.count) {
    const uint32_t cn = next_kept(c + 1);
    const double m2 = site_sc[0][c], mean2 = site_sc[1][c], rsx2 = site_sc[2][c];
    double P[SLOTS][9];
    uint32_t vbits;
    double sxy;
    const Relabel rl = relabel(m1, m2, mean1, mean2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slice copied during the previous pair has landed
    stage_pair<SLOTS, MASKED, false, true, true>(pa, A.np, i0, reinterpret_cast<const double *>(lds_b),
                                                  (uint32_t)(SLOTS * 64), (uint32_t)lane, i0, A.n_ind, rl.mean1, rl.mean2, P,
                                                  vbits, sxy, rl.flip1, rl.flip2, kARegs ? a_regs : nullptr, kNA);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // a, b and the scalars are all consumed
    if (cn < it.count) dma_slice(it.s2_begin + cn);
    uint32_t x = count_valid<SLOTS>(vbits);
    // the cross moment comes uncentred (round 3: read off P, no bounds tests, a DPP-only reduction -- ~80 instructions less
    // per wavefront and pair than centring every element and folding with permlane swaps); n mean1 mean2 is taken off once
    sxy = wave_sum1_bcast(sxy);
    const double centre = (double)A.n_ind * rl.mean1 * rl.mean2;
    if (kParked) {
      if (lane == 0) lds_post(lds_addr(&parked[c][sub]), sxy);
      x = A.n_ind;  // (the ballots of the wavefronts add up to it: padding lanes are the only ones left out)
    } else {
      // (no barrier behind the reads: xch0 is written again a pair later, and every EM loop has a barrier of its own that
      // no wavefront passes before all have read these)
      const uint32_t base = lds_addr(&xch0[0][0]);
      if (lane == 0) lds_post2(base + (uint32_t)sub * 16u, sxy, (double)x);
      lds_barrier();
      dbl2 q[WAVES];
      lds_gather<WAVES>(base, q);
      double sx = 0.0, xs = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        sx += q[w][0];
        xs += q[w][1];
      }
      sxy = sx - centre;
      x = (uint32_t)xs;
    }
    double f0, f1, f2, f3;
    const uint32_t n_iter = em_pair<SLOTS, WAVES>(P, vbits, kParked ? A.inv_n : 1.0 / (double)x, rl.m1, rl.m2, f0, f1, f2, f3,
                                                  xch, sub, lane, A.status, &xpar);
    unrelabel(rl.flip1, rl.flip2, f0, f1, f2, f3);
    if (lane == 0 && sub == 0) {
      PairResult &r = res[c];
      r.f[0] = f0; r.f[1] = f1; r.f[2] = f2; r.f[3] = f3;
      r.sxy = kParked ? centre : sxy;  // (parked partial sums: the centring term travels in their place)
      r.rsx2 = rsx2;
      r.x = x;
      r.n_iter = n_iter;
    }
    c = cn;
  }
  // the whole workgroup shares the item: thread t derives and writes the record of candidate t
  if (kParked) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the parked partial sums are stores the compiler does not see
  __syncthreads();
  const uint32_t t = threadIdx.x;
  if (t < it.count && ((it.mask >> t) & 1ull)) {
    const PairResult r = res[t];
    double sxy = r.sxy;
    if (kParked) {
      sxy = 0.0;  // (the order the exchange added them in)
      for (int w = 0; w < WAVES; ++w) sxy += parked[t][w];
      sxy -= r.sxy;  // centred: sum e1 e2 - n mean1 mean2
    }
    write_pair(A, rec0 + (uint64_t)__popcll(it.mask & ((1ull << t) - 1ull)), r.f[0], r.f[1], r.f[2], r.f[3], sxy, rsx1,
               r.rsx2, r.x, r.n_iter);
  }
}

// ---------------------------------------------------------------------------------------------
// One wavefront per pair (n_ind <= 640): the four wavefronts of a workgroup work on ONE row s1, whose vector sits in LDS.
// Each wavefront claims the next s2 from an LDS counter (dynamic balance of the 3..100-iteration spread), and as soon as
// it has turned the current buffer into P it starts the asynchronous copy (global_load_lds, 16 B per lane, no VGPR round
// trip) of the site it will work on NEXT -- the copy flies during the whole EM loop, so the ~2.5 us HBM/Infinity-Cache
// latency that a direct load pays at every pair start is off the critical path (measured: 1.62e8 against 1.42e8 pairs/s).
// ---------------------------------------------------------------------------------------------
// Asynchronous copy of one site's planes (SLOTS*1536 B, contiguous) into LDS, 1 KiB per wave-instruction
// (lane l moves 16 B to lds_dst + k*1024 + l*16).  With `stride` > 1 only chunks k % stride == first are
// issued (several wavefronts sharing one copy).  A trailing half chunk (odd SLOTS) is issued by lanes 0..31.
template <int SLOTS>
__device__ __forceinline__ void dma_site_to_lds(const double *site, char *lds_dst, int lane, int first, int stride) {
  constexpr int kBytes = SLOTS * 64 * 3 * 8;
  constexpr int kChunks = (kBytes + 1023) / 1024;
  const char *g = reinterpret_cast<const char *>(site) + lane * 16;
  // The instruction's immediate offset applies to the global AND the LDS address, and the copy is contiguous on both
  // sides: four chunks share one address pair (offsets 0 .. 3072 fit the 12-bit field) instead of one 64-bit add and
  // one M0 write per chunk.
#pragma unroll
  for (int k0 = 0; k0 < kChunks; k0 += 4) {
    glb_void_t *gb = (glb_void_t *)(g + k0 * 1024);
    lds_void_t *lb = (lds_void_t *)(lds_dst + k0 * 1024);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = k0 + kk;
      if (k >= kChunks) break;
      if (stride != 1 && (k % stride) != first) continue;
      if ((k + 1) * 1024 <= kBytes || lane * 16 < kBytes - k * 1024) {
        switch (kk) {
          case 0: __builtin_amdgcn_global_load_lds(gb, lb, 16, 0, 0); break;
          case 1: __builtin_amdgcn_global_load_lds(gb, lb, 16, 1024, 0); break;
          case 2: __builtin_amdgcn_global_load_lds(gb, lb, 16, 2048, 0); break;
          default: __builtin_amdgcn_global_load_lds(gb, lb, 16, 3072, 0); break;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Run kernel (n_ind <= 640; eight slots = the headline shape): the pair pipeline above without per-item costs.
// A workgroup works through a RUN of up to kRunItems consecutive items of one row (512 candidate sites) instead of
// one item: the row vector is brought into LDS once per run, the four wavefronts claim candidates from one LDS
// counter for the whole run, and NOTHING inside the run synchronises them -- no barrier at item boundaries, no
// workgroup turnover every 64 pairs (with one item per workgroup the SIMDs held 1.8 of 2 wavefronts on average:
// launch, the barrier at the item's end and the wait for its slowest wavefront).
//   * item headers of the run sit in LDS (claims need mask / count / first_record; a global load per claim would be
//     a ~2 us round trip on the critical path);
//   * a site's scalars {maf, mean_e, rsx} travel with its planes: one more 32-byte global->LDS copy behind the site
//     copy, so the pair loop has no ordinary global load to wait for at all;
//   * results collect in a wave-private LDS ring and are turned into records 32 at a time, one LANE per pair
//     (write_pair is ~100 wavefront-uniform f64 instructions: issued per pair they would cost 4 % of the kernel).
//   LDS: [row vector][4 x (site buffer + 32 B scalars)][4 x ring of 32 results][item headers][claim counter]
// ---------------------------------------------------------------------------------------------
struct RunResult {
  double f[4], sxy, rsx2;
  uint32_t x, n_iter;
  uint64_t rec;
};

// The computed pairs of a run, as a list: cand[j] = candidate index (64 * item + offset = s2 - s2 of the run's first
// candidate) of the run's j-th computed pair, in increasing s2 -- so its record is simply the run's first record + j.
// Built once per run from the items' masks by the whole workgroup; a claim is then one LDS atomic and one 2-byte read
// whatever the masks look like.  (Claiming candidate by candidate and skipping the masked-out ones cost a dependent LDS
// round trip per dropped candidate: -11 % at --rnd_sample 0.1, -49 % at 0.02.)
struct RunList {
  uint16_t cand[kRunItems * 64];
  uint32_t base[kRunItems + 1];  // computed pairs before each item; base[n_items] = all of the run's
  uint32_t claim;
  uint32_t pad[2];
  Item items[kRunItems];         // the run's item headers
};

// Called by all 256 threads; ends with a barrier (which also completes whatever global->LDS copies the callers issued
// before it: __syncthreads waits for the wavefront's own memory operations first).
__device__ __forceinline__ void build_run_list(RunList *L, const Item *g_items, uint32_t n_items) {
  if (threadIdx.x < n_items * 2)  // item headers, 16 bytes per thread
    reinterpret_cast<uint4 *>(L->items)[threadIdx.x] = reinterpret_cast<const uint4 *>(g_items)[threadIdx.x];
  if (threadIdx.x == 0) L->claim = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t k = 0; k < n_items; ++k) {
      L->base[k] = acc;
      acc += (uint32_t)__popcll(L->items[k].mask);  // bits at or beyond `count` are never set (items_kernel)
    }
    L->base[n_items] = acc;
  }
  __syncthreads();
  for (uint32_t idx = threadIdx.x; idx < n_items * 64; idx += 256) {
    const uint32_t k = idx >> 6, c = idx & 63u;
    const unsigned long long m = L->items[k].mask;
    if ((m >> c) & 1ull) L->cand[L->base[k] + (uint32_t)__popcll(m & ((1ull << c) - 1ull))] = (uint16_t)idx;
  }
  __syncthreads();
}

template <int SLOTS, bool MASKED>
__global__ __launch_bounds__(256, 2) void pair_ld_run_kernel(PairArgs A) {
  constexpr int kSiteBytes = SLOTS * 64 * 3 * 8;
  constexpr int kBuf = kSiteBytes + 32;
  constexpr uint32_t kNp = SLOTS * 64;
  // (ten slots: five site buffers of 15 KB leave 5 KB for rings and list under the 80 KB that let two workgroups share a CU)
  constexpr uint32_t kRing = SLOTS <= 9 ? 32 : 8;
  constexpr int kRingOff = kSiteBytes + 4 * kBuf;
  constexpr int kListOff = kRingOff + 4 * (int)(kRing * sizeof(RunResult));
  __shared__ __attribute__((aligned(16))) char smem[kListOff + sizeof(RunList)];
  static_assert(sizeof(smem) <= 81920, "run kernel: two workgroups per CU need <= 80 KB of LDS each");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // (An XCD-aware order -- each XCD taking 64 consecutive runs of every 512 -- was measured and dropped: neighbouring
  // rows drift apart by more pairs than the 4 MB L2 bridges, L2-miss traffic rose 35 % and the kernel lost 0.7 %.)
  const Run run = A.runs[blockIdx.x];
  const Item *g_items = A.items_all + run.first_item;
  const uint32_t s1 = g_items[0].s1;
  const double m1 = A.sc4[4 * (uint64_t)s1], mean1 = A.sc4[4 * (uint64_t)s1 + 1], rsx1 = A.sc4[4 * (uint64_t)s1 + 2];
  char *lds_a = smem;
  char *lds_b = smem + kSiteBytes + wave * kBuf;
  RunResult *ring = reinterpret_cast<RunResult *>(smem + kRingOff) + wave * kRing;
  RunList *L = reinterpret_cast<RunList *>(smem + kListOff);

  dma_site_to_lds<SLOTS>(A.planes + (uint64_t)s1 * A.site_stride, lds_a, lane, wave, 4);  // a quarter per wavefront
  build_run_list(L, g_items, run.n_items);  // its last barrier: row vector and list in place
  const uint32_t n_kept = (uint32_t)__builtin_amdgcn_readfirstlane((int)L->base[run.n_items]);
  const uint32_t s2_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)L->items[0].s2_begin);
  const uint64_t rec_base = g_items[0].first_record - A.out_base;

  // one computed pair of the run: site and record index
  struct Cand {
    uint32_t s2;
    uint64_t rec;
    bool ok;
  };
  auto claim_next = [&]() -> Cand {  // (the maf[s2] / sub-sampling filters, ngsLD.cpp:270-282, already shaped the list)
    uint32_t j = 0;
    if (lane == 0) j = atomicAdd(&L->claim, 1u);
    j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
    if (j >= n_kept) return Cand{0u, 0ull, false};
    const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)L->cand[j]);
    return Cand{s2_base + off, rec_base + j, true};
  };
  auto dma_site = [&](uint32_t s2) {
    dma_site_to_lds<SLOTS>(A.planes + (uint64_t)s2 * A.site_stride, lds_b, lane, 0, 1);
    if (lane < 2)
      __builtin_amdgcn_global_load_lds((glb_void_t *)(reinterpret_cast<const char *>(A.sc4 + 4 * (uint64_t)s2) + lane * 16),
                                       (lds_void_t *)(lds_b + kSiteBytes), 16, 0, 0);
  };
  auto flush = [&](uint32_t n) {  // lane t derives and writes the record of ring entry t
    if ((uint32_t)lane < n) {
      const RunResult r = ring[lane];
      write_pair(A, r.rec, r.f[0], r.f[1], r.f[2], r.f[3], r.sxy, rsx1, r.rsx2, r.x, r.n_iter);
    }
  };

  Cand cur = claim_next();
  if (cur.ok) dma_site(cur.s2);
  uint32_t held = 0;
  while (cur.ok) {
    const Cand nxt = claim_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's site copy (issued a pair ago) has landed
    const double *sc = reinterpret_cast<const double *>(lds_b + kSiteBytes);
    const double m2 = uniform(sc[0]), mean2 = uniform(sc[1]), rsx2 = uniform(sc[2]);
    double P[SLOTS][9];
    uint32_t vbits;
    double sxy;
    const Relabel rl = relabel(m1, m2, mean1, mean2);
    stage_pair<SLOTS, MASKED, !MASKED, true>(reinterpret_cast<const double *>(lds_a), kNp, (uint32_t)lane,
                                             reinterpret_cast<const double *>(lds_b), kNp, (uint32_t)lane, (uint32_t)lane,
                                             A.n_ind, rl.mean1, rl.mean2, P, vbits, sxy, rl.flip1, rl.flip2);
    // all ds_reads of the buffer are consumed (P is computed): start the copy of the next site over it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (nxt.ok) dma_site(nxt.s2);
    const uint32_t x = MASKED ? count_valid<SLOTS>(vbits) : A.n_ind;
    const double inv_x = MASKED ? 1.0 / (double)x : A.inv_n;
    sxy = fma(-(double)A.n_ind * rl.mean1, rl.mean2, wave_sum1_bcast(sxy));  // centred: sum e1 e2 - n mean1 mean2
    double f0, f1, f2, f3;
    const uint32_t n_iter = em_pair<SLOTS, 1>(P, vbits, inv_x, rl.m1, rl.m2, f0, f1, f2, f3, (double (*)[1][4]) nullptr, 0,
                                              lane, A.status);
    unrelabel(rl.flip1, rl.flip2, f0, f1, f2, f3);
    if (lane == 0) {
      RunResult &r = ring[held];
      r.f[0] = f0; r.f[1] = f1; r.f[2] = f2; r.f[3] = f3;
      r.sxy = sxy;
      r.rsx2 = rsx2;
      r.x = x;
      r.n_iter = n_iter;
      r.rec = cur.rec;
    }
    if (++held == kRing) {
      flush(held);
      held = 0;
    }
    cur = nxt;
  }
  flush(held);
}

// ---------------------------------------------------------------------------------------------
// Group kernel (n_ind <= 256): a group of G = 8, 16 or 32 lanes owns one pair, so a wavefront runs 8, 4 or 2 pairs in
// lockstep, all of one row s1.  With few individuals the per-iteration bookkeeping (f products, contraction,
// reduction, convergence test) outweighs the per-individual work; sharing each of those instructions between the
// pairs of a wavefront is worth more than the lanes lost to lockstep (a group that has converged idles until the
// slowest group of its wavefront has).
//   lane = G*grp + r;  individual of (lane, slot j) = G*j + r;  SLOTS = ceil(n_ind / G) <= 8;  np = G*SLOTS
//   LDS: [row vector a, linear][per wavefront: next-site buffers of its 64/G groups, interleaved in pieces of 16*G
//        bytes because global_load_lds writes wave-base + 16*lane: piece q of group gg sits at q*1024 + gg*16*G]
// Values that are wavefront-uniform in the 64-lane kernels (f, the f products, eps) are group-uniform VGPR values
// here; reductions are DPP steps inside the group, in a fixed order.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) { return v + dpp_mov<CTRL>(v); }

template <int G>
__device__ __forceinline__ double group_sum(double v) {  // sum over the G lanes of a group, result in every lane of it
  v = dpp_add<0xB1>(v);               // quad_perm:[1,0,3,2]
  v = dpp_add<0x4E>(v);               // quad_perm:[2,3,0,1]
  if (G == 8) return dpp_add<0x141>(v);  // row_half_mirror: lane l <-> 7 - l inside each 8-lane half row
  v = dpp_add<0x124>(v);              // row_ror:4
  v = dpp_add<0x128>(v);              // row_ror:8
  if (G == 32) v = fold16(v, v);      // odd rows trade places with even rows of the copy: row0+row1 | row2+row3
  return v;
}

// Three values at once (G = 16 or 32): after the first DPP level a value sits twice in every lane pair, after the second
// four times in every quad -- so the second level runs on TWO registers (t1 | t2 packed by lane parity, and t3) and the
// remaining ones on ONE (lane % 4 == 0: t1, 1: t2, 2 and 3: t3); three quad broadcasts hand the totals back to every lane
// of the group.  G = 16: 31 instructions instead of 36 (7 f64 adds instead of 12); G = 32: one v_permlane16_swap fold
// instead of three.  (G = 8 has only three levels: packing does not pay there.)
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {  // quad_perm broadcast of one lane of every quad
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int G>
__device__ __forceinline__ void group_sum3(double &t1, double &t2, double &t3, bool odd, bool upper) {
  // odd = lane & 1, upper = lane & 2 (loop invariants of the caller)
  if (G == 8) {
    t1 = group_sum<G>(t1); t2 = group_sum<G>(t2); t3 = group_sum<G>(t3);
    return;
  }
  t1 = dpp_add<0xB1>(t1); t2 = dpp_add<0xB1>(t2); t3 = dpp_add<0xB1>(t3);  // quad_perm:[1,0,3,2]
  double u = odd ? t2 : t1;
  u = dpp_add<0x4E>(u); t3 = dpp_add<0x4E>(t3);                              // quad_perm:[2,3,0,1]
  double w = upper ? t3 : u;
  w = dpp_add<0x124>(w);  // row_ror:4
  w = dpp_add<0x128>(w);  // row_ror:8
  if (G == 32) w = fold16(w, w);
  t1 = dpp_quad<0x00>(w);  // quad_perm:[0,0,0,0]
  t2 = dpp_quad<0x55>(w);  // quad_perm:[1,1,1,1]
  t3 = dpp_quad<0xAA>(w);  // quad_perm:[2,2,2,2]
}

template <int G, int SLOTS, bool MASKED>
__global__ __launch_bounds__(256, 2) void pair_ld_group_kernel(PairArgs A) {
  constexpr uint32_t kNp = SLOTS * G;
  constexpr int kSiteBytes = (int)kNp * 24;
  constexpr int kPiece = G * 16;                             // bytes one group moves per copy instruction
  constexpr int kPieces = (kSiteBytes + kPiece - 1) / kPiece;
  constexpr int kABytes = ((kSiteBytes + 1023) / 1024) * 1024;
  constexpr int kWaveBuf = (kPieces + 1) * 1024;             // the 64/G groups of a wavefront, interleaved, + their scalars
  constexpr int kGroups = 64 / G;
  constexpr uint32_t kRing = 32;
  constexpr int kRingOff = kABytes + 4 * kWaveBuf;
  constexpr int kListOff = kRingOff + 4 * (int)(kRing * sizeof(RunResult));
  constexpr unsigned long long kGroupMask = G == 32 ? 0xffffffffull : ((1ull << G) - 1ull);
  __shared__ __attribute__((aligned(16))) char smem[kListOff + sizeof(RunList)];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = lane / G, gl = lane % G;
  // Run form (see pair_ld_run_kernel): the workgroup works through up to kRunItems consecutive items of one row; the row
  // vector is loaded once, groups claim candidates from one counter for the whole run, item headers sit in LDS, a site's
  // scalars travel with its planes, results go through a wave-private ring.  With a pair costing a few microseconds at
  // these cohort sizes, one workgroup per 64-candidate item meant a workgroup turnover every ~10 us.
  const Run run = A.runs[blockIdx.x];
  const Item *g_items = A.items_all + run.first_item;
  const uint32_t s1 = g_items[0].s1;
  const double m1_in = A.sc4[4 * (uint64_t)s1], mean1_in = A.sc4[4 * (uint64_t)s1 + 1], rsx1 = A.sc4[4 * (uint64_t)s1 + 2];
  char *lds_a = smem;
  char *lds_w = smem + kABytes + wave * kWaveBuf;
  RunResult *ring = reinterpret_cast<RunResult *>(smem + kRingOff) + wave * kRing;
  RunList *L = reinterpret_cast<RunList *>(smem + kListOff);

  // one computed pair of the run per group: site and record index (group-uniform values)
  struct Cand {
    uint32_t s2;
    uint64_t rec;
    bool ok;
  };
  uint32_t n_kept = 0, s2_base = 0;   // set once the run's list is built
  uint64_t rec_base = 0;
  // a group claims the next computed pair of the run (the maf[s2] / sub-sampling filters, ngsLD.cpp:270-282, shaped the list)
  auto claim_group = [&]() -> Cand {
    uint32_t j = 0;
    if (gl == 0) j = atomicAdd(&L->claim, 1u);
    j = (uint32_t)__shfl((int)j, lane & ~(G - 1));
    if (j >= n_kept) return Cand{0u, 0ull, false};
    return Cand{s2_base + (uint32_t)L->cand[j], rec_base + j, true};
  };
  // byte offset of individual-slot j, genotype plane g of this lane's group inside the interleaved wave buffer
  auto b_off = [&](int g, int j) -> uint32_t {
    const uint32_t o = ((uint32_t)g * kNp + (uint32_t)j * (uint32_t)G + (uint32_t)gl) * 8u;  // offset inside the site
    return (o / (uint32_t)kPiece) * 1024u + (uint32_t)grp * (uint32_t)kPiece + (o % (uint32_t)kPiece);
  };
  // start the copy of every group's next site: lane (grp, r) moves the 16 bytes [q*kPiece + r*16, +16) of its group's
  // site for q = 0 .. kPieces-1; lanes r = 0, 1 then move the site's 32 bytes of scalars {maf, mean_e, rsx, 0}
  auto dma_groups = [&](const Cand &cd) {
    const char *g = reinterpret_cast<const char *>(A.planes + (uint64_t)(cd.ok ? cd.s2 : 0u) * A.site_stride) + gl * 16;
#pragma unroll
    for (int q = 0; q < kPieces; ++q)
      if (cd.ok && q * kPiece + gl * 16 < kSiteBytes)
        __builtin_amdgcn_global_load_lds((glb_void_t *)(g + q * kPiece), (lds_void_t *)(lds_w + q * 1024), 16, 0, 0);
    if (cd.ok && gl < 2)
      __builtin_amdgcn_global_load_lds((glb_void_t *)(reinterpret_cast<const char *>(A.sc4 + 4 * (uint64_t)cd.s2) + gl * 16),
                                       (lds_void_t *)(lds_w + kPieces * 1024), 16, 0, 0);
  };

  // the row vector: linear copy, 1 KiB per wave-instruction, chunks dealt round-robin to the four wavefronts
  {
    const char *g = reinterpret_cast<const char *>(A.planes + (uint64_t)s1 * A.site_stride) + lane * 16;
#pragma unroll
    for (int k = 0; k < kABytes / 1024; ++k)
      if ((k & 3) == wave && k * 1024 + lane * 16 < kSiteBytes)
        __builtin_amdgcn_global_load_lds((glb_void_t *)(g + k * 1024), (lds_void_t *)(lds_a + k * 1024), 16, 0, 0);
  }
  build_run_list(L, g_items, run.n_items);  // its last barrier: row vector and list in place
  n_kept = L->base[run.n_items];
  s2_base = L->items[0].s2_begin;
  rec_base = g_items[0].first_record - A.out_base;
  Cand cur = claim_group();
  dma_groups(cur);
  uint32_t held = 0;
  auto flush = [&](uint32_t n) {  // lane t derives and writes the record of ring entry t (holes: groups without a pair)
    if ((uint32_t)lane < n) {
      const RunResult r = ring[lane];
      if (r.rec != ~0ull) write_pair(A, r.rec, r.f[0], r.f[1], r.f[2], r.f[3], r.sxy, rsx1, r.rsx2, r.x, r.n_iter);
    }
  };

  while (__any(cur.ok)) {
    const bool active = cur.ok;
    Cand nxt = cur;
    if (active) nxt = claim_group();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the site copies issued a generation ago have landed
    const double *sc = reinterpret_cast<const double *>(lds_w + kPieces * 1024 + grp * kPiece);
    const double m2_in = active ? sc[0] : 0.5, mean2_in = active ? sc[1] : 0.0, rsx2 = active ? sc[2] : 0.0;
    // allele relabelling (see Relabel): site 1's is the same for the whole run, site 2's differs from group to group
    const Relabel rl = relabel(m1_in, m2_in, mean1_in, mean2_in);
    const double m1 = rl.m1, m2 = rl.m2, mean1 = rl.mean1, mean2 = rl.mean2;
    const int gb0 = rl.flip2 ? 2 : 0, gb2 = rl.flip2 ? 0 : 2;

    // ---- stage: P = a (x) b per lane, validity, Pearson cross moment (group sums) ----
    double P[SLOTS][9];
    uint32_t vbits = 0;
    double sxy = 0.0;
    const double *la = reinterpret_cast<const double *>(lds_a);
    const double *la0 = la + (rl.flip1 ? 2 * kNp : 0u), *la2 = la + (rl.flip1 ? 0u : 2 * kNp);
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      const uint32_t i = (uint32_t)j * (uint32_t)G + (uint32_t)gl;
      const double a0 = la0[i], a1 = la[kNp + i], a2 = la2[i];
      const double b0 = *reinterpret_cast<const double *>(lds_w + b_off(gb0, j));
      const double b1 = *reinterpret_cast<const double *>(lds_w + b_off(1, j));
      const double b2 = *reinterpret_cast<const double *>(lds_w + b_off(gb2, j));
      const bool inb = i < A.n_ind;
      bool ok = inb && active;
      if (MASKED) ok = ok && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);  // gen_func.cpp:1089
      vbits |= (ok ? 1u : 0u) << j;
      double z0 = a0, z1 = a1, z2 = a2;
      if (MASKED) {  // an individual without data is a ghost, P = (1, 0, ..., 0): see stage_pair
        const double keep = ok ? 1.0 : 0.0;
        z0 = a0 * keep; z1 = a1 * keep; z2 = a2 * keep;
        P[j][0] = fma(z0, b0, 1.0 - keep);
      } else if (j == SLOTS - 1) {  // padding lanes (zeros in the planes) of the last slot
        P[j][0] = fma(a0, b0, inb ? 0.0 : 1.0);
      } else {
        P[j][0] = a0 * b0;
      }
      P[j][1] = z0 * b1; P[j][2] = z0 * b2;
      P[j][3] = z1 * b0; P[j][4] = z1 * b1; P[j][5] = z1 * b2;
      P[j][6] = z2 * b0; P[j][7] = z2 * b1; P[j][8] = z2 * b2;
      // expected genotypes p1 + 2 p2 (ngsLD.cpp:113); pearson_r runs over ALL individuals (ngsLD.cpp:290); the planes
      // hold zeros beyond n_ind, so the uncentred cross moment needs no bounds test
      if (!MASKED)  // (a1 + 2 a2)(b1 + 2 b2) = P4 + 2 P5 + 2 P7 + 4 P8
        sxy += fma(4.0, P[j][8], fma(2.0, P[j][5] + P[j][7], P[j][4]));
      else          // P of individuals without data is zeroed: take the moment from a and b
        sxy = fma(fma(2.0, a2, a1), fma(2.0, b2, b1), sxy);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // buffers consumed: start the next generation's copies
    dma_groups(nxt);
    // individuals with data in this group's pair (gen_func.cpp:1091), integer exact: everybody without --ignore_miss_data
    // (then 1/x comes precomputed -- the same IEEE quotient -- instead of a ~35-instruction f64 division per generation)
    uint32_t x = A.n_ind;
    if (MASKED) {
      x = 0;
#pragma unroll
      for (int j = 0; j < SLOTS; ++j)
        x += (uint32_t)__popcll((__ballot((vbits >> j) & 1u) >> (grp * G)) & kGroupMask);
    }
    // centred once per pair: sum e1 e2 - n mean1 mean2 (as the run kernel)
    sxy = fma(-(double)A.n_ind * mean1, mean2, group_sum<G>(sxy));

    // ---- haplo_freq (gen_func.cpp:1027-1059), 64/G pairs in lockstep ----
    double f0 = (1 - m1) * (1 - m2), f1 = (1 - m1) * m2, f2 = m1 * (1 - m2), f3 = m1 * m2;
    if (active && (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1)) {
      if (gl == 0) atomicExch(A.status, (int)NGSLD_ERR_MAF_RANGE);
      f0 = f1 = f2 = f3 = __builtin_nan("");
    }
    const double inv_x = MASKED ? 1.0 / (double)x : A.inv_n;
    // one reciprocal per lane and iteration (RcpTree), empty slots are ghosts: see em_pair
    constexpr bool kTree = SLOTS > 1;
    auto em_step = [&](auto tree_tag, double &n0, double &n1, double &n2, double &n3) {
      constexpr bool kT = decltype(tree_tag)::value;
      constexpr bool kDrop = kT;  // shared-reciprocal step: three-value form; the other one: full (see em_pair)
      const double p00 = f0 * f0, p01 = f0 * f1, p02 = f0 * f2, p03 = f0 * f3, p11 = f1 * f1;
      const double p12 = f1 * f2, p13 = f1 * f3, p22 = f2 * f2, p23 = f2 * f3, p33 = f3 * f3;
      const double w1 = p01 + p01, w3 = p02 + p02, w4 = 2.0 * (p03 + p12), w5 = p13 + p13, w7 = p23 + p23;
      double R0 = 0, R1 = 0, R2 = 0, R3 = 0, R4 = 0, R5 = 0, R6 = 0, R7 = 0, R8 = 0;
      auto slot_s = [&](int j) -> double {
        double s = p00 * P[j][0];
        s = fma(w1, P[j][1], s); s = fma(p11, P[j][2], s);
        s = fma(w3, P[j][3], s); s = fma(w4, P[j][4], s); s = fma(w5, P[j][5], s);
        s = fma(p22, P[j][6], s); s = fma(w7, P[j][7], s); s = fma(p33, P[j][8], s);
        return s;
      };
      auto slot_acc = [&](int j, double r) {
        if (!kDrop) R0 = fma(P[j][0], r, R0);
        R1 = fma(P[j][1], r, R1); R2 = fma(P[j][2], r, R2);
        R3 = fma(P[j][3], r, R3); R4 = fma(P[j][4], r, R4); R5 = fma(P[j][5], r, R5);
        R6 = fma(P[j][6], r, R6); R7 = fma(P[j][7], r, R7); R8 = fma(P[j][8], r, R8);
      };
      if constexpr (kT) {
        double sv[SLOTS], rv[SLOTS];
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(0);  // dense stretch: see em_pair
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) sv[j] = slot_s(j);
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(3);
        // 1/x rides on the root inverse (as in em_pair): every R, and with them the three t_k, come out divided by x
        RcpTree<SLOTS>::down(sv, rcp_refined(RcpTree<SLOTS>::prod(sv)) * inv_x, rv);
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) slot_acc(j, rv[j]);
        if (NGSLD_SETPRIO) __builtin_amdgcn_s_setprio(3);
      } else {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
          // without --ignore_miss_data only the last slot can hold padding lanes; a group without a pair computes
          // on stale buffers there, which is harmless (it is `done` from the start and never written)
          if ((!MASKED && j < SLOTS - 1) || ((vbits >> j) & 1u)) slot_acc(j, rcp_refined(slot_s(j)));
        }
      }
      double t1 = fma(p13, R5, fma(p12, R4, fma(p11, R2, p01 * R1)));
      double t2 = fma(p23, R7, fma(p22, R6, fma(p12, R4, p02 * R3)));
      double t3 = fma(p33, R8, fma(p23, R7, fma(p13, R5, p03 * R4)));
      group_sum3<G>(t1, t2, t3, (lane & 1) != 0, (lane & 2) != 0);
      if (kT) {
        n1 = t1; n2 = t2; n3 = t3;  // already divided by x
      } else {
        n1 = t1 * inv_x; n2 = t2 * inv_x; n3 = t3 * inv_x;
      }
      if (kDrop) {  // the first frequency is what the other three leave (see em_pair)
        n0 = 1.0 - ((n1 + n2) + n3);
      } else {
        const double t0 = group_sum<G>(fma(p03, R4, fma(p02, R3, fma(p01, R1, p00 * R0))));
        n0 = kT ? t0 : t0 * inv_x;
      }
    };
    bool done = !active, tie = false;
    uint32_t n_iter = (uint32_t)kIterMax;
    constexpr bool kMaskDone = SLOTS >= NGSLD_MASK_SLOTS;
    // As in em_pair: the hot loop holds the shared-reciprocal step in its three-value form only; a step that is not sane
    // in some live group leaves it for one iteration with a reciprocal per individual, and as soon as hap 0 of any live
    // group falls below kFullBelow the wavefront leaves it for good and finishes in the full four-value form.
    constexpr double kFullBelow = 0x1p-10;
    bool full = __any(!done && f0 < kFullBelow);
    uint32_t itn = 0;
    while (itn < (uint32_t)kIterMax) {
      if (kTree && !full) {
        bool all_done = false;
        for (; itn < (uint32_t)kIterMax; ++itn) {
          double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
          // A group that has converged idles until the slowest group of its wavefront has.  It idles with its lanes
          // SWITCHED OFF (EXEC), not computing on stale values: the device runs these kernels at its power limit
          // (1.35 kW, 2.05-2.1 GHz of 2.4: profiles/r02/clocks_power_r02.txt), so what idle lanes do not burn comes
          // back as clock.  (Every cross-lane step of the EM stays inside a group, all of whose lanes are on or off.)
          // Same-box A/B (tools/ab_mask.sh): +5.5 % at n_ind 100 (configs[1]), +6.9 % at 64, +2 % at 48, +1 % at 200 (two
          // groups: little to idle); -2.5 % at 24 / 32 / 40 and in the genotype-combination kernel, whose iterations are
          // too short for the mask's own bookkeeping -- hence only from NGSLD_MASK_SLOTS individuals per lane on.
          if (!kMaskDone || !done) em_step(PairedTag(), n0, n1, n2, n3);
          if (__any(!done && !(n1 < 2.0))) break;  // an odd step (one NaN reciprocal poisons every R, see em_pair)
          // as in em_pair: eps is at least the change of hap 1, and while that alone is above EPSILON in every live
          // group the other three differences are not formed
          if (__any(!done && fabs(n1 - f1) < kEpsilonTie)) {
            const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
            if (!done && fabs(eps - kEpsilon) < kTieMargin) tie = true;  // too close to call: replayed
            if (!done && eps < kEpsilon) {  // gen_func.cpp:1054-1055
              done = true;
              n_iter = itn;
              f0 = n0; f1 = n1; f2 = n2; f3 = n3;
            }
          }
          if (!done) {
            f0 = n0; f1 = n1; f2 = n2; f3 = n3;
          }
          if (__all(done)) {
            all_done = true;
            break;
          }
          if (__any(!done && f0 < kFullBelow)) {
            full = true;
            ++itn;  // this iteration is complete
            break;
          }
        }
        if (all_done || itn >= (uint32_t)kIterMax) break;
        if (full) continue;
      }
      double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
      if (!kMaskDone || !done) em_step(SingleTag(), n0, n1, n2, n3);
      const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
      if (!done) {
        if (!(n1 < 2.0)) {  // the reference's all-NaN step: "converges" at this iteration (see em_pair)
          f0 = f1 = f2 = f3 = __builtin_nan("");
          done = true;
          n_iter = itn;
        } else {
          f0 = n0; f1 = n1; f2 = n2; f3 = n3;
          if (fabs(eps - kEpsilon) < kTieMargin) tie = true;
          if (eps < kEpsilon) {
            done = true;
            n_iter = itn;
          }
        }
      }
      if (__all(done)) break;
      ++itn;
    }

    unrelabel(rl.flip1, rl.flip2, f0, f1, f2, f3);
    if (gl == 0) {  // one ring entry per group and generation; a group without a pair leaves a hole
      RunResult &r = ring[held + (uint32_t)grp];
      r.f[0] = f0; r.f[1] = f1; r.f[2] = f2; r.f[3] = f3;
      r.sxy = sxy;
      r.rsx2 = rsx2;
      r.x = x;
      r.n_iter = n_iter | (tie ? kTieBit : 0u);
      r.rec = active ? cur.rec : ~0ull;
    }
    held += (uint32_t)kGroups;
    if (held + (uint32_t)kGroups > kRing) {
      flush(held);
      held = 0;
    }
    cur = nxt;
  }
  flush(held);
}

// ---------------------------------------------------------------------------------------------
// Streaming kernel (n_ind > 10,240; 5,121 .. 10,240 go to pair_ld_bres_kernel below): no limit on the number of individuals.  One 256-thread workgroup per pair at a
// time; wavefront w takes the 64-individual blocks w, w+4, w+8, ...  P does not fit in registers any more, so
// every EM iteration re-reads both site vectors (from L2: a pair's two vectors are 48*n_ind bytes) and forms
//   s = sum_g1 a[g1] * (sum_g2 W[g1][g2] b[g2]),   R[g1][g2] += (r a[g1]) * b[g2]
// on the fly: 24 f64 VALU + rcp + 6 loads per individual and iteration instead of 21 + rcp from registers.
// Same reduction order rules as the other kernels (fixed, deterministic).
// (Round 3 tried ONE wavefront per pair instead -- no exchange, no barrier, the four wavefronts of a workgroup walking the same
// row vector so that a neighbour's lines in the CU's L1 would serve the a-loads: -15..-17 % at 5,121..10,000 individuals,
// profiles/r03/sweep_stream.txt.  A wavefront streaming a whole pair alone has a quarter of the loads in flight per pair, and
// the L1 sharing did not happen.)
// ---------------------------------------------------------------------------------------------
template <bool MASKED>
__global__ __launch_bounds__(256, 2) void pair_ld_stream_kernel(PairArgs A) {
  __shared__ double xch[2][4][4];
  __shared__ double xch0[4][2];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const Item it = A.items[blockIdx.x];
  const uint32_t s1 = it.s1;
  const double m1 = A.maf[s1];
  const double mean1 = A.mean_e[s1];
  const double rsx1 = A.rsx[s1];
  const uint64_t rec0 = it.first_record - A.out_base;
  const double *pa = A.planes + (uint64_t)s1 * A.site_stride;
  const uint32_t np = A.np;
  const uint32_t n_blocks = np / 64;

  for (uint32_t c = 0; c < it.count; ++c) {
    if (!((it.mask >> c) & 1ull)) continue;  // ngsLD.cpp:270-282
    const uint32_t s2 = it.s2_begin + c;
    const double *pb = A.planes + (uint64_t)s2 * A.site_stride;
    const double m2 = A.maf[s2], mean2 = A.mean_e[s2], rsx2 = A.rsx[s2];

    // ---- pass 0: individuals with data, Pearson cross moment ----
    uint32_t x = 0;
    double sxy = 0.0;
    for (uint32_t b = (uint32_t)wave; b < n_blocks; b += 4) {
      const uint32_t i = b * 64 + (uint32_t)lane;
      const double a0 = pa[i], a1 = pa[np + i], a2 = pa[2 * np + i];
      const double b0 = pb[i], b1 = pb[np + i], b2 = pb[2 * np + i];
      const bool inb = i < A.n_ind;
      bool ok = inb;
      if (MASKED) ok = inb && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);
      x += (uint32_t)__popcll(__ballot(ok));
      const double c1 = inb ? fma(2.0, a2, a1) - mean1 : 0.0;
      const double c2 = inb ? fma(2.0, b2, b1) - mean2 : 0.0;
      sxy = fma(c1, c2, sxy);
    }
    sxy = wave_sum1(sxy);
    if (lane == 0) {
      xch0[wave][0] = sxy;
      xch0[wave][1] = (double)x;
    }
    __syncthreads();
    sxy = ((xch0[0][0] + xch0[1][0]) + xch0[2][0]) + xch0[3][0];
    x = (uint32_t)(((xch0[0][1] + xch0[1][1]) + xch0[2][1]) + xch0[3][1]);
    __syncthreads();

    // ---- haplo_freq (gen_func.cpp:1027-1059) ----
    double f0 = (1 - m1) * (1 - m2), f1 = (1 - m1) * m2, f2 = m1 * (1 - m2), f3 = m1 * m2;
    if (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1) {
      if (threadIdx.x == 0) atomicExch(A.status, (int)NGSLD_ERR_MAF_RANGE);
      f0 = f1 = f2 = f3 = __builtin_nan("");
    }
    const double inv_x = 1.0 / (double)x;
    bool bad = false, tie = false;
    uint32_t n_iter = 0;
    for (; n_iter < (uint32_t)kIterMax; ++n_iter) {
      const double p00 = f0 * f0, p01 = f0 * f1, p02 = f0 * f2, p03 = f0 * f3, p11 = f1 * f1;
      const double p12 = f1 * f2, p13 = f1 * f3, p22 = f2 * f2, p23 = f2 * f3, p33 = f3 * f3;
      const double w1 = p01 + p01, w3 = p02 + p02, w4 = 2.0 * (p03 + p12), w5 = p13 + p13, w7 = p23 + p23;
      double R0 = 0, R1 = 0, R2 = 0, R3 = 0, R4 = 0, R5 = 0, R6 = 0, R7 = 0, R8 = 0;
      // Four blocks of 64 individuals per trip: their 24 loads are in flight together, and an individual that does not count
      // (padding, no data) takes part with r = 0 instead of being branched around -- one block per trip waited an L2 round
      // trip for every 64 individuals, and its branch kept the loads of the next block behind the arithmetic of this one.
      // (Same additions in the same order: adding +0 changes nothing.)
      for (uint32_t bq = (uint32_t)wave; bq < n_blocks; bq += 16) {
        double av[4][3], bv[4][3];
        bool okv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t b = bq + 4u * (uint32_t)u;
          const bool in = b < n_blocks;
          const uint32_t i = (in ? b : bq) * 64 + (uint32_t)lane;
          av[u][0] = pa[i]; av[u][1] = pa[np + i]; av[u][2] = pa[2 * np + i];
          bv[u][0] = pb[i]; bv[u][1] = pb[np + i]; bv[u][2] = pb[2 * np + i];
          okv[u] = in && i < A.n_ind;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double a0 = av[u][0], a1 = av[u][1], a2 = av[u][2], b0 = bv[u][0], b1 = bv[u][1], b2 = bv[u][2];
          bool ok = okv[u];
          if (MASKED) ok = ok && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);
          const double v0 = fma(p11, b2, fma(w1, b1, p00 * b0));  // sum_g2 W[0][g2] b[g2]
          const double v1 = fma(w5, b2, fma(w4, b1, w3 * b0));
          const double v2 = fma(p33, b2, fma(w7, b1, p22 * b0));
          const double s = fma(a2, v2, fma(a1, v1, a0 * v0));
          const double r = ok ? rcp_refined(s) : 0.0;
          const double r0 = r * a0, r1 = r * a1, r2 = r * a2;
          R0 = fma(r0, b0, R0); R1 = fma(r0, b1, R1); R2 = fma(r0, b2, R2);
          R3 = fma(r1, b0, R3); R4 = fma(r1, b1, R4); R5 = fma(r1, b2, R5);
          R6 = fma(r2, b0, R6); R7 = fma(r2, b1, R7); R8 = fma(r2, b2, R8);
        }
      }
      double t0 = fma(p03, R4, fma(p02, R3, fma(p01, R1, p00 * R0)));
      double t1 = fma(p13, R5, fma(p12, R4, fma(p11, R2, p01 * R1)));
      double t2 = fma(p23, R7, fma(p22, R6, fma(p12, R4, p02 * R3)));
      double t3 = fma(p33, R8, fma(p23, R7, fma(p13, R5, p03 * R4)));
      wave_sum4(t0, t1, t2, t3);
      const int par = (int)(n_iter & 1u);
      if (lane == 0) {
        xch[par][wave][0] = t0; xch[par][wave][1] = t1; xch[par][wave][2] = t2; xch[par][wave][3] = t3;
      }
      __syncthreads();
      t0 = ((xch[par][0][0] + xch[par][1][0]) + xch[par][2][0]) + xch[par][3][0];
      t1 = ((xch[par][0][1] + xch[par][1][1]) + xch[par][2][1]) + xch[par][3][1];
      t2 = ((xch[par][0][2] + xch[par][1][2]) + xch[par][2][2]) + xch[par][3][2];
      t3 = ((xch[par][0][3] + xch[par][1][3]) + xch[par][2][3]) + xch[par][3][3];
      const double n0 = t0 * inv_x, n1 = t1 * inv_x, n2 = t2 * inv_x, n3 = t3 * inv_x;
      const double sn = (n0 + n1) + (n2 + n3);
      if (__builtin_amdgcn_readfirstlane((int)!(sn < 2.0))) {  // the reference's all-NaN step (see em_pair)
        bad = true;
        break;
      }
      const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
      f0 = n0; f1 = n1; f2 = n2; f3 = n3;
      if (fabs(eps - kEpsilon) < kTieMargin) tie = true;
      if (__builtin_amdgcn_readfirstlane((int)(eps < kEpsilon))) break;
    }
    if (bad) f0 = f1 = f2 = f3 = __builtin_nan("");
    if (threadIdx.x == 0)
      write_pair(A, rec0 + (uint64_t)__popcll(it.mask & ((1ull << c) - 1ull)), f0, f1, f2, f3, sxy, rsx1, rsx2, x,
                 n_iter | (tie ? kTieBit : 0u));
  }
}

// ---------------------------------------------------------------------------------------------
// Streaming kernel with the CANDIDATE's vector resident (5,121 .. 10,240 individuals; ld_pair_stream.hip).  Half of what
// the streaming kernel reads in every EM iteration never changes during a pair: the candidate site's vector b.  Eight
// wavefronts share a pair here -- wavefront w takes the 64-individual blocks w, w + 8, ... -- and with at most 20 blocks
// per wavefront b fits in registers (6 VGPRs per slot), loaded once in the pass that counts the individuals and forms the
// Pearson moment.  An iteration then reads the row vector a only: 24 bytes per individual from L2 instead of 48, half the
// load instructions, and the step itself is the streaming kernel's (same four-value form, one reciprocal per individual).
// One workgroup per item of 16 candidates, as there.
// ---------------------------------------------------------------------------------------------
template <int SLOTS, bool MASKED, bool TAIL = false>
__global__ __launch_bounds__(512, 2) void pair_ld_bres_kernel(PairArgs A) {
  constexpr int kWaves = 8;
  constexpr int kChunk = SLOTS <= 14 ? 4 : (SLOTS <= 17 ? 3 : 2), kChunks = (SLOTS + kChunk - 1) / kChunk;  // slots whose row values travel together
  __shared__ double xch[2][kWaves][4];
  __shared__ double xch0[kWaves][2];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const Item it = A.items[blockIdx.x];
  const uint32_t s1 = it.s1;
  const double m1 = A.maf[s1];
  const double mean1 = A.mean_e[s1];
  const double rsx1 = A.rsx[s1];
  const uint64_t rec0 = it.first_record - A.out_base;
  typedef const __attribute__((address_space(1))) double gdouble_t;  // global_load with an SGPR base + one 32-bit lane offset
  const uint32_t np = A.np;
  // the three plane bases of a site as wavefront-uniform pointers (see stage_pair, A_GLOBAL: left to itself the compiler keeps a
  // 64-bit VGPR address per load of the unrolled loops)
  const double *pa = A.planes + (uint64_t)s1 * A.site_stride;
  gdouble_t *pa0 = (gdouble_t *)uniform_ptr(pa), *pa1 = (gdouble_t *)uniform_ptr(pa + np), *pa2 = (gdouble_t *)uniform_ptr(pa + 2 * np);
  const uint32_t n_blocks = np / 64;  // (> 8 * (SLOTS - 1): the launcher picked SLOTS = ceil(n_blocks / 8); TAIL: > 8 * SLOTS)
  // slot j of this wavefront is block j * 8 + wave; only the last slot can lie beyond the planes (then it re-reads slot 0's
  // block and counts for nothing)
  auto index_of = [&](int j) -> uint32_t {
    const uint32_t blk = (uint32_t)(j * kWaves + wave);
    return ((TAIL || j < SLOTS - 1 || blk < n_blocks) ? blk : (uint32_t)wave) * 64u + (uint32_t)lane;
  };
  // TAIL (more than 10,240 individuals): the blocks beyond the 8 * SLOTS resident ones are streamed as in the plain kernel --
  // both vectors from memory in every iteration, two blocks per trip, after the resident slots (a fixed order of additions)
  constexpr uint32_t kTail0 = (uint32_t)(SLOTS * kWaves);
  // Every load below is SGPR base + 32-bit byte offset of the slot (one VGPR per slot, shared by the six planes).  The offsets
  // never change, and that is what has to be hidden from the compiler: loop-invariant code motion otherwise forms each load's
  // 64-bit address once, in front of the loops, where nothing folds it into the addressing mode any more -- 6 * SLOTS
  // registers of addresses, spilled and reloaded one by one in front of their loads.
  typedef const __attribute__((address_space(1))) char gchar_t;
  uint32_t off[SLOTS];
#pragma unroll
  for (int j = 0; j < SLOTS; ++j) off[j] = index_of(j) * 8u;
  auto hide_offsets = [&]() {
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) asm volatile("" : "+v"(off[j]));
  };
  auto ld = [&](gdouble_t *base, int j) -> double { return *(gdouble_t *)((gchar_t *)base + off[j]); };

  auto a_of = [&](int j, int g) -> double { return ld(g == 0 ? pa0 : (g == 1 ? pa1 : pa2), j); };

  for (uint32_t c = 0; c < it.count; ++c) {
    if (!((it.mask >> c) & 1ull)) continue;  // ngsLD.cpp:270-282
    const uint32_t s2 = it.s2_begin + c;
    const double *pb = A.planes + (uint64_t)s2 * A.site_stride;
    gdouble_t *pb0 = (gdouble_t *)uniform_ptr(pb), *pb1 = (gdouble_t *)uniform_ptr(pb + np), *pb2 = (gdouble_t *)uniform_ptr(pb + 2 * np);
    const double m2 = A.maf[s2], mean2 = A.mean_e[s2], rsx2 = A.rsx[s2];

    hide_offsets();
    // ---- pass 0: b into registers; individuals with data, Pearson cross moment (ngsLD.cpp:290: over ALL individuals) ----
    // (kChunk slots at a time, here and in the iterations: with every load of the unrolled loop hoisted to its top the row
    // vector's values alone would take 6 * SLOTS registers beside b's 6 * SLOTS)
    double bv[SLOTS][3];
    uint32_t vbits = 0, x = 0;
    double sxy = 0.0;
#pragma unroll
    for (int j0 = 0; j0 < SLOTS; j0 += kChunk) {
#pragma unroll
      for (int j = j0; j < j0 + kChunk && j < SLOTS; ++j) {
        const double a0 = a_of(j, 0), a1 = a_of(j, 1), a2 = a_of(j, 2);
        bv[j][0] = ld(pb0, j); bv[j][1] = ld(pb1, j); bv[j][2] = ld(pb2, j);
        const bool inb = TAIL || ((j < SLOTS - 1 || (uint32_t)(j * kWaves + wave) < n_blocks) && index_of(j) < A.n_ind);
        bool ok = inb;
        if (MASKED) ok = inb && !miss_data(a0, a1, a2) && !miss_data(bv[j][0], bv[j][1], bv[j][2]);
        vbits |= (ok ? 1u : 0u) << j;
        x += (uint32_t)__popcll(__ballot(ok));
        const double c1 = inb ? fma(2.0, a2, a1) - mean1 : 0.0;
        const double c2 = inb ? fma(2.0, bv[j][2], bv[j][1]) - mean2 : 0.0;
        sxy = fma(c1, c2, sxy);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (TAIL) {
      for (uint32_t blk = kTail0 + (uint32_t)wave; blk < n_blocks; blk += (uint32_t)kWaves) {
        const uint32_t i = blk * 64u + (uint32_t)lane;
        const double a0 = pa0[i], a1 = pa1[i], a2 = pa2[i], b0 = pb0[i], b1 = pb1[i], b2 = pb2[i];
        const bool inb = i < A.n_ind;
        bool ok = inb;
        if (MASKED) ok = inb && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);
        x += (uint32_t)__popcll(__ballot(ok));
        const double c1 = inb ? fma(2.0, a2, a1) - mean1 : 0.0;
        const double c2 = inb ? fma(2.0, b2, b1) - mean2 : 0.0;
        sxy = fma(c1, c2, sxy);
      }
    }
    sxy = wave_sum1(sxy);
    if (lane == 0) {
      xch0[wave][0] = sxy;
      xch0[wave][1] = (double)x;
    }
    __syncthreads();
    sxy = 0.0;
    double xs = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      sxy += xch0[w][0];
      xs += xch0[w][1];
    }
    x = (uint32_t)xs;
    __syncthreads();

    // ---- haplo_freq (gen_func.cpp:1027-1059) ----
    double f0 = (1 - m1) * (1 - m2), f1 = (1 - m1) * m2, f2 = m1 * (1 - m2), f3 = m1 * m2;
    if (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1) {
      if (threadIdx.x == 0) atomicExch(A.status, (int)NGSLD_ERR_MAF_RANGE);
      f0 = f1 = f2 = f3 = __builtin_nan("");
    }
    const double inv_x = 1.0 / (double)x;
    bool bad = false, tie = false;
    uint32_t n_iter = 0;
    // The row vector's values arrive one chunk of slots ahead of the arithmetic: chunk k + 1 is requested before chunk k is
    // worked on -- and chunk 0 of the NEXT iteration (the same values: a does not change) before this iteration's sums meet,
    // so that its round trip to L2 runs beside the reduction and the exchange instead of in front of the next step.
    double av[2][kChunk][3];
    auto fetch = [&](int k) {
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const int j = k * kChunk + u;
        if (j < SLOTS) {
          av[k & 1][u][0] = a_of(j, 0); av[k & 1][u][1] = a_of(j, 1); av[k & 1][u][2] = a_of(j, 2);
        }
      }
    };
    fetch(0);
    for (; n_iter < (uint32_t)kIterMax; ++n_iter) {
      const double p00 = f0 * f0, p01 = f0 * f1, p02 = f0 * f2, p03 = f0 * f3, p11 = f1 * f1;
      const double p12 = f1 * f2, p13 = f1 * f3, p22 = f2 * f2, p23 = f2 * f3, p33 = f3 * f3;
      const double w1 = p01 + p01, w3 = p02 + p02, w4 = 2.0 * (p03 + p12), w5 = p13 + p13, w7 = p23 + p23;
      double R0 = 0, R1 = 0, R2 = 0, R3 = 0, R4 = 0, R5 = 0, R6 = 0, R7 = 0, R8 = 0;
      // An individual that does not count -- padding, no data -- takes part with r = 0 instead of being branched around, as
      // in the streaming kernel.
      hide_offsets();
#pragma unroll
      for (int k = 0; k < kChunks; ++k) {
        if (k + 1 < kChunks) fetch(k + 1);
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
          const int j = k * kChunk + u;
          if (j < SLOTS) {
            const double a0 = av[k & 1][u][0], a1 = av[k & 1][u][1], a2 = av[k & 1][u][2];
            const double b0 = bv[j][0], b1 = bv[j][1], b2 = bv[j][2];
            const double v0 = fma(p11, b2, fma(w1, b1, p00 * b0));  // sum_g2 W[0][g2] b[g2]
            const double v1 = fma(w5, b2, fma(w4, b1, w3 * b0));
            const double v2 = fma(p33, b2, fma(w7, b1, p22 * b0));
            const double s = fma(a2, v2, fma(a1, v1, a0 * v0));
            // (every individual counts: the cohort ends inside the LAST slot -- block (n_ind - 1) / 64 is slot SLOTS - 1 of its
            // wavefront, the planes being padded to the next 64 -- or, TAIL, beyond the resident slots: no select before it)
            const double r = ((!MASKED && (TAIL || j < SLOTS - 1)) || ((vbits >> j) & 1u)) ? rcp_refined(s) : 0.0;
            const double r0 = r * a0, r1 = r * a1, r2 = r * a2;
            R0 = fma(r0, b0, R0); R1 = fma(r0, b1, R1); R2 = fma(r0, b2, R2);
            R3 = fma(r1, b0, R3); R4 = fma(r1, b1, R4); R5 = fma(r1, b2, R5);
            R6 = fma(r2, b0, R6); R7 = fma(r2, b1, R7); R8 = fma(r2, b2, R8);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (TAIL) {
        for (uint32_t bq = kTail0 + (uint32_t)wave; bq < n_blocks; bq += 2u * (uint32_t)kWaves) {
          double ta[2][3], tb[2][3];
          bool okv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const uint32_t blk = bq + (uint32_t)(u * kWaves);
            const bool in = blk < n_blocks;
            const uint32_t i = (in ? blk : bq) * 64u + (uint32_t)lane;
            ta[u][0] = pa0[i]; ta[u][1] = pa1[i]; ta[u][2] = pa2[i];
            tb[u][0] = pb0[i]; tb[u][1] = pb1[i]; tb[u][2] = pb2[i];
            okv[u] = in && i < A.n_ind;
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const double a0 = ta[u][0], a1 = ta[u][1], a2 = ta[u][2], b0 = tb[u][0], b1 = tb[u][1], b2 = tb[u][2];
            bool ok = okv[u];
            if (MASKED) ok = ok && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);
            const double v0 = fma(p11, b2, fma(w1, b1, p00 * b0));
            const double v1 = fma(w5, b2, fma(w4, b1, w3 * b0));
            const double v2 = fma(p33, b2, fma(w7, b1, p22 * b0));
            const double r = ok ? rcp_refined(fma(a2, v2, fma(a1, v1, a0 * v0))) : 0.0;
            const double r0 = r * a0, r1 = r * a1, r2 = r * a2;
            R0 = fma(r0, b0, R0); R1 = fma(r0, b1, R1); R2 = fma(r0, b2, R2);
            R3 = fma(r1, b0, R3); R4 = fma(r1, b1, R4); R5 = fma(r1, b2, R5);
            R6 = fma(r2, b0, R6); R7 = fma(r2, b1, R7); R8 = fma(r2, b2, R8);
          }
        }
      }
      fetch(0);  // for the next iteration (dropped if this one converges)
      __builtin_amdgcn_sched_barrier(0);
      double t0 = fma(p03, R4, fma(p02, R3, fma(p01, R1, p00 * R0)));
      double t1 = fma(p13, R5, fma(p12, R4, fma(p11, R2, p01 * R1)));
      double t2 = fma(p23, R7, fma(p22, R6, fma(p12, R4, p02 * R3)));
      double t3 = fma(p33, R8, fma(p23, R7, fma(p13, R5, p03 * R4)));
      wave_sum4(t0, t1, t2, t3);
      const int par = (int)(n_iter & 1u);
      if (lane == 0) {
        xch[par][wave][0] = t0; xch[par][wave][1] = t1; xch[par][wave][2] = t2; xch[par][wave][3] = t3;
      }
      lds_barrier();  // (LDS traffic only: the prefetch stays in flight)
      t0 = t1 = t2 = t3 = 0.0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {  // the same order in every wavefront: they leave the loop together
        t0 += xch[par][w][0]; t1 += xch[par][w][1]; t2 += xch[par][w][2]; t3 += xch[par][w][3];
      }
      const double n0 = t0 * inv_x, n1 = t1 * inv_x, n2 = t2 * inv_x, n3 = t3 * inv_x;
      const double sn = (n0 + n1) + (n2 + n3);
      if (__builtin_amdgcn_readfirstlane((int)!(sn < 2.0))) {  // the reference's all-NaN step (see em_pair)
        bad = true;
        break;
      }
      const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
      f0 = n0; f1 = n1; f2 = n2; f3 = n3;
      if (fabs(eps - kEpsilon) < kTieMargin) tie = true;
      if (__builtin_amdgcn_readfirstlane((int)(eps < kEpsilon))) break;
    }
    if (bad) f0 = f1 = f2 = f3 = __builtin_nan("");
    if (threadIdx.x == 0)
      write_pair(A, rec0 + (uint64_t)__popcll(it.mask & ((1ull << c) - 1ull)), f0, f1, f2, f3, sxy, rsx1, rsx2, x,
                 n_iter | (tie ? kTieBit : 0u));
  }
}
constexpr int kBresMinSlots = 11, kBresMaxSlots = 20;  // 8 wavefronts x 64 lanes x 11..20 blocks: 5,121 .. 10,240 individuals
constexpr int kBresTailSlots = 20;                      // beyond: 20 blocks per wavefront resident (10,240 individuals), the rest streamed

// host-callable launchers, defined in ld_pair_w1.hip / ld_pair_wn.hip
// Kernel families (pair_config picks by cohort size, by measurement: profiles/r03/sweep_513_1024.txt):
//   kGroup  8 / 16 / 32 lanes per pair, several pairs per wavefront in lockstep (n_ind <= 128, some shapes up to 224)
//   kRun    one wavefront per pair, the row vector shared in LDS, runs of items (n_ind <= 640: up to TEN individuals per lane)
//   kRunAB  one wavefront per pair, EM step in its a/b form, run pipeline (ld_pair_ab.hip: 641..960)
//   kMulti  2 / 4 / 8 wavefronts per pair: P form (pair_ld_kernel, 5..10 per lane, 961..5,120) or a/b form (pair_ld_abm_kernel,
//           9..15 per lane with the row slice in registers: most of 1,281..7,680 -- pair_config has the table)
//   kStream any n_ind: the candidate's vector (its first 10,240 individuals beyond that many) in registers, the row vector -- or,
//           with cfg.waves == 4 (NGSLD_PAIR_KERNEL=stream), both -- re-read every iteration
//   kHard   every likelihood triple of the matrix is a called genotype or "no data": the pairs' 16 genotype-combination
//           counts replace the individuals (any n_ind up to kHardMaxInd)
enum PairKernel { kGroup = 0, kMulti = 2, kStream = 4, kRun = 5, kHard = 6, kRunAB = 7 };
// NGSLD_PAIR_KERNEL=multi | ab | stream (tests, A/B): the multi-wavefront kernel from 513 individuals on / the a/b kernel for
// 513..1024 / the plain streaming kernel (nothing resident) beyond 5,120; abm | bres: several wavefronts per pair in the a/b
// form wherever it has a shape / never (P form up to 5,120, the streaming kernel with the candidate's vector resident beyond)
enum PairChoice { kChooseAuto = 0, kChooseMulti = 1, kChooseAB = 2, kChoosePlainStream = 3, kChooseABMulti = 4, kChooseResidentStream = 5 };
// kernels launched over runs of items (one workgroup per run, candidates addressed as 64 * item + offset)
inline bool uses_runs(int kernel) { return kernel == kRun || kernel == kGroup || kernel == kHard || kernel == kRunAB; }
constexpr uint32_t kHardMaxWords = 512;                 // row bit sets in LDS: 4 x 512 x 8 B = 16 KB
constexpr uint64_t kHardMaxInd = 64ull * kHardMaxWords;
struct PairConfig {
  int kernel;   // PairKernel
  int group;    // kGroup: lanes per pair (8, 16 or 32); 64 otherwise
  int slots;    // individuals per lane
  int waves;    // wavefronts per pair
  int form;     // kMulti: 0 = P form (pair_ld_kernel), 1 = a/b form (pair_ld_abm_kernel, ld_pair_ab.hip)
  uint32_t np;  // padded individuals per genotype plane
};
bool pair_config(uint64_t n_ind, PairConfig *cfg, int choice = kChooseAuto, bool masked = false);
// the kernel a launch really takes (a hook for families that differ with --ignore_miss_data; none does at present)
inline int effective_kernel(const PairConfig &cfg, bool /*masked*/) { return cfg.kernel; }
// ... and the shape of the multi-wavefront kernel: 2 x 10 slots (1,153..1,280 individuals) run as 4 x 5 under
// --ignore_miss_data (measured -2.6 % otherwise; both shapes read the same planes: np = 1,280)
inline void multi_shape(const PairConfig &cfg, bool masked, int *slots, int *waves) {
  *slots = cfg.slots;
  *waves = cfg.waves;
  if (masked && cfg.waves == 2 && cfg.slots == 10) {
    *slots = 5;
    *waves = 4;
  }
}
hipError_t launch_pair_kernel(const PairConfig &cfg, bool masked, const PairArgs &args, hipStream_t stream);
hipError_t launch_pair_hard(bool masked, const PairArgs &args, hipStream_t stream);  // ld_pair_hard.hip
hipError_t launch_pair_ab(int slots, bool masked, const PairArgs &args, hipStream_t stream);  // ld_pair_ab.hip
hipError_t launch_pair_abm(int slots, int waves, bool masked, const PairArgs &args, hipStream_t stream);  // ld_pair_ab.hip
hipError_t launch_pair_bres(int slots, bool masked, const PairArgs &args, hipStream_t stream);  // ld_pair_stream.hip
// Per-site classification behind kHard (ld_pair_hard.hip): masks / u as in PairArgs; *all_hard (device int, preset to 1) is
// cleared when any triple is neither a called genotype (1,0,0) / (0,1,0) / (0,0,1) nor three equal values
hipError_t launch_classify_hard(const double *planes, uint64_t site_stride, uint32_t np, uint32_t n_ind, uint64_t n_sites,
                                uint64_t *masks, double *u, int *all_hard, hipStream_t stream);
// candidate s2 sites per work item
inline uint32_t item_span(const PairConfig &cfg, uint32_t pairs_per_item) {
  if (uses_runs(cfg.kernel)) return 64u;  // run form: candidates are addressed as 64 * item + offset
  // (multi-wavefront kernel: one workgroup works through the item pair by pair; 64 candidates per item instead of 16 means a
  // quarter of the workgroups and of the per-item scalar loads: -1.3 % kernel time at n_ind 1000, -4.0 % at 2000)
  const uint32_t span = cfg.kernel == kMulti ? 4u * pairs_per_item : pairs_per_item;
  return span > 64u ? 64u : span;
}

}  // namespace ngsld
