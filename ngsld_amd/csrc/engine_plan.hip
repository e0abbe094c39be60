// engine_plan.hip -- the pair-space plan (replaces the per-s1 walk of calc_pair_LD, ngsLD.cpp:240-282, and the thread pool's
// job list, ngsLD.cpp:153-198): row ends on the host in O(n_sites), items and per-row pair counts on the device, runs.
#include "engine.h"

namespace ngsld {
namespace eng {

// The s2 walk of calc_pair_LD (ngsLD.cpp:240-262) for every s1, in O(n_sites) when the gaps are the
// positive integers read_dist produces (prefix sums are then exact and the walk is monotone);
// any other pos_dist falls back to the literal running-sum walk.
void plan_rows(const std::vector<double> &pos_dist, const std::vector<double> &maf, const ngsld_params &p,
               uint64_t n, std::vector<uint32_t> &row_end) {
  row_end.assign(n, 0);
  const bool use_dist = p.max_kb_dist > 0;
  const double limit = (double)(p.max_kb_dist * 1000);
  bool exact = true;
  std::vector<double> cum;
  std::vector<uint32_t> seg;
  if (use_dist) {
    cum.assign(n, 0.0);
    seg.assign(n, 0);
    double run = 0.0;
    uint32_t sg = 0;
    for (uint64_t s = 0; s < n; ++s) {
      const double g = pos_dist[s];
      if (std::isinf(g) && g > 0) {
        if (s > 0) ++sg;
      } else if (s > 0) {
        if (!(g >= 1.0) || g != std::floor(g) || run + g > 9.0e15) exact = false;
        run += g;
      }
      cum[s] = run;
      seg[s] = sg;
    }
  }
  uint64_t e = 0;
  for (uint64_t s1 = 0; s1 < n; ++s1) {
    uint64_t end;
    if (maf[s1] < p.min_maf) {  // ngsLD.cpp:264 (a NaN maf compares false and passes)
      end = s1 + 1;
    } else if (!use_dist) {
      end = n;
    } else if (exact) {
      if (e < s1 + 1) e = s1 + 1;
      while (e < n && seg[e] == seg[s1] && !(limit < cum[e] - cum[s1])) ++e;  // ngsLD.cpp:252
      end = e;
    } else {
      double dist = 0.0;
      end = s1 + 1;
      while (end < n) {
        dist += pos_dist[end];
        if (limit < dist) break;
        ++end;
      }
    }
    if (p.max_snp_dist > 0 && end > s1 + 1 + p.max_snp_dist) end = s1 + 1 + p.max_snp_dist;  // ngsLD.cpp:258
    if (end > n) end = n;
    row_end[s1] = (uint32_t)end;
  }
}

// Runs: a row's items cut into ceil(items / run_len) runs of near-equal length, one workgroup each.  run_len = kRunItems
// (16 items = 1,024 candidates: a whole 100 kb row) is what the pair kernel likes best in one big launch; a run that goes
// out in SMALL batches -- text batches are 2^21 pairs, i.e. only four rounds of such workgroups on 512 slots, each batch
// ending in a ragged tail -- is cut finer (ngsld_run).
//
// Tails.  A launch of equal workgroups of length L ends in a drain: the device's 2 x CUs workgroup slots finish evenly over
// the last L (2.5 ms for whole-row runs at n_ind 500), i.e. L / 2 of the whole device is lost per launch -- 1.4 ms,
// measured: 12 launches of configs[2] take 500 ms, one launch 484 (profiles/r04/sink_ab.txt).  The rows at the END of every
// launch are therefore cut into short runs (run_len / 8): as many of them as fill that triangle (CUs x one full run of
// pairs), so that every slot that falls free during the drain still finds work and all of them end within one short
// workgroup of each other.  `launch_ends` = the rows (exclusive, increasing) at which the launches this list is for end.
// (tests: NGSLD_TEST_TAIL_LEN=0 turns the shaping off, NGSLD_TEST_TAIL_PAIRS / _TAIL_LEN override its two numbers)
// (Also tried, round 4: the FIRST rows of a launch in runs of mixed lengths, so that the workgroups that start together do
// not turn over together for their first generations -- no gain, 0.9884 against 0.9894 of the device-resident rate,
// profiles/r04/sink_rr3.txt: dropped.)
int build_runs(ngsld_ctx *c, uint64_t run_len, const std::vector<uint64_t> &launch_ends) {
  run_len = std::max<uint64_t>(1, std::min<uint64_t>(run_len, kRunItems));
  if (c->run_len == run_len && c->run_ends == launch_ends) return NGSLD_OK;
  const uint64_t n = c->n_sites;
  uint64_t tail_len = std::max<uint64_t>(1, run_len / 8);
  uint64_t tail_pairs = (uint64_t)c->n_cus * run_len * item_span(c->cfg, c->pairs_per_item);
  if (const char *e = test_knob("TAIL_LEN")) tail_len = (uint64_t)std::max(0l, atol(e));
  if (const char *e = test_knob("TAIL_PAIRS")) tail_pairs = std::strtoull(e, nullptr, 10);
  std::vector<uint8_t> in_tail(n, 0);
  if (tail_len > 0 && tail_len < run_len) {
    uint64_t begin = 0;
    for (const uint64_t end : launch_ends) {
      if (end > n || end < begin) continue;
      for (uint64_t s1 = end; s1 > begin && c->h_row_off[end] - c->h_row_off[s1 - 1] <= tail_pairs; --s1) in_tail[s1 - 1] = 1;
      begin = end;
    }
  }
  std::vector<Run> runs;
  c->h_run_off.assign(n + 1, 0);
  for (uint64_t s1 = 0; s1 < n; ++s1) {
    const uint64_t i0 = c->h_item_off[s1], m = c->h_item_off[s1 + 1] - i0;
    const uint64_t len = in_tail[s1] ? tail_len : run_len;
    const uint64_t parts = (m + len - 1) / len;
    for (uint64_t q = 0; q < parts; ++q) {
      const uint64_t b = i0 + m * q / parts, e = i0 + m * (q + 1) / parts;
      runs.push_back(Run{(uint32_t)b, (uint32_t)(e - b)});
    }
    c->h_run_off[s1 + 1] = runs.size();
  }
  if (c->run_len != 0) HIP_TRY(c, hipDeviceSynchronize());  // (a launch, on whatever stream, still reading the old list)
  HIP_TRY(c, c->d_runs.resize(runs.empty() ? 1 : runs.size()));
  if (!runs.empty())
    HIP_TRY(c, hipMemcpy(c->d_runs.p, runs.data(), runs.size() * sizeof(Run), hipMemcpyHostToDevice));
  c->run_len = run_len;
  c->run_ends = launch_ends;
  return NGSLD_OK;
}
}  // namespace eng
}  // namespace ngsld

extern "C" {

int ngsld_plan(ngsld_ctx *c, const ngsld_params *p, uint64_t *n_pairs) try {
  if (c == nullptr || p == nullptr) return NGSLD_ERR_INVALID;
  if (!c->have_geno) return fail(c, NGSLD_ERR_INVALID, "no genotype data set");
  if (c->h_pos_dist.size() != c->n_sites) {
    if (p->max_kb_dist > 0)  // parse_args.cpp:174-175
      return fail(c, NGSLD_ERR_INVALID, "position file necessary in order to filter by maximum distance!");
    c->h_pos_dist.assign(c->n_sites, std::numeric_limits<double>::infinity());
  }
  if (p->min_maf < 0 || p->min_maf > 1)  // parse_args.cpp:176-177
    return fail(c, NGSLD_ERR_INVALID, "minimum allele frequency must be in [0,1]!");
  HIP_TRY(c, hipSetDevice(c->device));
  (void)hipGetLastError();  // (a failure some earlier call already reported must not surface as a launch's "last error")
  Range range_("ngsld:plan");
  {  // a run left on a caller's stream still needs the CURRENT plan (record index -> pair) for its replay
    const int rcp = finish_device_run(c);
    if (rcp != NGSLD_OK) return rcp;
  }
  const uint64_t n = c->n_sites;
  c->params = *p;
  c->planned = false;
  if (!(p->rnd_sample >= 0 && p->rnd_sample <= 1))  // parse_args.cpp:180-181 (0 is taken as "off" here)
    return fail(c, NGSLD_ERR_INVALID, "proportion of comparisons to sample must be in ]0,1]!");
  const bool sampling = p->rnd_sample > 0 && p->rnd_sample < 1;
  c->replayed_sites = 0;
  if (c->replay_on && (c->replay_read != nullptr || c->replay_matrix != nullptr) && !c->normalised) {
    // A frequency that ties --min_maf to the last bits falls on either side of `maf < min_maf` (ngsLD.cpp:264-275)
    // depending on the order est_maf adds its terms up in (the prep kernel block-reduces them), and one that sits on a
    // rounding point of the sixth decimal prints a different last digit (maf1 / maf2, ngsLD.cpp:338-339).  Such sites get
    // the reference's own sequential est_maf from the caller's raw values, and keep it for everything downstream.
    bool changed = false;
    std::vector<double> tmp;
    ReplaySite site;
    for (uint64_t s = 0; s < n; ++s) {
      const double m = c->h_maf[s], t = std::fabs(m) * 1e6;
      const bool tie = p->min_maf > 0 && std::fabs(m - p->min_maf) <= 1e-12;
      const bool edge = p->extend_out && std::fabs((t - std::floor(t)) - 0.5) < 1e-6;  // within 1e-12 of a rounding point
      if (!tie && !edge) continue;
      const int rcs = fetch_replay_site(c, s, tmp, &site);
      if (rcs != NGSLD_OK) return fail(c, rcs, "the replay source callback failed");
      ++c->replayed_sites;
      if (site.maf != m && !(site.maf != site.maf && m != m)) {
        c->h_maf[s] = site.maf;
        changed = true;
      }
    }
    if (changed) {
      HIP_TRY(c, hipMemcpy(c->d_maf.p, c->h_maf.data(), n * sizeof(double), hipMemcpyHostToDevice));
      HIP_TRY(c, launch_pack_scalars(c->d_maf.p, c->d_mean.p, c->d_rsx.p, c->d_skip.p, c->d_sc4.p, n, c->stream));
    }
  }
  plan_rows(c->h_pos_dist, c->h_maf, *p, n, c->h_row_end);
  c->h_keep.resize(n);
  for (uint64_t s = 0; s < n; ++s) c->h_keep[s] = (c->h_maf[s] < p->min_maf) ? 0 : 1;  // ngsLD.cpp:270
  const uint64_t ch = item_span(c->cfg, c->pairs_per_item);
  c->h_item_off.resize(n + 1);
  c->h_item_off[0] = 0;
  for (uint64_t s1 = 0; s1 < n; ++s1) {
    const uint64_t end = c->h_row_end[s1];
    const uint64_t span = end > s1 + 1 ? end - (s1 + 1) : 0;
    c->h_item_off[s1 + 1] = c->h_item_off[s1] + (span + ch - 1) / ch;
  }
  c->n_items = c->h_item_off[n];
  if (uses_runs(c->cfg.kernel)) {
    if (c->n_items > 0xffffffffull) return fail(c, NGSLD_ERR_UNSUPPORTED, "more than 2^32 work items in one plan");
    c->run_len = 0;  // (new items: whatever list there was is stale; the new one is cut once the rows' pair counts are known)
  }
  HIP_TRY(c, c->d_row_end.resize(n));
  HIP_TRY(c, c->d_keep.resize(n));
  HIP_TRY(c, c->d_row_off.resize(n + 1));
  HIP_TRY(c, c->d_item_off.resize(n + 1));
  HIP_TRY(c, c->d_row_count.resize(n));
  HIP_TRY(c, c->d_items.resize(c->n_items ? c->n_items : 1));
  HIP_TRY(c, hipMemcpyAsync(c->d_row_end.p, c->h_row_end.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_keep.p, c->h_keep.data(), n, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_item_off.p, c->h_item_off.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  std::vector<uint64_t> seeds;
  if (sampling) {
    // ngsLD.cpp:69-70,165-166: one master gsl_rng_taus stream, row s1's seed = (unsigned long)(uniform * 1e15),
    // drawn for s1 = 0, 1, 2, ... (serial by construction; n_sites draws)
    seeds.resize(n);
    Taus master;
    master.set(p->seed);
    for (uint64_t k = 0; k < p->first_row; ++k) master.get();  // rows that live on other GPUs
    for (uint64_t s = 0; s < n; ++s) seeds[s] = master.row_seed();
    HIP_TRY(c, c->d_row_seed.resize(n));
    HIP_TRY(c, hipMemcpyAsync(c->d_row_seed.p, seeds.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  }
  ItemArgs ia{};
  ia.row_end = c->d_row_end.p;
  ia.keep = c->d_keep.p;
  ia.row_seed = sampling ? c->d_row_seed.p : nullptr;
  ia.row_off = c->d_row_off.p;
  ia.item_off = c->d_item_off.p;
  ia.row_count = c->d_row_count.p;
  ia.items = c->d_items.p;
  ia.n_sites = (uint32_t)n;
  ia.span = (uint32_t)ch;
  ia.rnd_sample = p->rnd_sample;
  // pass 1: pairs per row (the sub-sampling makes this data dependent), prefix sum on the host
  ia.count_only = 1;
  HIP_TRY(c, launch_items(ia, c->stream));
  std::vector<uint64_t> counts(n);
  HIP_TRY(c, hipMemcpyAsync(counts.data(), c->d_row_count.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->h_row_off.resize(n + 1);
  c->h_row_off[0] = 0;
  for (uint64_t s1 = 0; s1 < n; ++s1) c->h_row_off[s1 + 1] = c->h_row_off[s1] + counts[s1];
  HIP_TRY(c, hipMemcpyAsync(c->d_row_off.p, c->h_row_off.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  // pass 2: the items (same draws again), and a host copy for the sink
  ia.count_only = 0;
  HIP_TRY(c, launch_items(ia, c->stream));
  c->h_items.clear();  // the host copy is fetched on demand by ngsld_run (the sink needs it, ngsld_run_device does not)
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (uses_runs(c->cfg.kernel)) {  // the run list of one launch over the whole plan (ngsld_run_device; ngsld_run cuts its own)
    const int rcr = build_runs(c, kRunItems, std::vector<uint64_t>{n});
    if (rcr != NGSLD_OK) return rcr;
  }
  c->planned = true;
  if (n_pairs) *n_pairs = c->h_row_off[n];
  return NGSLD_OK;
} NGSLD_CATCH(c)

int ngsld_plan_rows(ngsld_ctx *c, const uint64_t **row_off, const uint32_t **row_end) {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  if (!c->planned) return fail(c, NGSLD_ERR_INVALID, "ngsld_plan has not been called");
  if (row_off) *row_off = c->h_row_off.data();
  if (row_end) *row_end = c->h_row_end.data();
  return NGSLD_OK;
}
int ngsld_window_ends(const double *pos_dist, uint64_t n_sites, const ngsld_params *p, uint32_t *row_end) try {
  if (p == nullptr || row_end == nullptr || n_sites == 0 || n_sites >= 0xffffffffull) return NGSLD_ERR_INVALID;
  std::vector<double> pd;
  if (pos_dist == nullptr)
    pd.assign(n_sites, std::numeric_limits<double>::infinity());
  else
    pd.assign(pos_dist, pos_dist + n_sites);
  ngsld_params q = *p;
  q.min_maf = 0.0;  // the maf filters can only shorten a row
  const std::vector<double> maf(n_sites, 0.5);
  std::vector<uint32_t> ends;
  plan_rows(pd, maf, q, n_sites, ends);
  std::memcpy(row_end, ends.data(), n_sites * sizeof(uint32_t));
  return NGSLD_OK;
} NGSLD_CATCH((ngsld_ctx *)nullptr)

}  // extern "C"
