// stream.cpp -- slab streaming for matrices larger than the device budget (BASELINE configs[4]: "GL matrix exceeds
// per-GPU HBM; streamed site-window tiles").  Plain host C++ on top of the public C-ABI (include/ngsld.h): two
// contexts on one device alternate, a loader thread fills the idle one (file read -> chunked H2D -> prep kernel ->
// plan) while the calling thread runs the pair kernels of the other and feeds the sink.
//
// What it replaces in the reference: nothing -- the reference holds the whole matrix (twice during the transpose,
// ngsLD.cpp:87-89) and has no out-of-core mode.  Results are those of the resident run: a row's window
// (ngsLD.cpp:240-262) lies entirely inside its slab, per-site quantities (est_maf, expected genotypes) do not depend
// on the slab, and the --rnd_sample row seeds are taken from the master stream at the row's GLOBAL index
// (ngsld_params.first_row, ngsLD.cpp:165-166).
#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ngsld.h"
#include "host_buf.h"

namespace {

void set_err(char *err, size_t errlen, const std::string &msg) {
  if (err != nullptr && errlen > 0) std::snprintf(err, errlen, "%s", msg.c_str());
}

// The sink of one slab: same records, site indices moved from the slab's frame to the global one.
struct Rebase {
  uint64_t base;
  ngsld_sink_fn sink;
  void *user;
  std::vector<ngsld_item> items;
};

int rebase_sink(void *user, const ngsld_batch *b) {
  Rebase *r = static_cast<Rebase *>(user);
  r->items.assign(b->items, b->items + b->n_items);
  for (ngsld_item &it : r->items) {
    it.s1 += (uint32_t)r->base;
    it.s2_begin += (uint32_t)r->base;
  }
  ngsld_batch g = *b;
  g.s1_begin += r->base;
  g.s1_end += r->base;
  g.items = r->items.data();
  return r->sink(r->user, &g);
}

}  // namespace

namespace {
std::mutex g_streamed_mu;
ngsld_replay_stats_t g_streamed_replay{};  // the last streamed job of this process, summed over its slabs
}  // namespace

extern "C" {

int ngsld_plan_slabs(const double *pos_dist, uint64_t n_sites, const ngsld_params *params, uint64_t max_slab_sites,
                     ngsld_slab *slabs, uint64_t cap, uint64_t *n_slabs) {
  if (params == nullptr || slabs == nullptr || n_slabs == nullptr || n_sites == 0 || max_slab_sites == 0)
    return NGSLD_ERR_INVALID;
  std::vector<uint32_t> row_end;
  try {
    row_end.resize(n_sites);
  } catch (const std::bad_alloc &) {
    return NGSLD_ERR_NOMEM;
  }
  const int rc = ngsld_window_ends(pos_dist, n_sites, params, row_end.data());
  if (rc != NGSLD_OK) return rc;
  uint64_t k = 0, r0 = 0;
  while (r0 < n_sites) {
    uint64_t r1 = r0, hi = r0;
    while (r1 < n_sites) {
      const uint64_t h = std::max<uint64_t>(hi, std::max<uint64_t>(row_end[r1], r1 + 1));
      if (h - r0 > max_slab_sites) break;
      hi = h;
      ++r1;
    }
    if (r1 == r0) return NGSLD_ERR_NOMEM;  // one row's window alone exceeds the slab
    if (k == cap) return NGSLD_ERR_INVALID;
    slabs[k++] = ngsld_slab{r0, r1, hi};
    r0 = r1;
  }
  *n_slabs = k;
  return NGSLD_OK;
}

int ngsld_run_streamed(int device, uint64_t n_sites, uint64_t n_ind, const double *pos_dist,
                       const ngsld_params *params, const ngsld_geno_opts *opts, uint64_t max_slab_sites,
                       ngsld_read_sites_fn read, void *read_user, double *maf_out, ngsld_sink_fn sink,
                       void *sink_user, uint64_t *n_pairs, uint64_t *n_slabs_out, char *err, size_t errlen) {
  return ngsld_run_streamed_text(device, n_sites, n_ind, pos_dist, params, opts, max_slab_sites, read, read_user, maf_out,
                                 sink, sink_user, n_pairs, n_slabs_out, err, errlen, nullptr, 0);
}

int ngsld_run_streamed_text(int device, uint64_t n_sites, uint64_t n_ind, const double *pos_dist,
                            const ngsld_params *params, const ngsld_geno_opts *opts, uint64_t max_slab_sites,
                            ngsld_read_sites_fn read, void *read_user, double *maf_out, ngsld_sink_fn sink,
                            void *sink_user, uint64_t *n_pairs, uint64_t *n_slabs_out, char *err, size_t errlen,
                            const char *const *labels, int text_output) {
  if (params == nullptr || opts == nullptr || read == nullptr || sink == nullptr || n_sites == 0 || n_ind == 0) {
    set_err(err, errlen, "invalid argument");
    return NGSLD_ERR_INVALID;
  }
  if (opts->on_device) {
    set_err(err, errlen, "a streamed run reads host memory");
    return NGSLD_ERR_INVALID;
  }
  std::vector<ngsld_slab> slabs;
  std::vector<uint64_t> slab_pairs;
  try {
    slabs.resize(n_sites);
  } catch (const std::bad_alloc &) {
    set_err(err, errlen, "cannot allocate the slab table");
    return NGSLD_ERR_NOMEM;
  }
  uint64_t n_slabs = 0;
  int rc = ngsld_plan_slabs(pos_dist, n_sites, params, max_slab_sites, slabs.data(), slabs.size(), &n_slabs);
  if (rc == NGSLD_ERR_NOMEM) {
    set_err(err, errlen,
            "the window of a single site does not fit the device memory budget (all-pairs runs cannot be streamed: "
            "set --max_kb_dist / --max_snp_dist or raise the budget)");
    return rc;
  }
  if (rc != NGSLD_OK) {
    set_err(err, errlen, "cannot plan the slabs");
    return rc;
  }
  slabs.resize(n_slabs);
  if (n_slabs_out) *n_slabs_out = n_slabs;
  uint64_t max_sites = 0;
  for (const ngsld_slab &s : slabs) max_sites = std::max(max_sites, s.site_end - s.row_begin);

  ngsld_ctx *ctx[2] = {nullptr, nullptr};
  ngsld::HostMatrix host[2];  // (host_buf.h: no zero-fill, huge pages)
  for (int k = 0; k < 2; ++k) {
    if (k == 1 && n_slabs < 2) break;
    rc = ngsld_create(device, &ctx[k]);
    if (rc != NGSLD_OK) {
      set_err(err, errlen, ngsld_last_error(nullptr));
      if (ctx[0]) ngsld_destroy(ctx[0]);
      return rc;
    }
    if (!host[k].alloc((size_t)(max_sites * n_ind * 3))) {
      set_err(err, errlen, "cannot allocate the host slab buffer");
      for (int q = 0; q <= k; ++q) ngsld_destroy(ctx[q]);
      return NGSLD_ERR_NOMEM;
    }
  }

  // slot state: 0 = free (the loader may fill it), 1 = loaded and planned (the runner may compute it)
  std::mutex mu;
  std::condition_variable cv;
  int state[2] = {0, 0};
  bool stop = false;
  int load_rc = NGSLD_OK;
  std::string load_msg;
  slab_pairs.assign(n_slabs, 0);  // (n_slabs <= n_sites words: not guarded separately)

  std::thread loader([&]() {
    uint64_t maf_done = 0;  // maf_out[0, maf_done) is final
    int hard_job = -1;      // kernel family of the job, fixed by its first slab: 1 = genotype-combination kernel, 0 = per individual
    for (uint64_t k = 0; k < n_slabs; ++k) {
      const int b = (int)(k & 1);
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return state[b] == 0 || stop; });
        if (stop) return;
      }
      const ngsld_slab &sl = slabs[k];
      const uint64_t m = sl.site_end - sl.row_begin;
      int r = NGSLD_OK;
      std::string msg;
      try {  // host allocations below: an exception must not leave this thread
      if (read(read_user, sl.row_begin, m, host[b].data()) != 0) {
        r = NGSLD_ERR_INVALID;
        msg = "cannot read the genotype data of a slab";
      }
      if (r == NGSLD_OK) {
        // One kernel family for the whole job: a slab that happens to hold only called genotypes must not switch to the
        // genotype-combination kernel while its neighbours run the per-individual one (same values to 1e-12, not the
        // same bits).  Known in advance for one kind of job: --call_geno with N_thresh == call_thresh (the default, 0 and
        // 0) leaves every triple either called or "no data" (gen_func.cpp:886-914: below N_thresh -> missing, at or
        // above call_thresh -> called, nothing in between), so every slab qualifies, as the resident run does.
        // "Every slab qualifies" is checked, not assumed: a triple the classification rejects after all (a NaN under text
        // semantics sets no NaN status) puts ITS slab on the per-individual kernels.  The first slab fixes the job's family;
        // if it does not qualify, every later slab is told to stay per individual too.  A later slab that disagrees with a
        // first slab that did qualify -- its predecessors' rows are out already -- is loaded again for the per-individual
        // kernels, and so are the slabs after it: their records are the ones the resident run would give (which runs per
        // individual as a whole, holding that triple), the earlier slabs' agree with them to 1e-12 instead of bit for bit.
        // (Up to round 5 such a job ended with an error and a truncated table.)
        ngsld_geno_opts so = *opts;
        if (!(opts->call_geno && opts->N_thresh == opts->call_thresh) || hard_job == 0) so.per_individual_only = 1;
        r = ngsld_set_geno_raw_opts(ctx[b], host[b].data(), m, n_ind, &so);
        if (r == NGSLD_OK && !so.per_individual_only) {
          const bool is_hard = std::strcmp(ngsld_pair_kernel(ctx[b]), "hard") == 0;
          if (hard_job < 0) {
            hard_job = is_hard ? 1 : 0;
          } else if (hard_job == 1 && !is_hard) {
            hard_job = 0;
            so.per_individual_only = 1;
            r = ngsld_set_geno_raw_opts(ctx[b], host[b].data(), m, n_ind, &so);
          }
        }
        // exact-order replay: the slab's raw values stay in host[b] until its run is over
        if (r == NGSLD_OK) r = ngsld_set_replay_matrix(ctx[b], host[b].data());
        if (r == NGSLD_OK) r = ngsld_set_pos_dist(ctx[b], pos_dist ? pos_dist + sl.row_begin : nullptr);
        if (r == NGSLD_OK) {
          ngsld_params p = *params;
          p.first_row = params->first_row + sl.row_begin;
          uint64_t all_rows = 0;  // includes the halo rows, which the next slab computes
          r = ngsld_plan(ctx[b], &p, &all_rows);
          if (r == NGSLD_OK && maf_out != nullptr && sl.site_end > maf_done) {
            // (after the plan: a frequency that ties --min_maf has been settled by then.)  Only the sites no earlier slab
            // delivered: entries a sink may be reading are never rewritten
            std::vector<double> maf(m);
            r = ngsld_get_maf(ctx[b], maf.data());
            if (r == NGSLD_OK) {
              const uint64_t from = std::max(maf_done, sl.row_begin);
              std::memcpy(maf_out + from, maf.data() + (from - sl.row_begin), (sl.site_end - from) * sizeof(double));
              maf_done = sl.site_end;
            }
          }
          const uint64_t *row_off = nullptr;
          if (r == NGSLD_OK) r = ngsld_plan_rows(ctx[b], &row_off, nullptr);
          if (r == NGSLD_OK) slab_pairs[k] = row_off[sl.row_end - sl.row_begin];
          // device-side TSV: the slab's sites carry their own labels (set after the matrix, which resets it)
          if (r == NGSLD_OK && text_output) r = ngsld_set_text_output(ctx[b], labels ? labels + sl.row_begin : nullptr, 1);
        }
        if (r != NGSLD_OK && msg.empty()) msg = ngsld_last_error(ctx[b]);
      }
      } catch (...) {
        r = NGSLD_ERR_NOMEM;
        msg = "out of host memory while loading a slab";
      }
      std::lock_guard<std::mutex> lk(mu);
      if (r != NGSLD_OK) {
        load_rc = r;
        load_msg = msg;
        stop = true;
        cv.notify_all();
        return;
      }
      state[b] = 1;
      cv.notify_all();
    }
  });

  uint64_t total = 0;
  int run_rc = NGSLD_OK;
  std::string run_msg;
  {
    std::lock_guard<std::mutex> lk(g_streamed_mu);
    g_streamed_replay = ngsld_replay_stats_t{};
  }
  Rebase rb{0, sink, sink_user, {}};
  for (uint64_t k = 0; k < n_slabs; ++k) {
    const int b = (int)(k & 1);
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return state[b] == 1 || stop; });
      if (stop) break;
    }
    rb.base = slabs[k].row_begin;
    run_rc = ngsld_run(ctx[b], 0, slabs[k].row_end - slabs[k].row_begin, rebase_sink, &rb);
    if (run_rc != NGSLD_OK) run_msg = ngsld_last_error(ctx[b]);
    {  // where the slab's flagged pairs were replayed, added up over the job (ngsld_streamed_replay_info)
      ngsld_replay_stats_t st;
      if (run_rc == NGSLD_OK && ngsld_replay_info(ctx[b], &st) == NGSLD_OK) {
        std::lock_guard<std::mutex> lk(g_streamed_mu);
        g_streamed_replay.pairs_flagged += st.pairs_flagged;
        g_streamed_replay.pairs_replayed += st.pairs_replayed;
        g_streamed_replay.pairs_on_device += st.pairs_on_device;
        g_streamed_replay.pairs_on_host += st.pairs_on_host;
        g_streamed_replay.sites_reevaluated += st.sites_reevaluated;
        g_streamed_replay.sites_degenerate += st.sites_degenerate;
        g_streamed_replay.exact_store_build_s += st.exact_store_build_s;
        g_streamed_replay.exact_store = std::max(g_streamed_replay.exact_store, st.exact_store);
        g_streamed_replay.text_rows_patched += st.text_rows_patched;
      }
    }
    total += slab_pairs[k];
    std::lock_guard<std::mutex> lk(mu);
    if (run_rc != NGSLD_OK) stop = true;
    state[b] = 0;
    cv.notify_all();
    if (stop) break;
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    if (run_rc != NGSLD_OK) stop = true;
    cv.notify_all();
  }
  loader.join();
  for (int k = 0; k < 2; ++k)
    if (ctx[k]) ngsld_destroy(ctx[k]);
  if (n_pairs) *n_pairs = total;
  if (run_rc != NGSLD_OK) {
    set_err(err, errlen, run_msg);
    return run_rc;
  }
  if (load_rc != NGSLD_OK) {
    set_err(err, errlen, load_msg);
    return load_rc;
  }
  return NGSLD_OK;
}

int ngsld_streamed_replay_info(ngsld_replay_stats_t *out) {
  if (out == nullptr) return NGSLD_ERR_INVALID;
  std::lock_guard<std::mutex> lk(g_streamed_mu);
  *out = g_streamed_replay;
  return NGSLD_OK;
}

}  // extern "C"
