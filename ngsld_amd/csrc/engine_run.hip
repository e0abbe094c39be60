// engine_run.hip -- ngsld_run / ngsld_run_device: the batch pipeline pair kernel -> (device-side replay) -> records or
// device-side TSV -> sink (replaces threadpool_add(calc_pair_LD) ... threadpool_wait and the fprintf block,
// ngsLD.cpp:153-198, 310-352).
#include "engine.h"
#include "../../include/ngsld_host.h"

namespace ngsld {
namespace eng {

hipError_t timed_launch(ngsld_ctx *c, const PairArgs &a, hipStream_t stream) {
  if (c->ev_used == c->ev_pool.size()) {
    hipEvent_t b, e;
    hipError_t r = hipEventCreate(&b);
    if (r != hipSuccess) return r;
    r = hipEventCreate(&e);
    if (r != hipSuccess) return r;
    c->ev_pool.emplace_back(b, e);
  }
  auto &ev = c->ev_pool[c->ev_used++];
  hipError_t r = hipEventRecord(ev.first, stream);
  if (r != hipSuccess) return r;
  r = launch_pair_kernel(c->cfg, c->params.ignore_miss_data != 0, a, stream);
  if (r != hipSuccess) return r;
  return hipEventRecord(ev.second, stream);
}

PairArgs make_args(ngsld_ctx *c, uint64_t r0, uint64_t r1, ngsld_rec_std *d_std, ngsld_rec_ext *d_ext, uint32_t *d_flags,
                   uint32_t flag_cap, uint64_t flag_n) {
  PairArgs a{};
  a.flags = d_flags;
  a.flags_host = d_flags != nullptr ? d_flags + flag_head_words(flag_cap) + flag_bitmap_words(flag_n) : nullptr;
  a.flag_cap = flag_cap;
  a.flag_text = 1;
  a.pearson_on_device = lkl_device_eligible(c) ? 1 : 0;
  // the pairs of degenerate sites (sc4[.][3]) skip their EM: every one of them is flagged and the exact-order replay is their
  // only evaluation -- on while the device-side replay of likelihood matrices can take them
  a.skip_degenerate = (d_flags != nullptr && c->h_skip_count > 0 && c->skip_on && lkl_device_eligible(c) && !c->exact_failed) ? 1 : 0;
  a.planes = c->d_planes.p;
  a.site_stride = 3ull * c->np;
  a.np = c->np;
  a.n_ind = (uint32_t)c->n_ind;
  a.inv_n = 1.0 / (double)c->n_ind;
  a.maf = c->d_maf.p;
  a.mean_e = c->d_mean.p;
  a.rsx = c->d_rsx.p;
  a.items = c->d_items.p + c->h_item_off[r0];
  a.n_items = c->h_item_off[r1] - c->h_item_off[r0];
  if (uses_runs(c->cfg.kernel)) {
    a.runs = c->d_runs.p + c->h_run_off[r0];
    a.n_runs = c->h_run_off[r1] - c->h_run_off[r0];
  }
  a.hard_masks = c->d_hard_masks.p;
  a.hard_u = c->d_hard_u.p;
  a.mask_words = c->mask_words;
  a.items_all = c->d_items.p;
  a.item_off = c->d_item_off.p;
  a.h_item_off = c->h_item_off.data();
  a.row0 = (uint32_t)r0;
  a.row1 = (uint32_t)r1;
  a.planes_bytes = c->n_sites * 3ull * c->np * sizeof(double);
  a.sc4 = c->d_sc4.p;
  a.out_base = c->h_row_off[r0];
  a.out_std = d_std;
  a.out_ext = d_ext;
  a.status = c->d_status.p;
  return a;
}
}  // namespace eng
}  // namespace ngsld

extern "C" {

int ngsld_set_text_output(ngsld_ctx *c, const char *const *labels, int enable) try {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  if (!c->have_geno) return fail(c, NGSLD_ERR_INVALID, "set the genotype data before the labels");
  HIP_TRY(c, hipSetDevice(c->device));
  (void)hipGetLastError();  // (a failure some earlier call already reported must not surface as a launch's "last error")
  c->text_mode = false;
  if (!enable) return NGSLD_OK;
  c->have_labels = labels != nullptr;
  c->max_label = 6;
  if (labels != nullptr) {
    std::vector<uint64_t> off(c->n_sites + 1, 0);
    for (uint64_t s = 0; s < c->n_sites; ++s) {
      if (labels[s] == nullptr) return fail(c, NGSLD_ERR_INVALID, "a label is NULL");
      const uint64_t n = std::strlen(labels[s]);
      off[s + 1] = off[s] + n;
      c->max_label = std::max<uint64_t>(c->max_label, n);
    }
    std::vector<char> blob(off[c->n_sites] ? off[c->n_sites] : 1);
    for (uint64_t s = 0; s < c->n_sites; ++s) std::memcpy(blob.data() + off[s], labels[s], off[s + 1] - off[s]);
    HIP_TRY(c, c->d_labels.resize(blob.size()));
    HIP_TRY(c, c->d_label_off.resize(c->n_sites + 1));
    HIP_TRY(c, hipMemcpy(c->d_labels.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_label_off.p, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  }
  c->text_mode = true;
  return NGSLD_OK;
} NGSLD_CATCH(c)

int ngsld_reserve_text_buffers(ngsld_ctx *c, uint64_t bytes_per_row) try {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  if (c->reserve_thread.joinable()) c->reserve_thread.join();
  if (bytes_per_row == 0) return NGSLD_OK;
  const uint64_t bytes_per_batch = bytes_per_row * std::min<uint64_t>(c->batch_pairs, kTextBatchPairs);
  c->reserve_thread = std::thread([c, bytes_per_batch] {
    if (hipSetDevice(c->device) != hipSuccess) return;
    for (int k = 0; k < ngsld_ctx::kSlots; ++k) (void)c->h_text[k].resize(bytes_per_batch);  // (a failure here is found again, and reported, at first use)
  });
  return NGSLD_OK;
} NGSLD_CATCH(c)
int ngsld_run_device(ngsld_ctx *c, uint64_t s1_begin, uint64_t s1_end, void *d_std, void *d_ext, void *hip_stream) try {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  if (!c->planned) return fail(c, NGSLD_ERR_INVALID, "ngsld_plan has not been called");
  if (s1_begin > s1_end || s1_end > c->n_sites) return fail(c, NGSLD_ERR_INVALID, "row range out of bounds");
  if (d_std == nullptr) return fail(c, NGSLD_ERR_INVALID, "d_std is NULL");
  HIP_TRY(c, hipSetDevice(c->device));
  (void)hipGetLastError();  // (a failure some earlier call already reported must not surface as a launch's "last error")
  Range range_("ngsld:run_device (pair kernels)");
  {
    // One pending run per context: the flag buffer and the record pointers of a run on a caller's stream are single.  A second
    // run before ngsld_finish_device first finishes the earlier one (waits for its stream, replays what it flagged) -- clearing
    // the flags under kernels still setting them would leave those records with the kernels' unreplayed values.
    const int rcp = finish_device_run(c);
    if (rcp != NGSLD_OK) return rcp;
  }
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
  c->ev_used = 0;
  c->timed_stream = st;
  c->timed_overlap = false;
  c->timed_pairs = c->h_row_off[s1_end] - c->h_row_off[s1_begin];
  c->replayed_pairs = 0;
  c->replayed_on_device = 0;
  c->flagged_pairs = 0;
  c->text_rows_patched = 0;
  if (c->replay_on) {
    c->flag_cap_dev = flag_cap_for(c->timed_pairs);
    const int rcf = reset_flags(c, c->d_flags_dev, c->timed_pairs, c->flag_cap_dev, st);
    if (rcf != NGSLD_OK) return rcf;
    HIP_TRY(c, c->h_flags_dev.resize(flag_head_words(c->flag_cap_dev)));
  }
  // one launch per <= 2^31-1 workgroups; rows are cut so that each launch's grid fits
  const uint64_t max_items = 0x7ffffff0ull;
  std::vector<uint64_t> cuts;  // rows at which the launches end
  for (uint64_t r0 = s1_begin; r0 < s1_end;) {
    uint64_t r1 = r0 + 1;
    while (r1 < s1_end && c->h_item_off[r1 + 1] - c->h_item_off[r0] <= max_items) ++r1;
    cuts.push_back(r1);
    r0 = r1;
  }
  if (uses_runs(c->cfg.kernel)) {  // (big launches: whole-row runs, whatever an earlier ngsld_run cut them to, short ones at each launch's end)
    const int rcr = build_runs(c, kRunItems, cuts);
    if (rcr != NGSLD_OK) return rcr;
  }
  uint64_t r0 = s1_begin;
  for (const uint64_t r1 : cuts) {
    PairArgs a = make_args(c, r0, r1, (ngsld_rec_std *)d_std, (ngsld_rec_ext *)d_ext, c->replay_on ? c->d_flags_dev.p : nullptr,
                           c->flag_cap_dev, c->timed_pairs);
    a.out_base = c->h_row_off[s1_begin];
    // these records stay on the device: only numerically ill-conditioned pairs are replayed -- unless they are the records of
    // a group of text batches (run_grouped below): then also the pairs whose printed digits rounding could change
    a.flag_text = c->dev_run_flag_text ? 1 : 0;
    HIP_TRY(c, timed_launch(c, a, st));
    r0 = r1;
  }
  c->dev_run.dev_applied = false;
  if (c->replay_on) {
    int rcd = device_replay(c, c->d_flags_dev.p, c->flag_cap_dev, c->h_row_off[s1_begin], c->timed_pairs,
                            (ngsld_rec_std *)d_std, (ngsld_rec_ext *)d_ext, st, -1);
    if (rcd != NGSLD_OK) return rcd;
    // likelihood matrices: right behind the kernels when the exact store is there (or costs nothing); a run that turns out to
    // flag many pairs without one has it built in ngsld_finish_device
    if (lkl_device_eligible(c) && (exact_store_started(c) || exact_store_is_free(c))) {
      rcd = try_device_replay_lkl(c, c->d_flags_dev.p, c->flag_cap_dev, c->h_row_off[s1_begin], c->timed_pairs,
                                  (ngsld_rec_std *)d_std, (ngsld_rec_ext *)d_ext, st, c->dev_run_flag_text, -1, &c->dev_run.dev_applied,
                                  c->dev_run_flag_text ? exact_sites_needed(c, s1_begin, s1_end) : c->n_sites);
      if (rcd != NGSLD_OK) return rcd;
    }
    // which pairs the kernels flagged (and the device has not settled itself): the counter and the list come over behind
    // them, on their stream
    rcd = send_flag_head(c, c->d_flags_dev.p, c->h_flags_dev.p, c->flag_cap_dev, st, !c->dev_run.dev_applied);
    if (rcd != NGSLD_OK) return rcd;
  }
  c->dev_run.pending = true;
  c->dev_run.s1_begin = s1_begin;
  c->dev_run.s1_end = s1_end;
  c->dev_run.d_std = (ngsld_rec_std *)d_std;
  c->dev_run.d_ext = (ngsld_rec_ext *)d_ext;
  c->dev_run.st = st;
  if (hip_stream == nullptr) {
    const int rcd = finish_device_run(c);  // waits for the kernels, replays what they flagged
    if (rcd != NGSLD_OK) return rcd;
    return check_status(c);
  }
  return NGSLD_OK;  // (the caller's stream: the records are final after ngsld_finish_device)
} NGSLD_CATCH(c)

}  // extern "C"

namespace ngsld {
namespace eng {
// (a build of the exact store a run starts has ended when the run returns, whichever way: the registered source is read
// during runs only -- include/ngsld.h, BUFFER LIFETIME)
struct StoreGuard {
  ngsld_ctx *c;
  ~StoreGuard() {
    std::unique_lock<std::mutex> lk(c->exact_mu);
    while (c->exact_state.load() == 1) c->exact_cv.wait_for(lk, std::chrono::milliseconds(1));
  }
};

// Device-side TSV: the dist column needs prefix sums of pos_dist that are EXACT (the host writer adds the gaps one by one,
// ngsLD.cpp:241), i.e. integer gaps as read_dist produces them; otherwise (*text = false) the batches go out as records.
static int upload_text_prefix(ngsld_ctx *c, bool *text) {
  const uint64_t n = c->n_sites;
  std::vector<double> cum(n);
  std::vector<uint32_t> infc(n);
  double run = 0.0;
  uint32_t ic = 0;
  for (uint64_t s = 0; s < n && *text; ++s) {
    const double g = c->h_pos_dist[s];
    if (std::isinf(g) && g > 0) {
      ++ic;
    } else {
      if (!(g >= 0.0) || g != std::floor(g) || run + g > 9.0e15) *text = false;
      run += g;
    }
    cum[s] = run;
    infc[s] = ic;
  }
  if (*text) {
    HIP_TRY(c, c->d_cum.resize(n));
    HIP_TRY(c, c->d_infc.resize(n));
    HIP_TRY(c, hipMemcpy(c->d_cum.p, cum.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_infc.p, infc.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  return NGSLD_OK;
}

// Rows [s1_begin, s1_end) through the batch pipeline: pair kernels, replay, text or records, sink -- batch by batch.
static int run_range(ngsld_ctx *c, uint64_t s1_begin, uint64_t s1_end, ngsld_sink_fn sink, void *user) {
  const bool ext = c->params.extend_out != 0;
  c->ev_used = 0;
  c->timed_stream = c->stream;
  c->timed_pairs = c->h_row_off[s1_end] - c->h_row_off[s1_begin];
  c->replayed_pairs = 0;
  c->replayed_on_device = 0;
  c->flagged_pairs = 0;
  c->text_rows_patched = 0;
  const bool replay = c->replay_on;
  if (c->reserve_thread.joinable()) c->reserve_thread.join();  // (ngsld_reserve_text_buffers: h_text[] is this thread's again)

  bool text = c->text_mode;
  if (text) {
    const int rct = upload_text_prefix(c, &text);
    if (rct != NGSLD_OK) return rct;
  }
  auto need_host_items = [&]() -> int { return ensure_host_items(c); };
  if (!text) {  // (record batches carry their items to the sink; text batches need none -- the replay finds its pairs on the device)
    const int rc0 = need_host_items();
    if (rc0 != NGSLD_OK) return rc0;
  }
  // How the batches flow: two slots, the kernel of batch k + 1 runs while batch k is consumed.  Text: the rows are formatted
  // on the device and copied.  Records: the pair kernels write them straight into the slot's pinned host buffers
  // (run_direct; nothing is left to copy behind the last kernel; twice the pairs per batch, half the launches), or into
  // device buffers with a D2H copy per batch, the batches then shrinking towards the end of the run (run_taper).
  // NGSLD_TEST_RUN_STREAMS=2: three slots, two compute streams half a batch out of phase (see ngsld_ctx).
  const bool direct = !text && c->run_direct;
  // Text batches are small (2^19 rows: a 2.8 ms pair kernel, a tenth of it ramp and drain) and many: for them the two compute
  // streams half a batch out of phase DO pay, on every box -- while one stream's kernel drains the other's is in full
  // flight: configs[2]'s loop 0.58-0.63 -> 0.546-0.551 s (profiles/r04/e2e_text_streams.txt).  (tests: NGSLD_TEST_TEXT_STREAMS=1: one stream.)
  bool text_two = true;
  if (const char *e = test_knob("TEXT_STREAMS")) text_two = std::atoi(e) != 1;
  const bool two_streams = text ? text_two : c->run_streams == 2;
  c->timed_overlap = two_streams;  // (ngsld_last_kernel_time: launches on two streams share the device -- first start .. last end)
  // (text on ONE stream with three slots, two batches queued ahead, measured no different from two slots: the compute stream
  // does not run dry, profiles/r04/e2e_timeline.txt)
  const int S = two_streams ? ngsld_ctx::kSlots : 2;
  struct Batch {
    uint64_t r0, r1, n;
  };
  std::vector<Batch> batches;
  // text batches are cut sixteen times finer: smaller batches mean smaller pinned buffers and a finer kernel / copy overlap
  // (kTextBatchPairs; round 1, configs[2] end to end: 2^23 pairs per batch 2.2 s, 2^21 1.5 s)
  // (records written by the kernels themselves: every launch costs ~0.4 ms of drain and nothing has to be staged on the
  // device, so the batches are twice the size -- 2 x 1.2 GB of pinned host memory with the extended record)
  const uint64_t text_batch = kTextBatchPairs;
  uint64_t batch_pairs = text ? std::min<uint64_t>(c->batch_pairs, text_batch)
                              : ((direct && !c->batch_pairs_set) ? 2 * c->batch_pairs : c->batch_pairs);
  const bool taper = !text && !direct && c->run_taper;
  uint64_t cap = 1, last_cap = 0;
  for (;;) {  // (a second trip only when the pinned record buffers of this batch size cannot be had: half the size then)
    batches.clear();
    uint64_t left = c->timed_pairs;
    for (uint64_t r0 = s1_begin; r0 < s1_end;) {
      uint64_t target = batch_pairs;
      if (two_streams && batches.empty()) target = batch_pairs / 2;  // (the phase shift between the two streams)
      if (taper) target = std::min<uint64_t>(batch_pairs, std::max<uint64_t>(left / 3, std::min<uint64_t>(batch_pairs, 1ull << 19)));
      uint64_t r1 = r0 + 1;
      while (r1 < s1_end && c->h_row_off[r1 + 1] - c->h_row_off[r0] <= target) ++r1;
      batches.push_back({r0, r1, c->h_row_off[r1] - c->h_row_off[r0]});
      left -= std::min(left, c->h_row_off[r1] - c->h_row_off[r0]);
      r0 = r1;
    }
    cap = 1;
    for (auto &b : batches) cap = std::max(cap, b.n);
    if (text) break;
    // the batches' host buffers: pinned memory is the scarce kind -- a host that cannot pin two (three) buffers of this size
    // gets batches of half the size instead of an error, down to 2^20 pairs
    hipError_t e = hipSuccess;
    if (const char *lim = test_knob("PIN_LIMIT_BYTES"))  // tests: a host that cannot pin more than this per buffer
      if (cap * sizeof(ngsld_rec_std) > std::strtoull(lim, nullptr, 10)) e = hipErrorOutOfMemory;
    for (int k = 0; k < S && e == hipSuccess; ++k) {
      e = c->h_std[k].resize(cap);
      if (e == hipSuccess && ext) e = c->h_ext[k].resize(cap);
    }
    if (e == hipSuccess) break;
    (void)hipGetLastError();
    // (a batch holds at least one row: once the largest row is what sets `cap`, halving the target changes nothing)
    if (cap <= (1ull << 16) || batches.size() >= (1u << 20) || cap == last_cap || batch_pairs <= 1)
      return hip_fail(c, e, "pinned host buffers of a record batch");
    last_cap = cap;
    batch_pairs = std::min(batch_pairs, cap);  // (a run smaller than a batch: halve what it actually needed)
    for (int k = 0; k < S; ++k) {
      c->h_std[k].release();
      c->h_ext[k].release();
    }
    batch_pairs /= 2;
  }
  if (uses_runs(c->cfg.kernel)) {
    // every batch should be thousands of workgroups (512 run at a time): the smaller the batches, the shorter the runs.
    // configs[2] as text (48 batches of 2^21 pairs): 16 items per run 0.82 s for this loop, 8 0.72 s, 4 0.70 s
    uint64_t want = kRunItems;
    while (want > 2 && want * item_span(c->cfg, c->pairs_per_item) * 8192 > batch_pairs) want /= 2;
    std::vector<uint64_t> ends;  // every batch is a launch: its last rows go out as short runs (build_runs)
    for (auto &b : batches) ends.push_back(b.r1);
    const int rcr = build_runs(c, want, ends);
    if (rcr != NGSLD_OK) return rcr;
  }
  for (int k = 0; k < S; ++k) {
    if (!direct) {
      HIP_TRY(c, c->d_std[k].resize(cap));
      if (ext) HIP_TRY(c, c->d_ext[k].resize(cap));
    }
    if (text) {
      HIP_TRY(c, c->d_lens[k].resize(cap));
      HIP_TRY(c, c->d_offs[k].resize(cap));
      HIP_TRY(c, c->d_text_meta[k].resize(4));  // {total bytes, needs_host, a replayed row changed its length, the early write pass ran out of room}
      HIP_TRY(c, c->h_text_meta[k].resize(4));
      // the batch's rows are WRITTEN right behind their lengths, on the compute stream, before the host knows how long the
      // text is (see issue): room for the usual row -- a batch that needs more is written again once its length is known
      const uint64_t row_guess = 2 * c->max_label + (ext ? 200 : 72);
      if (c->d_text[k].n < cap * row_guess) HIP_TRY(c, c->d_text[k].resize(cap * row_guess));
    }
    if (replay) {
      c->flag_cap[k] = flag_cap_for(cap);
      HIP_TRY(c, c->d_flags[k].resize(flag_words(cap, c->flag_cap[k])));
      HIP_TRY(c, c->h_flags[k].resize(flag_head_words(c->flag_cap[k])));
      if (text) HIP_TRY(c, c->h_flag_rows[k].resize(kFlagRowsCap));
    }
  }
  // (run_direct: the device addresses of the pinned host buffers -- the same numbers under unified addressing, asked for anyway)
  ngsld_rec_std *dev_std[ngsld_ctx::kSlots] = {nullptr, nullptr, nullptr};
  ngsld_rec_ext *dev_ext[ngsld_ctx::kSlots] = {nullptr, nullptr, nullptr};
  for (int k = 0; k < S; ++k) {
    if (direct) {
      HIP_TRY(c, hipHostGetDevicePointer((void **)&dev_std[k], c->h_std[k].p, 0));
      if (ext) HIP_TRY(c, hipHostGetDevicePointer((void **)&dev_ext[k], c->h_ext[k].p, 0));
    } else {
      dev_std[k] = c->d_std[k].p;
      dev_ext[k] = ext ? c->d_ext[k].p : nullptr;
    }
  }
  size_t scan_bytes = 0;
  if (text) {
    if (c->text_stream == nullptr) HIP_TRY(c, hipStreamCreate(&c->text_stream));
    for (int k = 0; k < S; ++k)
      if (c->ev_scan_done[k] == nullptr) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_scan_done[k], hipEventDisableTiming));
    scan_bytes = text_scan_temp_bytes(cap);
    HIP_TRY(c, c->d_scan_tmp.resize(scan_bytes ? scan_bytes : 1));
    if (two_streams) HIP_TRY(c, c->d_scan_tmp_b.resize(scan_bytes ? scan_bytes : 1));
    if (replay) HIP_TRY(c, c->d_scan_tmp2.resize(scan_bytes ? scan_bytes : 1));
  }
  auto text_args = [&](const Batch &b, int k) -> TextArgs {
    TextArgs t{};
    t.items = c->d_items.p + c->h_item_off[b.r0];
    t.n_items = c->h_item_off[b.r1] - c->h_item_off[b.r0];
    t.out_base = c->h_row_off[b.r0];
    t.n_pairs = b.n;
    t.std_rec = c->d_std[k].p;
    t.ext_rec = ext ? c->d_ext[k].p : nullptr;
    t.maf = c->d_maf.p;
    t.cum = c->d_cum.p;
    t.infc = c->d_infc.p;
    t.labels = c->have_labels ? c->d_labels.p : nullptr;
    t.label_off = c->d_label_off.p;
    t.lens = c->d_lens[k].p;
    t.offs = c->d_offs[k].p;
    t.text = c->d_text[k].p;
    t.text_cap = 0;
    t.overflow = c->d_text_meta[k].p + 3;
    t.needs_host = reinterpret_cast<int *>(c->d_text_meta[k].p + 1);
    return t;
  };
  std::vector<Item> rel_items;
  std::vector<uint64_t> recs;
  std::vector<uint32_t> rep_s1, rep_s2;
  auto issue = [&](size_t bi) -> int {  // kernel on a compute stream; text: lengths behind it; records: D2H on `copy_stream`
    Range range_issue("ngsld:issue batch (pair kernel + D2H)");
    const int k = (int)(bi % (size_t)S);
    const Batch &b = batches[bi];
    hipStream_t st = (two_streams && (bi & 1)) ? c->stream2 : c->stream;
    if (replay) {
      const int rcf = reset_flags(c, c->d_flags[k], b.n, c->flag_cap[k], st);
      if (rcf != NGSLD_OK) return rcf;
    }
    PairArgs a = make_args(c, b.r0, b.r1, dev_std[k], dev_ext[k], replay ? c->d_flags[k].p : nullptr, c->flag_cap[k], b.n);
    HIP_TRY(c, timed_launch(c, a, st));
    c->slot_dev_applied[k] = false;
    if (replay) {  // (called genotypes: the flagged pairs settled on the device, before anything reads the records)
      int rcd = device_replay(c, c->d_flags[k].p, c->flag_cap[k], c->h_row_off[b.r0], b.n, dev_std[k], dev_ext[k], st, k);
      if (rcd != NGSLD_OK) return rcd;
      // (likelihoods: the same once the exact store is there -- a batch issued before that is settled when it is consumed)
      if (lkl_device_eligible(c) && (exact_store_started(c) || exact_store_is_free(c))) {
        rcd = try_device_replay_lkl(c, c->d_flags[k].p, c->flag_cap[k], c->h_row_off[b.r0], b.n, dev_std[k], dev_ext[k], st, true, k,
                                    &c->slot_dev_applied[k], exact_sites_needed(c, b.r0, b.r1));
        if (rcd != NGSLD_OK) return rcd;
      }
    }
    // which pairs the kernel flagged for the exact-order replay (counter + list, 32 KB): known to the host with the batch.
    // On the kernel's own stream, right behind it: on the copy stream, behind the records, this small copy took 9 ms per
    // batch -- it goes through a copy kernel, and that waited for the next batch's pair kernel to leave it a CU
    // (a small KERNEL, not a copy: send_flag_head)
    if (replay) {
      const int rch = send_flag_head(c, c->d_flags[k].p, c->h_flags[k].p, c->flag_cap[k], st, !c->slot_dev_applied[k]);
      if (rch != NGSLD_OK) return rch;
    }
    if (text) {  // row lengths and their prefix sums right behind the pair kernel, the rows behind those
      HIP_TRY(c, hipMemsetAsync(c->d_text_meta[k].p, 0, 4 * sizeof(uint64_t), st));
      TextArgs t = text_args(b, k);
      HIP_TRY(c, launch_text_lengths(t, st));
      HIP_TRY(c, text_scan(st == c->stream ? c->d_scan_tmp.p : c->d_scan_tmp_b.p, scan_bytes, c->d_lens[k].p, c->d_offs[k].p, b.n,
                           c->d_text_meta[k].p, st));
      // The rows themselves, at once: on a stream of their own that waits for this batch's prefix sums -- no trip through
      // the host.  Rounds 1-4 wrote the rows from the host's side of the loop (on the copy stream, once the host had read the
      // batch's length): with fewer hardware queues than busy streams (GPU_MAX_HW_QUEUES, or other streams in the process) that
      // kernel was SUBMITTED behind the pair kernels of the next two batches, queued behind them, and the host waited for
      // them before it issued more -- 0.55 s for configs[2]'s loop on four queues, 0.80 on two, 0.94 on one
      // (profiles/r04/hw_queues_ab.txt).  Submitted here it stands before them in whatever queue it shares; the copy stream
      // carries nothing but the D2H copies (SDMA).  A batch whose records are patched afterwards (host replay) or that
      // outgrows the buffer is written again when consumed.
      t.text_cap = c->d_text[k].n;
      if (replay) {  // (the pairs left to the host: which they are, where their rows lie -- see consume)
        const int rcw = send_flag_rows(c, c->d_flags[k].p, c->flag_cap[k], c->slot_dev_applied[k], c->d_offs[k].p, c->d_lens[k].p, b.n,
                                       c->h_row_off[b.r0], c->h_flag_rows[k].p, st);
        if (rcw != NGSLD_OK) return rcw;
      }
      HIP_TRY(c, hipEventRecord(c->ev_scan_done[k], st));
      HIP_TRY(c, hipStreamWaitEvent(c->text_stream, c->ev_scan_done[k], 0));
      HIP_TRY(c, launch_text_write(t, c->text_stream));
      HIP_TRY(c, hipMemcpyAsync(c->h_text_meta[k].p, c->d_text_meta[k].p, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->text_stream));
      HIP_TRY(c, hipEventRecord(c->ev_kernel_done[k], c->text_stream));
      return NGSLD_OK;
    }
    HIP_TRY(c, hipEventRecord(c->ev_kernel_done[k], st));
    if (direct) return NGSLD_OK;  // (the records are in host memory when the kernel is done)
    HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->ev_kernel_done[k], 0));
    if (b.n) {
      HIP_TRY(c, hipMemcpyAsync(c->h_std[k].p, c->d_std[k].p, b.n * sizeof(ngsld_rec_std), hipMemcpyDeviceToHost,
                                c->copy_stream));
      if (ext)
        HIP_TRY(c, hipMemcpyAsync(c->h_ext[k].p, c->d_ext[k].p, b.n * sizeof(ngsld_rec_ext), hipMemcpyDeviceToHost,
                                  c->copy_stream));
    }
    HIP_TRY(c, hipEventRecord(c->ev_copy_done[k], c->copy_stream));
    return NGSLD_OK;
  };
  const auto t_run = std::chrono::steady_clock::now();
  auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run).count(); };
  // ---- what consuming a text batch is made of (see the loop below) ----
  const bool host_patch_on = [] {
    const char *e = test_knob("TEXT_HOST_PATCH");  // (tests: 0 = the replayed rows through the device again, the fallback of a patch that does not fit)
    return e == nullptr || std::strcmp(e, "0") != 0;
  }();
  std::vector<ngsld_rec_std> hp_std;
  std::vector<ngsld_rec_ext> hp_ext;
  std::vector<uint32_t> hp_row;  // recs[j]'s entry of the slot's h_flag_rows
  std::vector<std::pair<uint64_t, uint32_t>> by_rec;
  auto batch_needs_host = [&](int k, size_t bi) -> bool {  // a value beyond the device formatter's fast path: the batch goes out as records
    bool needs_host = (c->h_text_meta[k].p[1] & 0xffffffffull) != 0;
    if (const char *e = test_knob("TEXT_FALLBACK_EVERY")) {  // tests: every n-th batch takes the record path
      const uint64_t every = std::strtoull(e, nullptr, 10);
      if (every > 0 && bi % every == every - 1) needs_host = true;
    }
    return needs_host;
  };
  // recs' entries among what send_flag_rows wrote for the slot (the same list, the same bound as the kernel's): rep_s1 / rep_s2 /
  // hp_row filled; false when the batch left more pairs than the kernel writes, or one is not there
  auto find_flag_rows = [&](int k, bool applied) -> bool {
    const uint32_t *hd = c->h_flags[k].p;
    const bool host_list = applied || hd[7] != 0;
    const uint32_t n = host_list ? hd[1] : hd[0];
    if (n > kFlagRowsCap || n > (host_list ? kFlagHostCap : c->flag_cap[k]) || recs.size() > n) return false;
    const FlagRow *rows = c->h_flag_rows[k].p;
    rep_s1.resize(recs.size());
    rep_s2.resize(recs.size());
    hp_row.resize(recs.size());
    by_rec.resize(n);  // (the list is in the order the atomics landed in)
    for (uint32_t q = 0; q < n; ++q) by_rec[q] = {rows[q].rec, q};
    std::sort(by_rec.begin(), by_rec.end());
    for (size_t j = 0; j < recs.size(); ++j) {
      const auto hit = std::lower_bound(by_rec.begin(), by_rec.end(), std::make_pair(recs[j], 0u));
      const uint32_t at = hit != by_rec.end() && hit->first == recs[j] ? hit->second : n;
      if (at == n || rows[at].s1 >= c->n_sites || rows[at].s2 >= c->n_sites) return false;
      hp_row[j] = at;
      rep_s1[j] = rows[at].s1;
      rep_s2[j] = rows[at].s2;
    }
    return true;
  };
  // the replayed records' value columns over the old ones in the text the host holds; false: a row has another length now (or
  // is not where it should be) -- the caller takes the device's way
  uint64_t patch_fail_every = 0;  // tests: every n-th patched batch pretends its last row changed length (the fallback's way out)
  if (const char *e = test_knob("TEXT_HOST_PATCH_FAIL_EVERY")) patch_fail_every = std::strtoull(e, nullptr, 10);
  uint64_t patched_batches = 0;
  auto apply_host_patch = [&](int k) -> bool {
    const uint64_t total = c->h_text_meta[k].p[0];
    const bool pretend = patch_fail_every != 0 && ++patched_batches % patch_fail_every == 0;
    char *text_p = c->h_text[k].p;
    const int tabs = ext ? 16 : 4;  // the columns behind `dist` (ngsLD.cpp:314-351): labels may hold tabs of their own, values never
    char buf[2048];
    for (size_t j = 0; j < recs.size(); ++j) {
      const FlagRow &r = c->h_flag_rows[k].p[hp_row[j]];
      if (r.len < 2 || r.off + r.len > total || text_p[r.off + r.len - 1] != '\n') return false;
      char *row = text_p + r.off;
      uint32_t at = r.len;
      for (int seen = 0; at > 0 && seen < tabs;)
        if (row[--at] == '\t') ++seen;
      if (row[at] != '\t') return false;
      const size_t m = ngsld_host_format_pair(buf, sizeof(buf), "", "", 0.0, &hp_std[j], ext ? &hp_ext[j] : nullptr, c->h_maf[rep_s1[j]],
                                              c->h_maf[rep_s2[j]]);
      if (m < 4 || m - 3 != (size_t)(r.len - at)) return false;  // ("\t\t0" stands for the labels and dist)
      if (pretend && j + 1 == recs.size()) return false;
      std::memcpy(row + at, buf + 3, m - 3);
    }
    return true;
  };
  auto device_patch = [&](int k, const Batch &b, bool &rewrite) -> int {
    // flagged pairs: replayed on the host, patched into the device records, and the row lengths derived again --
    // all on the copy stream, beside the next batch's pair kernel
    int rcr = replay_flagged(c, recs, c->h_row_off[b.r0], nullptr, nullptr, c->d_std[k].p, ext ? c->d_ext[k].p : nullptr,
                           c->copy_stream, &rep_s1, &rep_s2);
    if (rcr != NGSLD_OK) return rcr;
    // Only the replayed rows' lengths are derived again (replay_flagged left their record indices in d_patch_idx); the
    // prefix sums are taken again only if one of them changed -- a full length pass + scan beside the next batch's pair
    // kernel cost that kernel ~1 ms of every 11 (profiles/r04/e2e_timeline.txt)
    if (!recs.empty()) {
      rewrite = true;
      HIP_TRY(c, c->d_patch_s1.resize(recs.size()));
      HIP_TRY(c, c->d_patch_s2.resize(recs.size()));
      HIP_TRY(c, hipMemcpyAsync(c->d_patch_s1.p, rep_s1.data(), recs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->copy_stream));
      HIP_TRY(c, hipMemcpyAsync(c->d_patch_s2.p, rep_s2.data(), recs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->copy_stream));
      HIP_TRY(c, hipMemsetAsync(c->d_text_meta[k].p + 2, 0, sizeof(uint64_t), c->copy_stream));
      const TextArgs t = text_args(b, k);
      HIP_TRY(c, launch_text_relength(t, c->d_patch_idx.p, c->d_patch_s1.p, c->d_patch_s2.p, recs.size(), c->d_text_meta[k].p + 2,
                                      c->copy_stream));
      HIP_TRY(c, hipMemcpyAsync(c->h_text_meta[k].p, c->d_text_meta[k].p, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                c->copy_stream));
      HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
      if (c->h_text_meta[k].p[2] != 0) {
        HIP_TRY(c, text_scan(c->d_scan_tmp2.p, scan_bytes, c->d_lens[k].p, c->d_offs[k].p, b.n, c->d_text_meta[k].p, c->copy_stream));
        HIP_TRY(c, hipMemcpyAsync(c->h_text_meta[k].p, c->d_text_meta[k].p, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                  c->copy_stream));
        HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
      }
    }
    return NGSLD_OK;
  };
  auto finish_text = [&](int k, size_t bi, const Batch &b, bool &rewrite, bool &as_records, ngsld_batch &out, double &t_wr) -> int {
    const uint64_t total = c->h_text_meta[k].p[0];
    if (batch_needs_host(k, bi)) {
      as_records = true;  // a value beyond the device formatter's fast path: this batch goes out as records
      HIP_TRY(c, c->h_std[k].resize(cap));
      if (ext) HIP_TRY(c, c->h_ext[k].resize(cap));
      if (b.n) {
        HIP_TRY(c, hipMemcpyAsync(c->h_std[k].p, c->d_std[k].p, b.n * sizeof(ngsld_rec_std), hipMemcpyDeviceToHost,
                                  c->copy_stream));
        if (ext)
          HIP_TRY(c, hipMemcpyAsync(c->h_ext[k].p, c->d_ext[k].p, b.n * sizeof(ngsld_rec_ext), hipMemcpyDeviceToHost,
                                    c->copy_stream));
      }
      HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
      const int rc1 = need_host_items();
      if (rc1 != NGSLD_OK) return rc1;
    } else {
      if (total > c->d_text[k].n) {
        HIP_TRY(c, c->d_text[k].resize(total + total / 8));
        rewrite = true;
      }
      if (c->h_text_meta[k].p[3] != 0) rewrite = true;  // (the early write pass ran out of room)
      if (total > c->h_text[k].n) HIP_TRY(c, c->h_text[k].resize(total + total / 8));
      if (total) {
        if (rewrite) {
          const TextArgs t = text_args(b, k);
          HIP_TRY(c, launch_text_write(t, c->copy_stream));
        }
        HIP_TRY(c, hipMemcpyAsync(c->h_text[k].p, c->d_text[k].p, total, hipMemcpyDeviceToHost, c->copy_stream));
      }
      HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
      t_wr = now_ms();
      out.text = c->h_text[k].p;
      out.text_len = total;
    }
    return NGSLD_OK;
  };
  int rc = NGSLD_OK;
  const bool trace = std::getenv("NGSLD_TRACE") != nullptr;  // dev: per-batch host timeline on stderr
  // S - 1 batches are in flight while one is consumed: the slot of batch bi + S - 1 was last used by batch bi - 1, whose
  // sink call has returned
  for (size_t bi = 0; rc == NGSLD_OK && bi + 1 < (size_t)S && bi < batches.size(); ++bi) rc = issue(bi);
  for (size_t bi = 0; rc == NGSLD_OK && bi < batches.size(); ++bi) {
    const int k = (int)(bi % (size_t)S);
    const double t_a = now_ms();
    if (bi + (size_t)S - 1 < batches.size()) {
      rc = issue(bi + (size_t)S - 1);
      if (rc != NGSLD_OK) break;
    }
    const double t_b = now_ms();
    const Batch &b = batches[bi];
    const uint64_t i0 = c->h_item_off[b.r0], i1 = c->h_item_off[b.r1];
    ngsld_batch out{};
    out.s1_begin = b.r0;
    out.s1_end = b.r1;
    out.n_pairs = b.n;
    bool as_records = !text;
    Range range_wait(text ? "ngsld:consume batch (text rows, D2H, replay, sink)" : "ngsld:consume batch (wait for records, replay, sink)");
    double t_ev = 0.0, t_wr = 0.0;
    if (text) {
      // the batch's text: its length is known now; the rows are written and copied on the copy stream while the pair
      // kernel of the next batch (already enqueued) runs on the compute stream
      HIP_TRY(c, hipEventSynchronize(c->ev_kernel_done[k]));
      t_ev = now_ms();
      bool rewrite = false;  // the rows the issue wrote are stale: records were patched since
      bool host_patch = false;  // the host's pairs go into the text the host holds (below)
      if (replay) c->flagged_pairs += c->h_flags[k].p[0];
      if (replay && c->h_flags[k].p[0] != 0) {
        bool applied = c->slot_dev_applied[k];
        if (!applied && exact_store_wanted(c, c->h_flags[k].p[0] - c->h_flags[k].p[1])) {
          // a likelihood matrix that flags more pairs than the host should replay, and this batch went out before the exact
          // store was there: build it (once), replay the batch's pairs on the device, take every row's length again
          int rcx = try_device_replay_lkl(c, c->d_flags[k].p, c->flag_cap[k], c->h_row_off[b.r0], b.n, c->d_std[k].p,
                                          ext ? c->d_ext[k].p : nullptr, c->copy_stream, true, k, &applied, exact_sites_needed(c, b.r0, b.r1));
          if (rcx != NGSLD_OK) return rcx;
          if (applied) {
            rcx = send_flag_head(c, c->d_flags[k].p, c->h_flags[k].p, c->flag_cap[k], c->copy_stream, false);
            if (rcx != NGSLD_OK) return rcx;
            HIP_TRY(c, hipMemsetAsync(c->d_text_meta[k].p, 0, 4 * sizeof(uint64_t), c->copy_stream));
            const TextArgs t = text_args(b, k);
            HIP_TRY(c, launch_text_lengths(t, c->copy_stream));
            HIP_TRY(c, text_scan(c->d_scan_tmp2.p, scan_bytes, c->d_lens[k].p, c->d_offs[k].p, b.n, c->d_text_meta[k].p, c->copy_stream));
            HIP_TRY(c, hipMemcpyAsync(c->h_text_meta[k].p, c->d_text_meta[k].p, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->copy_stream));
            HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
            rewrite = true;
          }
        }
        int rcr = flagged_records(c, c->h_flags[k].p, c->d_flags[k].p, c->flag_cap[k], b.n, recs, applied);
        if (rcr != NGSLD_OK) return rcr;
        // The pairs left to the host (the headline's few dozen, a batch's host-only pairs behind a device-side replay): replayed
        // on the host's threads and written over their rows' value columns in the text the host receives -- send_flag_rows told
        // which pairs they are and where their rows lie, so NOTHING is submitted to the device here.  (Rounds 1-5 located them,
        // patched the device's records, took the rows' lengths again and wrote the batch's text again -- kernels that, in a
        // hardware queue shared with the compute streams, stood behind the pair kernels of the next two batches.)
        host_patch = host_patch_on && !recs.empty() && !rewrite && applied == c->slot_dev_applied[k] && !batch_needs_host(k, bi) &&
                     c->h_text_meta[k].p[3] == 0 && c->h_text_meta[k].p[0] <= c->d_text[k].n && find_flag_rows(k, applied);
        if (host_patch) {
          hp_std.resize(recs.size());
          hp_ext.resize(ext ? recs.size() : 0);
          rcr = replay_pairs_on_host(c, rep_s1.data(), rep_s2.data(), recs.size(), hp_std.data(), ext ? hp_ext.data() : nullptr);
        } else {
          rcr = device_patch(k, b, rewrite);
        }
        if (rcr != NGSLD_OK) return rcr;
      }
      int rcf = finish_text(k, bi, b, rewrite, as_records, out, t_wr);
      if (rcf != NGSLD_OK) return rcf;
      if (host_patch) {
        if (!as_records && apply_host_patch(k)) {
          c->replayed_pairs += recs.size();
          c->host_replayed_total += recs.size();
          c->text_rows_patched += recs.size();
        } else {  // (a replayed row of another length -- once in a few thousand replays: the device's way, the text copied again)
          rcf = device_patch(k, b, rewrite);
          if (rcf == NGSLD_OK) rcf = finish_text(k, bi, b, rewrite, as_records, out, t_wr);
          if (rcf != NGSLD_OK) return rcf;
        }
      }
    } else {
      HIP_TRY(c, hipEventSynchronize(direct ? c->ev_kernel_done[k] : c->ev_copy_done[k]));
      if (trace) {
        const uint32_t *hd = c->h_flags[k].p;
        std::fprintf(stderr, "[trace] batch %zu (%llu pairs): issue next %.2f..%.2f, records on the host %.2f, flag head %u %u %u %u %u %u %u %u\n", bi,
                     (unsigned long long)b.n, t_a, t_b, now_ms(), replay ? hd[0] : 0u, replay ? hd[1] : 0u, replay ? hd[2] : 0u, replay ? hd[3] : 0u,
                     replay ? hd[4] : 0u, replay ? hd[5] : 0u, replay ? hd[6] : 0u, replay ? hd[7] : 0u);
      }
      if (replay) c->flagged_pairs += c->h_flags[k].p[0];
      if (replay && c->h_flags[k].p[0] != 0) {  // flagged pairs: replayed on the host, patched into the batch's buffers
        bool applied = c->slot_dev_applied[k];
        if (!applied && exact_store_wanted(c, c->h_flags[k].p[0] - c->h_flags[k].p[1])) {  // (see the text branch above)
          int rcx = try_device_replay_lkl(c, c->d_flags[k].p, c->flag_cap[k], c->h_row_off[b.r0], b.n, dev_std[k], dev_ext[k],
                                          c->copy_stream, true, k, &applied, exact_sites_needed(c, b.r0, b.r1));
          if (rcx != NGSLD_OK) return rcx;
          if (applied) {
            rcx = send_flag_head(c, c->d_flags[k].p, c->h_flags[k].p, c->flag_cap[k], c->copy_stream, false);
            if (rcx != NGSLD_OK) return rcx;
            if (!direct && b.n) {  // (the records had been copied already: once more)
              HIP_TRY(c, hipMemcpyAsync(c->h_std[k].p, c->d_std[k].p, b.n * sizeof(ngsld_rec_std), hipMemcpyDeviceToHost, c->copy_stream));
              if (ext)
                HIP_TRY(c, hipMemcpyAsync(c->h_ext[k].p, c->d_ext[k].p, b.n * sizeof(ngsld_rec_ext), hipMemcpyDeviceToHost, c->copy_stream));
            }
            HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
          }
        }
        int rcr = flagged_records(c, c->h_flags[k].p, c->d_flags[k].p, c->flag_cap[k], b.n, recs, applied);
        if (rcr == NGSLD_OK)
          rcr = replay_flagged(c, recs, c->h_row_off[b.r0], c->h_std[k].p, ext ? c->h_ext[k].p : nullptr, nullptr, nullptr, nullptr);
        if (rcr != NGSLD_OK) return rcr;
      }
    }
    if (as_records) {
      rel_items.assign(c->h_items.begin() + (ptrdiff_t)i0, c->h_items.begin() + (ptrdiff_t)i1);
      for (auto &it : rel_items) it.first_record -= c->h_row_off[b.r0];
      out.n_items = i1 - i0;
      out.items = rel_items.data();
      out.std = c->h_std[k].p;
      out.ext = ext ? c->h_ext[k].p : nullptr;
    }
    if (trace)
      std::fprintf(stderr, "[trace] batch %zu: issue next %.2f..%.2f, its kernels done %.2f, text written + on the host %.2f, replay done %.2f\n",
                   bi, t_a, t_b, t_ev, t_wr, now_ms());
    Range range_sink("ngsld:sink");
    if (sink(user, &out) != 0) rc = fail(c, NGSLD_ERR_SINK, "sink callback failed");
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream2));
  if (c->text_stream) HIP_TRY(c, hipStreamSynchronize(c->text_stream));
  HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
  if (rc != NGSLD_OK) return rc;
  if (c->exact_state.load() == 1) {  // (the store's builder is still at the sites behind this run's rows: its errors are this run's)
    bool have = false;
    const int rcs = wait_exact_store(c, c->n_sites, &have);
    if (rcs != NGSLD_OK) return rcs;
  }
  return check_status(c);
}

// ---------------------------------------------------------------------------------------------------------------
// Text output of a likelihood matrix that is NOT SNP-called (degenerate sites were found at ngsld_set_geno_*): a third of the
// pairs will be replayed, and the replay that does that at speed -- a lane per pair -- wants launches of millions of records,
// not a text batch's 2^19 (a wavefront per pair there: 5.4e7 replayed pairs/s against 1.3e8; configs[2]'s un-called twin
// through the binary 1.57 s against 0.90 SNP-called).  So the rows go in GROUPS of up to 2^25 pairs:
//   A(g)  one launch of pair kernels + one device-side replay into records in device memory, as ngsld_run_device +
//         ngsld_finish_device make them (the few pairs only the host settles patched in), on stream2;
//   T(g)  the text of the whole group from its final records, on the device: lengths and rows batch by batch (a batch that meets
//         a value beyond the formatter's range goes out as records, as in run_range), ONE prefix sum over the group;
//   C(g)  the group's text to the host and the sink, batch by batch through the pinned buffers (SDMA copies: they do not care what
//         the compute units do).
// Order on the host: finish A(g) -> T(g) -> launch A(g+1) -> C(g).  T(g) goes BEFORE A(g+1) is launched: once a pair kernel's
// grid is being dispatched no other queue's kernel gets a workgroup in, whatever its priority (measured: the first text kernels
// of a group issued beside the next group's pair kernel waited 80-90 ms, the kernel's whole length); what is exposed instead is
// T(g)'s own ~30 ms a group.
// ---------------------------------------------------------------------------------------------------------------
struct ReadyRecords {  // records that are final in device memory: record 0 = plan record `base`
  const ngsld_rec_std *std;
  const ngsld_rec_ext *ext;
  uint64_t base;
};

__global__ void group_bounds_kernel(const uint64_t *offs, const uint64_t *first_rec, uint32_t n_batches, uint64_t n_group, const uint64_t *total,
                                    const int *needs, uint64_t *h_bounds, int *h_needs) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_batches) {
    // (a batch of rows without a pair behind the group's last record -- a group of nothing but such rows has one -- starts where the text ends:
    // offs[] has n_group entries)
    h_bounds[j] = first_rec[j] < n_group ? offs[first_rec[j]] : *total;
    h_needs[j] = needs[j];
  } else if (j == n_batches) {
    h_bounds[j] = *total;
  }
}

struct TextBatch {
  uint64_t r0, r1, n, rec0;  // rows, pairs, first record relative to the group
};

// T(g): returns with every kernel of the group's text enqueued on c->stream and ev_kernel_done[0] recorded behind them
static int group_text_kernels(ngsld_ctx *c, const ReadyRecords &rr, const std::vector<TextBatch> &tb, uint64_t n_group) {
  const bool ext = c->params.extend_out != 0;
  const uint32_t nb = (uint32_t)tb.size();
  hipStream_t st = c->stream;
  HIP_TRY(c, c->d_group_lens.resize(n_group ? n_group : 1));
  HIP_TRY(c, c->d_group_offs.resize(n_group ? n_group : 1));
  HIP_TRY(c, c->d_group_meta.resize(4));
  HIP_TRY(c, c->d_group_needs.resize(nb + 1));
  HIP_TRY(c, c->d_group_first.resize(nb + 1));
  HIP_TRY(c, c->h_group_bounds.resize(nb + 1));
  HIP_TRY(c, c->h_group_needs.resize(nb + 1));
  const size_t scan_bytes = text_scan_temp_bytes(n_group);
  HIP_TRY(c, c->d_scan_tmp.resize(scan_bytes ? scan_bytes : 1));
  std::vector<uint64_t> first(nb + 1, 0);
  for (uint32_t j = 0; j < nb; ++j) first[j] = tb[j].rec0;
  HIP_TRY(c, hipMemcpyAsync(c->d_group_first.p, first.data(), (nb + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemsetAsync(c->d_group_needs.p, 0, (nb + 1) * sizeof(int), st));
  HIP_TRY(c, hipMemsetAsync(c->d_group_meta.p, 0, 4 * sizeof(uint64_t), st));
  auto args = [&](const TextBatch &b, uint32_t j) -> TextArgs {
    TextArgs t{};
    t.items = c->d_items.p + c->h_item_off[b.r0];
    t.n_items = c->h_item_off[b.r1] - c->h_item_off[b.r0];
    t.out_base = c->h_row_off[b.r0];
    t.n_pairs = b.n;
    t.std_rec = rr.std + b.rec0;
    t.ext_rec = ext ? rr.ext + b.rec0 : nullptr;
    t.maf = c->d_maf.p;
    t.cum = c->d_cum.p;
    t.infc = c->d_infc.p;
    t.labels = c->have_labels ? c->d_labels.p : nullptr;
    t.label_off = c->d_label_off.p;
    t.lens = c->d_group_lens.p + b.rec0;
    t.offs = c->d_group_offs.p + b.rec0;   // (prefix sums over the GROUP: offsets into the group's text)
    t.text = c->d_group_text.p;
    t.text_cap = 0;
    t.overflow = c->d_group_meta.p + 3;
    t.needs_host = c->d_group_needs.p + j;
    return t;
  };
  for (uint32_t j = 0; j < nb; ++j) HIP_TRY(c, launch_text_lengths(args(tb[j], j), st));
  HIP_TRY(c, text_scan(c->d_scan_tmp.p, scan_bytes, c->d_group_lens.p, c->d_group_offs.p, n_group, c->d_group_meta.p, st));
  // the group's text length: the one thing the host has to know before the rows are written (their buffer)
  uint64_t total = 0;
  HIP_TRY(c, hipMemcpyAsync(&total, c->d_group_meta.p, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipStreamSynchronize(st));
  if (total > c->d_group_text.n) HIP_TRY(c, c->d_group_text.resize(total + total / 16));
  for (uint32_t j = 0; j < nb; ++j) {
    TextArgs t = args(tb[j], j);
    t.text = c->d_group_text.p;
    HIP_TRY(c, launch_text_write(t, st));
  }
  uint64_t *hb = nullptr;
  int *hn = nullptr;
  HIP_TRY(c, hipHostGetDevicePointer((void **)&hb, c->h_group_bounds.p, 0));
  HIP_TRY(c, hipHostGetDevicePointer((void **)&hn, c->h_group_needs.p, 0));
  hipLaunchKernelGGL(group_bounds_kernel, dim3((nb + 1 + 255) / 256), dim3(256), 0, st, c->d_group_offs.p, c->d_group_first.p, nb, n_group, c->d_group_meta.p,
                     c->d_group_needs.p, hb, hn);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipEventRecord(c->ev_kernel_done[0], st));
  return NGSLD_OK;
}

// C(g): the group's text (or, for a batch the formatter gave up on, its records) to the host and the sink, batch by batch
static int group_text_to_sink(ngsld_ctx *c, const ReadyRecords &rr, const std::vector<TextBatch> &tb, ngsld_sink_fn sink, void *user) {
  const bool ext = c->params.extend_out != 0;
  const size_t nb = tb.size();
  HIP_TRY(c, hipEventSynchronize(c->ev_kernel_done[0]));  // T(g) is through: bounds and fallbacks are in host memory
  const uint64_t *bounds = c->h_group_bounds.p;
  const int *needs = c->h_group_needs.p;
  constexpr int S = ngsld_ctx::kSlots;
  std::vector<Item> rel_items;
  auto fallback = [&](size_t j) {
    bool f = needs[j] != 0;
    if (const char *e = test_knob("TEXT_FALLBACK_EVERY")) {  // tests: every n-th batch takes the record path
      const uint64_t every = std::strtoull(e, nullptr, 10);
      if (every > 0 && j % every == every - 1) f = true;
    }
    return f;
  };
  auto issue_copy = [&](size_t j) -> int {
    const int k = (int)(j % (size_t)S);
    const TextBatch &b = tb[j];
    if (fallback(j)) {
      HIP_TRY(c, c->h_std[k].resize(b.n ? b.n : 1));
      if (ext) HIP_TRY(c, c->h_ext[k].resize(b.n ? b.n : 1));
      if (b.n) {
        HIP_TRY(c, hipMemcpyAsync(c->h_std[k].p, rr.std + b.rec0, b.n * sizeof(ngsld_rec_std), hipMemcpyDeviceToHost, c->copy_stream));
        if (ext) HIP_TRY(c, hipMemcpyAsync(c->h_ext[k].p, rr.ext + b.rec0, b.n * sizeof(ngsld_rec_ext), hipMemcpyDeviceToHost, c->copy_stream));
      }
    } else {
      const uint64_t len = bounds[j + 1] - bounds[j];
      if (len > c->h_text[k].n) HIP_TRY(c, c->h_text[k].resize(len + len / 8));
      if (len) HIP_TRY(c, hipMemcpyAsync(c->h_text[k].p, c->d_group_text.p + bounds[j], len, hipMemcpyDeviceToHost, c->copy_stream));
    }
    HIP_TRY(c, hipEventRecord(c->ev_copy_done[k], c->copy_stream));
    return NGSLD_OK;
  };
  int rc = NGSLD_OK;
  for (size_t j = 0; rc == NGSLD_OK && j + 1 < (size_t)S && j < nb; ++j) rc = issue_copy(j);
  for (size_t j = 0; rc == NGSLD_OK && j < nb; ++j) {
    const int k = (int)(j % (size_t)S);
    if (j + (size_t)S - 1 < nb) {
      rc = issue_copy(j + (size_t)S - 1);
      if (rc != NGSLD_OK) break;
    }
    const TextBatch &b = tb[j];
    Range range_wait("ngsld:consume batch (text of a group, D2H, sink)");
    HIP_TRY(c, hipEventSynchronize(c->ev_copy_done[k]));
    ngsld_batch out{};
    out.s1_begin = b.r0;
    out.s1_end = b.r1;
    out.n_pairs = b.n;
    if (fallback(j)) {
      const int rci = ensure_host_items(c);
      if (rci != NGSLD_OK) return rci;
      const uint64_t i0 = c->h_item_off[b.r0], i1 = c->h_item_off[b.r1];
      rel_items.assign(c->h_items.begin() + (ptrdiff_t)i0, c->h_items.begin() + (ptrdiff_t)i1);
      for (auto &it : rel_items) it.first_record -= c->h_row_off[b.r0];
      out.n_items = i1 - i0;
      out.items = rel_items.data();
      out.std = c->h_std[k].p;
      out.ext = ext ? c->h_ext[k].p : nullptr;
    } else {
      out.text = c->h_text[k].p;
      out.text_len = bounds[j + 1] - bounds[j];
    }
    Range range_sink("ngsld:sink");
    if (sink(user, &out) != 0) rc = fail(c, NGSLD_ERR_SINK, "sink callback failed");
  }
  HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
  return rc;
}

static int run_grouped(ngsld_ctx *c, uint64_t s1_begin, uint64_t s1_end, ngsld_sink_fn sink, void *user, uint64_t group_pairs) {
  const bool ext = c->params.extend_out != 0;
  struct Group {
    uint64_t r0, r1, n;
  };
  std::vector<Group> groups;
  uint64_t cap = 1;
  for (uint64_t r0 = s1_begin; r0 < s1_end;) {
    // (the first group a quarter of the size: nothing overlaps its kernels, the sink waits for them)
    const uint64_t target = groups.empty() ? std::max<uint64_t>(group_pairs / 4, std::min<uint64_t>(group_pairs, 1ull << 22)) : group_pairs;
    uint64_t r1 = r0 + 1;
    while (r1 < s1_end && c->h_row_off[r1 + 1] - c->h_row_off[r0] <= target) ++r1;
    groups.push_back({r0, r1, c->h_row_off[r1] - c->h_row_off[r0]});
    cap = std::max(cap, groups.back().n);
    r0 = r1;
  }
  // (... and the last one small too: nothing overlaps the trip of its text to the host)
  if (groups.size() >= 2 && groups.back().n > group_pairs / 3) {
    Group last = groups.back();
    uint64_t cut = last.r1;
    while (cut > last.r0 + 1 && c->h_row_off[last.r1] - c->h_row_off[cut - 1] <= group_pairs / 4) --cut;
    if (cut > last.r0 && cut < last.r1) {
      groups.back() = {last.r0, cut, c->h_row_off[cut] - c->h_row_off[last.r0]};
      groups.push_back({cut, last.r1, c->h_row_off[last.r1] - c->h_row_off[cut]});
    }
  }
  for (int k = 0; k < 2; ++k) {
    HIP_TRY(c, c->d_group_std[k].resize(cap));
    if (ext) HIP_TRY(c, c->d_group_ext[k].resize(cap));
  }
  HIP_TRY(c, c->d_group_lens.resize(cap));
  HIP_TRY(c, c->d_group_offs.resize(cap));
  {
    // (the largest group's text too: ~170 bytes a row with the extended columns, labels on top -- grown if a group needs more)
    const uint64_t row_guess = 2 * c->max_label + (ext ? 170 : 60);
    HIP_TRY(c, c->d_group_text.resize(cap * row_guess));
    const size_t scan_bytes = text_scan_temp_bytes(cap);
    HIP_TRY(c, c->d_scan_tmp.resize(scan_bytes ? scan_bytes : 1));
    const int rcr = reserve_device_run(c, cap);
    if (rcr != NGSLD_OK) return rcr;
  }
  if (c->reserve_thread.joinable()) c->reserve_thread.join();  // (ngsld_reserve_text_buffers: h_text[] is this thread's again)
  struct FlagText {  // (ngsld_run_device's launches flag what text needs flagged while this run lasts)
    ngsld_ctx *c;
    ~FlagText() { c->dev_run_flag_text = false; }
  } flag_text{c};
  c->dev_run_flag_text = true;
  const uint64_t text_batch = std::min<uint64_t>(c->batch_pairs, kTextBatchPairs);
  uint64_t flagged = 0, replayed = 0, on_device = 0;
  auto compute = [&](size_t g) -> int {  // A(g), enqueued on stream2 (no host wait)
    const int k = (int)(g & 1);
    return ngsld_run_device(c, groups[g].r0, groups[g].r1, c->d_group_std[k].p, ext ? c->d_group_ext[k].p : nullptr, c->stream2);
  };
  const bool trace = std::getenv("NGSLD_TRACE") != nullptr;  // dev: the groups' host timeline on stderr
  const auto t_run = std::chrono::steady_clock::now();
  auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run).count(); };
  int rc = groups.empty() ? NGSLD_OK : compute(0);
  std::vector<TextBatch> tb;
  for (size_t g = 0; rc == NGSLD_OK && g < groups.size(); ++g) {
    const double t_0 = now_ms();
    rc = finish_device_run(c);  // waits for A(g); the host's few pairs replayed and patched into the records
    if (rc == NGSLD_OK) rc = check_status(c);
    if (rc != NGSLD_OK) break;
    flagged += c->flagged_pairs;
    replayed += c->replayed_pairs;
    on_device += c->replayed_on_device;
    const int k = (int)(g & 1);
    const ReadyRecords ready{c->d_group_std[k].p, ext ? c->d_group_ext[k].p : nullptr, c->h_row_off[groups[g].r0]};
    tb.clear();
    for (uint64_t r0 = groups[g].r0; r0 < groups[g].r1;) {
      uint64_t r1 = r0 + 1;
      while (r1 < groups[g].r1 && c->h_row_off[r1 + 1] - c->h_row_off[r0] <= text_batch) ++r1;
      tb.push_back({r0, r1, c->h_row_off[r1] - c->h_row_off[r0], c->h_row_off[r0] - ready.base});
      r0 = r1;
    }
    const double t_1 = now_ms();
    rc = group_text_kernels(c, ready, tb, groups[g].n);                 // T(g)
    const double t_2 = now_ms();
    if (rc == NGSLD_OK && g + 1 < groups.size()) rc = compute(g + 1);   // A(g+1)
    const double t_3 = now_ms();
    if (rc == NGSLD_OK) rc = group_text_to_sink(c, ready, tb, sink, user);  // C(g)
    if (trace)
      std::fprintf(stderr, "[trace] group %zu (%llu pairs, %zu batches): its records final %.1f..%.1f, text kernels enqueued (length known) %.1f, "
                           "next group launched %.1f, text on the host and through the sink %.1f\n", g, (unsigned long long)groups[g].n, tb.size(), t_0, t_1, t_2, t_3, now_ms());
  }
  if (rc != NGSLD_OK) {  // (nothing of this run stays pending behind an error)
    (void)hipStreamSynchronize(c->stream2);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->copy_stream);
    c->dev_run.pending = false;
    return rc;
  }
  c->flagged_pairs = flagged;
  c->replayed_pairs = replayed;
  c->replayed_on_device = on_device;
  c->timed_pairs = c->h_row_off[s1_end] - c->h_row_off[s1_begin];
  if (c->exact_state.load() == 1) {
    bool have = false;
    const int rcs = wait_exact_store(c, c->n_sites, &have);
    if (rcs != NGSLD_OK) return rcs;
  }
  return check_status(c);
}
}  // namespace eng
}  // namespace ngsld

extern "C" {

int ngsld_run(ngsld_ctx *c, uint64_t s1_begin, uint64_t s1_end, ngsld_sink_fn sink, void *user) try {
  if (c == nullptr || sink == nullptr) return NGSLD_ERR_INVALID;
  if (!c->planned) return fail(c, NGSLD_ERR_INVALID, "ngsld_plan has not been called");
  if (s1_begin > s1_end || s1_end > c->n_sites) return fail(c, NGSLD_ERR_INVALID, "row range out of bounds");
  HIP_TRY(c, hipSetDevice(c->device));
  (void)hipGetLastError();  // (a failure some earlier call already reported must not surface as a launch's "last error")
  Range range_("ngsld:run");
  {
    const int rcp = finish_device_run(c);  // (see ngsld_run_device)
    if (rcp != NGSLD_OK) return rcp;
  }
  StoreGuard store_guard{c};
  // Groups (run_grouped) where the matrix is known to be un-called -- one site in 64 or more is degenerate --, the device can
  // replay its pairs, the rows become text on the device and there are at least four text batches' worth of them; the groups'
  // records, lengths, offsets and text (~360 bytes a pair, two groups of records) must have room: halved down to 2^22 pairs a
  // group before the run goes batch by batch.
  const uint64_t n_pairs = c->h_row_off[s1_end] - c->h_row_off[s1_begin];
  bool grouped = c->text_mode && c->replay_on && c->h_skip_count * 64ull >= c->n_sites && c->h_skip_count > 0 && lkl_device_eligible(c) &&
                 !c->exact_failed && (n_pairs >= (1ull << 21) || test_knob("TEXT_GROUP_PAIRS") != nullptr) && !test_knob_is("TEXT_GROUPS", "0");
  if (grouped) {
    // The store first -- the matrix is known to be un-called, so it is wanted, and its builder can work beside the first group's
    // kernels -- and then the groups' buffers out of what is left: under a memory cap (ngsld_set_memory_budget) the store must not
    // lose its room to them.  Without the lanes' individual-major copy a group has nothing over a batch: batch by batch then.
    if (!exact_store_started(c) && !c->exact_failed) {
      const int rcs = start_exact_store(c);
      if (rcs != NGSLD_OK) return rcs;
    }
    grouped = !c->exact_failed && c->xT_ready.load();
  }
  if (grouped) {
    uint64_t group_pairs = 1ull << 25;
    if (const char *e = test_knob("TEXT_GROUP_PAIRS")) group_pairs = std::max<uint64_t>(1024, std::strtoull(e, nullptr, 10));  // tests: many small groups
    const uint64_t per_pair = 2 * (sizeof(ngsld_rec_std) + (c->params.extend_out ? sizeof(ngsld_rec_ext) : 0)) + 16 + 2 * c->max_label + 200;
    while (group_pairs > (1ull << 22) && !room_for(std::min(group_pairs, n_pairs) * per_pair, 8ull << 30, 1ull << 30)) group_pairs /= 2;
    grouped = test_knob("TEXT_GROUP_PAIRS") != nullptr || room_for(std::min(group_pairs, n_pairs) * per_pair, 4ull << 30, 512ull << 20);
    if (grouped) {  // (text on the device needs exact prefix sums of the gaps -- made here, once, for all the groups; where they cannot
                    // be had run_range sends records to the host formatter: that path stays whole)
      const int rct = upload_text_prefix(c, &grouped);
      if (rct != NGSLD_OK) return rct;
    }
    if (grouped) return run_grouped(c, s1_begin, s1_end, sink, user, group_pairs);
  }
  return run_range(c, s1_begin, s1_end, sink, user);
} NGSLD_CATCH(c)

int ngsld_last_kernel_time(ngsld_ctx *c, double *total_ms, uint64_t *n_launches, uint64_t *n_pairs) {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  (void)hipGetLastError();  // (a failure some earlier call already reported must not surface as a launch's "last error")
  if (c->timed_stream) HIP_TRY(c, hipStreamSynchronize(c->timed_stream));
  if (c->timed_overlap) HIP_TRY(c, hipStreamSynchronize(c->stream2));
  double ms = 0.0;
  for (size_t k = 0; k < c->ev_used; ++k) {
    float t = 0.f;
    // launches that shared the device (two streams): the span from the first start to the last end, not the sum
    HIP_TRY(c, hipEventElapsedTime(&t, c->timed_overlap ? c->ev_pool[0].first : c->ev_pool[k].first, c->ev_pool[k].second));
    ms = c->timed_overlap ? std::max(ms, (double)t) : ms + (double)t;
  }
  if (total_ms) *total_ms = ms;
  if (n_launches) *n_launches = c->ev_used;
  if (n_pairs) *n_pairs = c->timed_pairs;
  return NGSLD_OK;
}

}  // extern "C"
