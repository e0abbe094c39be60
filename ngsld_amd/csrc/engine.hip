// engine.hip -- host side of the C-ABI in include/ngsld.h: contexts, the genotype matrix (upload + per-site prep on the
// device), positions, tuning and the small queries.  The plan, the batch pipeline and the exact-order replay live in
// engine_plan.hip, engine_run.hip and engine_replay.hip; engine.h is what they share.
#include "engine.h"

#include <atomic>

namespace ngsld {
namespace eng {

thread_local std::string g_create_error;

int set_geno_common(ngsld_ctx *c, const double *gl, const double *maf, uint64_t n_sites, uint64_t n_ind,
                    const ngsld_geno_opts &o, bool normalised) {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  const int log_scale = o.log_scale, ignore_miss = o.ignore_miss_data, on_device = o.on_device;
  Range range_("ngsld:set_geno (upload + per-site prep)");
  if (o.call_geno && o.N_thresh > o.call_thresh)  // gen_func.cpp:887-888
    return fail(c, NGSLD_ERR_INVALID, "missing data threshold must be smaller than calling genotype threshold!");
  if (gl == nullptr || n_sites == 0 || n_ind == 0) return fail(c, NGSLD_ERR_INVALID, "empty genotype matrix");
  if (normalised && maf == nullptr) return fail(c, NGSLD_ERR_INVALID, "maf missing");
  if (n_sites >= 0xffffffffull) return fail(c, NGSLD_ERR_UNSUPPORTED, "n_sites must be below 2^32 - 1");
  PairConfig cfg;
  if (!pair_config(n_ind, &cfg, c->kernel_choice, ignore_miss != 0))
    return fail(c, NGSLD_ERR_UNSUPPORTED, "n_ind is outside the supported range");
  HIP_TRY(c, hipSetDevice(c->device));
  (void)hipGetLastError();  // (a failure some earlier call already reported must not surface as a launch's "last error")
  {  // a run left on a caller's stream is finished under the matrix and plan it was started with
    const int rcp = finish_device_run(c);
    if (rcp != NGSLD_OK) return rcp;
  }
  stop_exact_store(c);  // (the builder of the last matrix' exact store, if it is still at it)
  c->have_geno = false;
  c->planned = false;
  c->text_mode = false;  // labels belong to a matrix
  c->replay_read = nullptr;  // and so does the replay source
  c->replay_user = nullptr;
  c->replay_matrix = nullptr;
  c->exact_ready = false;    // ... and the exact store of the device-side replay
  c->exact_alias = false;
  c->exact_failed = false;
  c->exact_build_s = 0.0;
  c->host_replayed_total = 0;
  c->d_xplanes.release();
  c->d_xmaf.release();
  c->d_xT.release();
  c->d_xdepth.release();
  c->d_xperm.release();
  c->xT_ready = false;
  // (the buffers of a grouped text run -- records, lengths, text of up to 2^25 pairs -- belong to the matrix they were sized for)
  for (int k = 0; k < 2; ++k) {
    c->d_group_std[k].release();
    c->d_group_ext[k].release();
  }
  c->d_group_lens.release();
  c->d_group_offs.release();
  c->d_group_text.release();
  c->gopts = o;
  c->normalised = normalised;
  c->n_sites = n_sites;
  c->n_ind = n_ind;
  c->cfg = cfg;
  c->np = cfg.np;
  const bool trace = std::getenv("NGSLD_TRACE") != nullptr;  // dev: where this call's time goes, on stderr
  const auto t_0 = std::chrono::steady_clock::now();
  auto ms_since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count(); };
  const size_t plane_elems = (size_t)n_sites * 3 * c->np;
  HIP_TRY(c, c->d_planes.resize(plane_elems));
  HIP_TRY(c, c->d_maf.resize(n_sites));
  HIP_TRY(c, c->d_mean.resize(n_sites));
  HIP_TRY(c, c->d_rsx.resize(n_sites));
  HIP_TRY(c, c->d_sc4.resize(4 * n_sites));
  HIP_TRY(c, c->d_status.resize(1));
  HIP_TRY(c, hipMemsetAsync(c->d_status.p, 0, sizeof(int), c->stream));

  // Host matrices are ingested in chunks of sites through two staging buffers: the H2D copy of chunk k+1 overlaps
  // the prep kernel of chunk k, and the device never holds more than planes + 2 chunks (a 48 GB matrix does not need
  // a second 48 GB on the device).  Device-resident input is prepped in place in one launch.  Chunks of 64 MB: a pageable
  // (huge-page) host matrix crosses at 50 GB/s whatever the chunk (tools/probe_h2d.hip: one hipMemcpy 50.6 GB/s, from pinned
  // memory 55.6), and what this call costs on configs[2] -- 90..117 ms for 1.2 GB -- is hipMalloc of the planes (27..74 ms)
  // as much as the 25..35 ms of copies; 256 MB chunks measured 10 ms behind 64 / 32 MB (profiles/r04/e2e_stage_ab.txt).
  PrepArgs a{};
  a.planes = c->d_planes.p;
  a.site_stride = 3ull * c->np;
  a.np = c->np;
  a.n_ind = (uint32_t)n_ind;
  a.log_scale = log_scale;
  a.ignore_miss = ignore_miss;
  a.normalised_input = normalised ? 1 : 0;
  a.text_semantics = o.text_semantics;
  a.call_geno = o.call_geno;
  {
    const char *pe = test_knob("PREP_EXACT");  // tests: every triple through the reference's log / exp chain
    a.exact_chain = (pe != nullptr && std::strcmp(pe, "0") != 0) ? 1 : 0;
  }
  a.N_thresh = o.N_thresh;
  a.call_thresh = o.call_thresh;
  a.maf = c->d_maf.p;
  a.mean_e = c->d_mean.p;
  a.rsx = c->d_rsx.p;
  a.status = c->d_status.p;
  // text genotypes: are the individuals without data the reader's own triple?  (then the device-side replay of called
  // genotypes takes the pairs of their sites too, ld_replay.hip)
  c->missing_canonical = false;
  const bool look_for_odd_missing = o.text_semantics && log_scale && !o.call_geno && !normalised;
  if (look_for_odd_missing) {
    HIP_TRY(c, c->d_odd_missing.resize(1));
    HIP_TRY(c, hipMemsetAsync(c->d_odd_missing.p, 0, sizeof(int), c->stream));
    a.odd_missing = c->d_odd_missing.p;
    a.missing_canon = replay_missing_raw_text();
  }
  if (on_device) {
    a.raw = gl;
    a.maf_in = maf;
    a.site0 = 0;
    a.n_sites = n_sites;
    HIP_TRY(c, launch_prep(a, c->stream));
  } else {
    const uint64_t site_bytes = n_ind * 3 * sizeof(double);
    uint64_t stage_bytes = 64ull << 20;
    if (const char *e = test_knob("STAGE_BYTES")) stage_bytes = std::strtoull(e, nullptr, 10);  // tests: force many chunks
    uint64_t chunk = stage_bytes / site_bytes;
    if (chunk < 1) chunk = 1;
    if (chunk > n_sites) chunk = n_sites;
    DevBuf<double> stage[2], maf_stage[2];
    hipEvent_t prepped[2] = {nullptr, nullptr};
    for (int k = 0; k < 2; ++k) {
      HIP_TRY(c, stage[k].resize(chunk * n_ind * 3));
      if (normalised) HIP_TRY(c, maf_stage[k].resize(chunk));
      HIP_TRY(c, hipEventCreateWithFlags(&prepped[k], hipEventDisableTiming));
    }
    int rc = NGSLD_OK;
    uint64_t k = 0;
    if (trace) std::fprintf(stderr, "[trace] set_geno: device buffers + staging allocated at %.2f ms\n", ms_since());
    for (uint64_t s0 = 0; s0 < n_sites && rc == NGSLD_OK; s0 += chunk, ++k) {
      const int b = (int)(k & 1);
      const uint64_t m = std::min(chunk, n_sites - s0);
      hipError_t e = hipSuccess;
      if (k >= 2) e = hipEventSynchronize(prepped[b]);  // the prep kernel that last read this buffer is done
      if (e == hipSuccess)
        e = hipMemcpyAsync(stage[b].p, gl + s0 * n_ind * 3, m * site_bytes, hipMemcpyHostToDevice, c->copy_stream);
      if (e == hipSuccess && normalised)
        e = hipMemcpyAsync(maf_stage[b].p, maf + s0, m * sizeof(double), hipMemcpyHostToDevice, c->copy_stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->copy_stream);
      if (trace) std::fprintf(stderr, "[trace] set_geno: chunk %llu (%llu sites) on the device at %.2f ms\n", (unsigned long long)k, (unsigned long long)m, ms_since());
      a.raw = stage[b].p;
      a.maf_in = normalised ? maf_stage[b].p : nullptr;
      a.site0 = s0;
      a.n_sites = m;
      if (e == hipSuccess) e = launch_prep(a, c->stream);
      if (e == hipSuccess) e = hipEventRecord(prepped[b], c->stream);
      if (e != hipSuccess) rc = hip_fail(c, e, "chunked genotype upload");
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (trace) std::fprintf(stderr, "[trace] set_geno: last prep kernel done at %.2f ms\n", ms_since());
    for (int q = 0; q < 2; ++q) {
      stage[q].release();
      maf_stage[q].release();
      if (prepped[q]) (void)hipEventDestroy(prepped[q]);
    }
    if (trace) std::fprintf(stderr, "[trace] set_geno: staging buffers freed at %.2f ms\n", ms_since());
    if (rc != NGSLD_OK) return rc;
    if (e != hipSuccess) return hip_fail(c, e, "chunked genotype upload");
  }
  // degenerate sites (sites whose pairs the exact-order replay settles anyway: the pair kernels leave their EM out) -- looked
  // for only where the device-side replay of likelihood matrices can take their pairs
  c->h_skip_count = 0;
  // The marks are taken for every per-individual kernel (they also tell ngsld_run that the matrix is un-called: run_grouped); the
  // pair kernels SKIP the marked sites' EM only where that was measured to pay (profiles/r06/skip): one wavefront per pair +6.5 %
  // at 500 individuals, two +4.6 % at 1,000; not the lockstep kernel -- a wavefront is spared only what all its groups skip: -7 %
  // at 100 -- and not four wavefronts per pair: at 2,000 individuals a replayed pair costs 7x a computed one and the pairs
  // marked without need outweigh the EM saved, -6 %.
  const bool look_for_skip = c->skip_on && c->replay_on && c->replay_device && c->exact_mode != 0;
  c->skip_kernels = cfg.kernel == kRun || (cfg.kernel == kMulti && cfg.form == 0 && cfg.waves == 2);
  if (look_for_skip) {
    HIP_TRY(c, c->d_skip.resize(n_sites));
    HIP_TRY(c, c->d_skip_count.resize(1));
    HIP_TRY(c, hipMemsetAsync(c->d_skip_count.p, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(c, launch_site_skip(c->d_planes.p, 3ull * c->np, c->np, (uint32_t)n_ind, ignore_miss, c->d_maf.p, n_sites, c->d_skip.p,
                                c->d_skip_count.p, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&c->h_skip_count, c->d_skip_count.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  } else {
    c->d_skip.release();
  }
  HIP_TRY(c, launch_pack_scalars(c->d_maf.p, c->d_mean.p, c->d_rsx.p, c->d_skip.p, c->d_sc4.p, n_sites, c->stream));
  // Is every likelihood triple a called genotype or "no data" (text genotypes, --call_geno)?  Then the pairs run on the
  // 16 genotype-combination counts instead of the individuals (ld_pair_hard.hip).  NGSLD_TEST_HARD_KERNEL=0: never (A/B, tests).
  int &all_hard = c->h_all_hard;  // (ctx-owned: the asynchronous copies below must not target a stack frame an early return leaves)
  all_hard = 0;
  const char *hk = test_knob("HARD_KERNEL");
  const bool try_hard = n_ind <= kHardMaxInd && !o.per_individual_only && !(hk != nullptr && std::strcmp(hk, "0") == 0);
  if (try_hard) {
    c->mask_words = (uint32_t)((n_ind + 63) / 64);
    HIP_TRY(c, c->d_hard_masks.resize((size_t)n_sites * 4 * c->mask_words));
    HIP_TRY(c, c->d_hard_u.resize(n_sites));
    HIP_TRY(c, c->d_all_hard.resize(1));
    all_hard = 1;
    HIP_TRY(c, hipMemcpyAsync(c->d_all_hard.p, &all_hard, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, launch_classify_hard(c->d_planes.p, 3ull * c->np, c->np, (uint32_t)n_ind, n_sites, c->d_hard_masks.p,
                                    c->d_hard_u.p, c->d_all_hard.p, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&all_hard, c->d_all_hard.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (trace) std::fprintf(stderr, "[trace] set_geno: called-genotype check enqueued at %.2f ms\n", ms_since());
  }
  c->h_maf.resize(n_sites);
  int &status = c->h_prep_status;
  status = 0;
  HIP_TRY(c, hipMemcpyAsync(c->h_maf.data(), c->d_maf.p, n_sites * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(&status, c->d_status.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->h_odd_missing = 1;
  if (look_for_odd_missing)
    HIP_TRY(c, hipMemcpyAsync(&c->h_odd_missing, c->d_odd_missing.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->missing_canonical = look_for_odd_missing && c->h_odd_missing == 0;
  if (trace) std::fprintf(stderr, "[trace] set_geno: scalars packed, called-genotype check, maf on the host at %.2f ms\n", ms_since());
  if (status == NGSLD_ERR_NAN) return fail(c, NGSLD_ERR_NAN, "NaN found! Is the file format correct?");
  if (try_hard && all_hard) {
    c->cfg.kernel = kHard;
    c->cfg.group = 16;
    c->cfg.slots = 1;
    c->cfg.waves = 1;
  } else {
    c->d_hard_masks.release();
    c->d_hard_u.release();
  }
  c->have_geno = true;
  return NGSLD_OK;
}
}  // namespace eng
}  // namespace ngsld

extern "C" {

const char *ngsld_version(void) { return "ngsld-amd 0.4.0 (gfx950; reference ngsLD 1.2.1)"; }

int ngsld_create(int device, ngsld_ctx **out) {
  if (out == nullptr) return NGSLD_ERR_INVALID;
  *out = nullptr;
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev == 0) {
    g_create_error = std::string("no HIP device available (") + hipGetErrorString(e) +
                     "); this library has no CPU fallback";
    return NGSLD_ERR_DEVICE;
  }
  if (device < 0 || device >= n_dev) {
    g_create_error = "device index out of range";
    return NGSLD_ERR_INVALID;
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
    return NGSLD_ERR_DEVICE;
  }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 (MI355X) only";
    return NGSLD_ERR_DEVICE;
  }
  ngsld_ctx *c = new (std::nothrow) ngsld_ctx();
  if (c == nullptr) return NGSLD_ERR_NOMEM;
  c->device = device;
  if (prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
  if (const char *k = std::getenv("NGSLD_PAIR_KERNEL")) {
    // "multi": several wavefronts per pair from 513 individuals on; "ab": 513..1024 individuals on ONE wavefront per pair, EM
    // step in its a/b form (pair_config picks between them, and the ten-slot run kernel, by measurement); "stream": beyond
    // 5,120 individuals the plain streaming kernel instead of the one that keeps the candidate's vector in registers
    c->kernel_choice = std::strcmp(k, "multi") == 0 ? kChooseMulti
                       : (std::strcmp(k, "ab") == 0 ? kChooseAB
                          : (std::strcmp(k, "stream") == 0 ? kChoosePlainStream : (std::strcmp(k, "abm") == 0 ? kChooseABMulti
                                                               : (std::strcmp(k, "bres") == 0 ? kChooseResidentStream : kChooseAuto))));
  }
  if (const char *k = test_knob("BATCH_PAIRS")) {  // tests: many small batches through ngsld_run
    const uint64_t v = std::strtoull(k, nullptr, 10);
    if (v > 0) {
      c->batch_pairs = v;
      c->batch_pairs_set = true;
    }
  }
  if (const char *k = std::getenv("NGSLD_REPLAY")) c->replay_on = std::strcmp(k, "0") != 0;  // A/B, tests
  if (const char *k = std::getenv("NGSLD_REPLAY_THREADS")) c->replay_threads = std::atoi(k);
  if (const char *k = std::getenv("NGSLD_REPLAY_DEVICE")) c->replay_device = std::strcmp(k, "0") != 0;
  if (const char *k = std::getenv("NGSLD_REPLAY_SKIP")) c->skip_on = std::strcmp(k, "0") != 0;  // A/B, tests
  if (const char *k = std::getenv("NGSLD_EXACT_STORE")) c->exact_mode = std::max(0, std::min(2, std::atoi(k)));  // ngsld_set_exact_store
  if (const char *k = test_knob("RUN_DIRECT")) c->run_direct = std::strcmp(k, "0") != 0;  // A/B, tests (see ngsld_ctx)
  if (const char *k = test_knob("RUN_TAPER")) c->run_taper = std::strcmp(k, "0") != 0;
  if (const char *k = test_knob("RUN_STREAMS")) c->run_streams = std::atoi(k) >= 2 ? 2 : 1;
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&c->stream)) != hipSuccess ||
      (e = hipStreamCreate(&c->stream2)) != hipSuccess || (e = hipStreamCreate(&c->copy_stream)) != hipSuccess) {
    g_create_error = std::string("stream setup: ") + hipGetErrorString(e);
    delete c;
    return NGSLD_ERR_DEVICE;
  }
  for (int k = 0; k < ngsld_ctx::kSlots; ++k) {
    (void)hipEventCreateWithFlags(&c->ev_kernel_done[k], hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->ev_copy_done[k], hipEventDisableTiming);
  }
  *out = c;
  return NGSLD_OK;
}

void ngsld_destroy(ngsld_ctx *c) {
  if (c == nullptr) return;
  if (c->reserve_thread.joinable()) c->reserve_thread.join();
  stop_exact_store(c);
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  if (c->exact_stream) (void)hipStreamDestroy(c->exact_stream);
  c->d_planes.release(); c->d_maf.release(); c->d_mean.release(); c->d_rsx.release(); c->d_sc4.release(); c->d_runs.release(); c->d_skip.release(); c->d_skip_count.release();
  c->d_hard_masks.release(); c->d_hard_u.release(); c->d_all_hard.release();
  c->d_xplanes.release(); c->d_xmaf.release(); c->d_xT.release(); c->d_xdepth.release(); c->d_xperm.release(); c->h_xstage[0].release(); c->h_xstage[1].release();
  c->lane_scratch_dev.release();
  for (int k = 0; k < ngsld_ctx::kSlots; ++k) c->lane_scratch[k].release();
  c->d_labels.release(); c->d_scan_tmp.release(); c->d_scan_tmp_b.release(); c->d_label_off.release(); c->d_cum.release(); c->d_infc.release();
  for (int k = 0; k < ngsld_ctx::kSlots; ++k) {
    c->d_text[k].release(); c->d_lens[k].release(); c->d_offs[k].release(); c->d_text_meta[k].release();
    c->h_text[k].release(); c->h_text_meta[k].release();
  }
  c->d_status.release(); c->d_row_off.release(); c->d_item_off.release(); c->d_row_end.release();
  c->d_row_seed.release(); c->d_row_count.release(); c->d_keep.release(); c->d_items.release();
  for (int k = 0; k < ngsld_ctx::kSlots; ++k) {
    c->d_std[k].release(); c->d_ext[k].release(); c->h_std[k].release(); c->h_ext[k].release();
    if (c->ev_kernel_done[k]) (void)hipEventDestroy(c->ev_kernel_done[k]);
    if (c->ev_copy_done[k]) (void)hipEventDestroy(c->ev_copy_done[k]);
  }
  for (auto &ev : c->ev_pool) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->text_stream) (void)hipStreamDestroy(c->text_stream);
  for (int k = 0; k < 3; ++k)
    if (c->ev_scan_done[k]) (void)hipEventDestroy(c->ev_scan_done[k]);
  if (c->replay_stream) (void)hipStreamDestroy(c->replay_stream);
  delete c;
}

const char *ngsld_last_error(const ngsld_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int ngsld_set_geno_raw(ngsld_ctx *c, const double *gl_raw, uint64_t n_sites, uint64_t n_ind, int log_scale,
                       int ignore_miss_data, int on_device) try {
  ngsld_geno_opts o{};
  o.log_scale = log_scale;
  o.ignore_miss_data = ignore_miss_data;
  o.on_device = on_device;
  return set_geno_common(c, gl_raw, nullptr, n_sites, n_ind, o, false);
} NGSLD_CATCH(c)

int ngsld_set_geno_raw_opts(ngsld_ctx *c, const double *gl_raw, uint64_t n_sites, uint64_t n_ind,
                            const ngsld_geno_opts *opts) try {
  if (opts == nullptr) return c ? fail(c, NGSLD_ERR_INVALID, "opts is NULL") : NGSLD_ERR_INVALID;
  return set_geno_common(c, gl_raw, nullptr, n_sites, n_ind, *opts, false);
} NGSLD_CATCH(c)

int ngsld_set_geno_lkl(ngsld_ctx *c, const double *geno_lkl, const double *maf, uint64_t n_sites, uint64_t n_ind,
                       int on_device) try {
  ngsld_geno_opts o{};
  o.on_device = on_device;
  return set_geno_common(c, geno_lkl, maf, n_sites, n_ind, o, true);
} NGSLD_CATCH(c)

int ngsld_get_maf(ngsld_ctx *c, double *maf_out) try {
  if (c == nullptr || maf_out == nullptr) return NGSLD_ERR_INVALID;
  if (!c->have_geno) return fail(c, NGSLD_ERR_INVALID, "no genotype data set");
  std::memcpy(maf_out, c->h_maf.data(), c->n_sites * sizeof(double));
  return NGSLD_OK;
} NGSLD_CATCH(c)

int ngsld_set_pos_dist(ngsld_ctx *c, const double *pos_dist) try {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  if (!c->have_geno) return fail(c, NGSLD_ERR_INVALID, "set the genotype data before the positions");
  c->planned = false;
  if (pos_dist == nullptr)
    c->h_pos_dist.assign(c->n_sites, std::numeric_limits<double>::infinity());  // ngsLD.cpp:134
  else
    c->h_pos_dist.assign(pos_dist, pos_dist + c->n_sites);
  return NGSLD_OK;
} NGSLD_CATCH(c)
const char *ngsld_pair_kernel(const ngsld_ctx *c) {
  if (c == nullptr || !c->have_geno) return "";
  switch (effective_kernel(c->cfg, c->params.ignore_miss_data != 0)) {
    case kGroup: return "group";
    case kMulti: return c->cfg.form == 1 ? "multi-ab" : "multi";
    case kStream: return "stream";
    case kRun: return "run";
    case kHard: return "hard";
    case kRunAB: return "ab";
    default: return "";
  }
}

int ngsld_describe_dispatch(uint64_t n_ind, int ignore_miss_data, char *buf, size_t buf_len) {
  if (buf == nullptr || buf_len == 0) return NGSLD_ERR_INVALID;
  buf[0] = 0;
  PairConfig cfg;
  const bool masked = ignore_miss_data != 0;
  if (!pair_config(n_ind, &cfg, kChooseAuto, masked)) return NGSLD_ERR_UNSUPPORTED;
  const char *family = "";
  int slots = cfg.slots, waves = cfg.waves;
  switch (effective_kernel(cfg, masked)) {
    case kGroup: family = "group"; break;
    case kMulti:
      family = cfg.form == 1 ? "multi-ab" : "multi";
      if (cfg.form == 0) multi_shape(cfg, masked, &slots, &waves);
      break;
    case kStream: family = "stream"; break;
    case kRun: family = "run"; break;
    case kRunAB: family = "ab"; break;
    default: return NGSLD_ERR_UNSUPPORTED;
  }
  const int n = std::snprintf(buf, buf_len, "%s %dx%d lanes=%d np=%u", family, waves, slots, cfg.group, cfg.np);
  return (n < 0 || (size_t)n >= buf_len) ? NGSLD_ERR_INVALID : NGSLD_OK;
}

int ngsld_set_tuning(ngsld_ctx *c, uint32_t pairs_per_item, uint64_t batch_pairs) {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  if (pairs_per_item) {
    c->pairs_per_item = pairs_per_item;
    c->planned = false;
  }
  if (batch_pairs) {
    c->batch_pairs = batch_pairs;
    c->batch_pairs_set = true;
  }
  return NGSLD_OK;
}

int ngsld_selftest(ngsld_ctx *c) try {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  (void)hipGetLastError();  // (a failure some earlier call already reported must not surface as a launch's "last error")
  std::vector<double> in(320), out(71, 0.0);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  for (auto &v : in) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    v = (double)(st >> 11) / 9007199254740992.0 + 1e-3;
  }
  for (int l = 0; l < 64; ++l) in[256 + l] *= std::pow(10.0, -(l % 30));
  DevBuf<double> d_in, d_out;
  HIP_TRY(c, d_in.resize(in.size()));
  HIP_TRY(c, d_out.resize(out.size()));
  HIP_TRY(c, hipMemcpy(d_in.p, in.data(), in.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(c, launch_selftest(d_in.p, d_out.p, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(out.data(), d_out.p, out.size() * sizeof(double), hipMemcpyDeviceToHost));
  d_in.release();
  d_out.release();
  for (int k = 0; k < 7; ++k) {  // 0..3: wave_sum4, 4..6: wave_sum3 (the matrix-pipe reduction of the EM loop)
    const int v = k < 4 ? k : k - 4;
    const double got = k < 4 ? out[k] : out[68 + v];
    long double ref = 0;
    for (int l = 0; l < 64; ++l) ref += in[v * 64 + l];
    if (!(std::fabs((double)(got - ref)) <= 1e-14 * std::fabs((double)ref))) {
      char buf[160];
      std::snprintf(buf, sizeof(buf), "%s value %d: got %.17g expected %.17Lg", k < 4 ? "wave_sum4" : "wave_sum3", v, got, ref);
      return fail(c, NGSLD_ERR_DEVICE, buf);
    }
  }
  for (int l = 0; l < 64; ++l) {
    const double ref = 1.0 / in[256 + l];
    if (std::fabs(out[4 + l] - ref) > 4.5e-16 * ref) {
      char buf[160];
      std::snprintf(buf, sizeof(buf), "rcp_refined lane %d: got %.17g expected %.17g", l, out[4 + l], ref);
      return fail(c, NGSLD_ERR_DEVICE, buf);
    }
  }
  return NGSLD_OK;
} NGSLD_CATCH(c)
uint64_t ngsld_slab_sites_for_budget(uint64_t n_ind, uint64_t budget_bytes) { return ngsld_sites_for_budget(n_ind, budget_bytes, 3); }

uint64_t ngsld_sites_for_budget(uint64_t n_ind, uint64_t budget_bytes, int matrix_copies) {
  PairConfig cfg, cfg_masked;
  // (the engine's own default selection: the slabs hold what it will allocate -- the wider of the two layouts a cohort size
  // can get, with and without --ignore_miss_data)
  if (!pair_config(n_ind, &cfg) || !pair_config(n_ind, &cfg_masked, kChooseAuto, true)) return 0;
  if (cfg_masked.np > cfg.np) cfg.np = cfg_masked.np;
  // per context: planes (24*np per site) + maf/mean/rsx + row tables (~64 B per site), three record slots of
  // batch_pairs records, two staging chunks of 256 MiB, items; the fixed part is rounded up generously
  const uint64_t fixed = ((uint64_t)ngsld_ctx::kSlots * (1ull << 23) * (sizeof(ngsld_rec_std) + sizeof(ngsld_rec_ext))) + (768ull << 20);
  const uint64_t per_ctx = budget_bytes / 2;
  if (per_ctx <= fixed) return 0;
  // (the matrix three times: the planes, and the exact store of the device-side replay with its individual-major copy, which
  // un-called input has built -- engine_replay.hip; once: the planes, what a run cannot do without)
  const uint64_t copies = matrix_copies < 1 ? 1 : (matrix_copies > 3 ? 3 : (uint64_t)matrix_copies);
  return (per_ctx - fixed) / (24ull * copies * cfg.np + 64ull);
}

}  // extern "C"

// A cap on what this process may take of a device's memory (the command line's --max_gpu_mem): counted from the moment it is set
// -- free memory then, against free memory now -- and looked at where the library allocates what it can do without: the exact
// store of the device-side replay and its individual-major copy (engine_replay.hip).  0: no cap but the device's own.
namespace ngsld {
namespace eng {
static std::atomic<uint64_t> g_mem_budget{0}, g_mem_base_free{0};
bool room_for(uint64_t need_bytes, uint64_t device_margin, uint64_t budget_margin) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if ((uint64_t)free_b < need_bytes + device_margin) return false;
  const uint64_t budget = g_mem_budget.load();
  if (budget == 0) return true;
  const uint64_t base = g_mem_base_free.load(), used = base > (uint64_t)free_b ? base - (uint64_t)free_b : 0;
  return used + need_bytes + budget_margin <= budget;
}
}  // namespace eng
}  // namespace ngsld

extern "C" {

int ngsld_set_memory_budget(int device, uint64_t bytes) {
  uint64_t free_b = 0;
  if (ngsld_device_memory(device, &free_b, nullptr) != NGSLD_OK) return NGSLD_ERR_DEVICE;
  ngsld::eng::g_mem_base_free.store(free_b);
  ngsld::eng::g_mem_budget.store(bytes);
  return NGSLD_OK;
}

int ngsld_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
  // (a failure is this call's answer -- "no such device" is how callers count devices -- and must not stay behind as the
  // runtime's last error for whoever asks next: torch raised "invalid device ordinal" on its first allocation)
  if (hipSetDevice(device) != hipSuccess) {
    (void)hipGetLastError();
    return NGSLD_ERR_DEVICE;
  }
  size_t f = 0, t = 0;
  if (hipMemGetInfo(&f, &t) != hipSuccess) {
    (void)hipGetLastError();
    return NGSLD_ERR_DEVICE;
  }
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return NGSLD_OK;
}

}  // extern "C"
