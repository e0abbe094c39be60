// engine.h -- what the translation units of the engine share (NOT part of the C-ABI: include/ngsld.h is): the context, its
// buffers, the error plumbing, and the handful of functions one unit calls in another.
//   engine.hip         contexts, the genotype matrix (upload + per-site prep), small queries
//   engine_plan.hip    the pair-space plan: the s2 walk of calc_pair_LD for every s1 (ngsLD.cpp:240-282), items, runs
//   engine_run.hip     ngsld_run / ngsld_run_device: batches, pair-kernel launches, device-side TSV
//   engine_replay.hip  exact-order replay: flag lists, host threads, the device-side replays, the exact-value store
#pragma once

#include "knobs.h"

#include <hip/hip_runtime.h>
#include <sched.h>
#include <dlfcn.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/ngsld.h"
#include "ld_common.h"
#include "ld_dispatch.h"
#include "ld_prep.h"
#include "ld_replay.h"
#include "ld_text.h"
#include "replay.h"
#include "taus.h"

namespace ngsld {
namespace eng {
extern thread_local std::string g_create_error;

// roctx ranges around the phases of a run (upload / prep / plan / pair kernels / D2H / replay / sink), so that a
// `rocprofv3 --marker-trace --kernel-trace` timeline reads as phases.  The marker library (rocprofiler-sdk-roctx, or the
// older libroctx64) is resolved on first use and is not a link-time dependency: without it the ranges are no-ops.
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    if (const char *e = std::getenv("NGSLD_ROCTX"))
      if (std::strcmp(e, "0") == 0) return;
    for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void *lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);  // only if the profiler (or the caller) already loaded it
      if (lib == nullptr && std::getenv("NGSLD_ROCTX") != nullptr) lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib == nullptr) continue;
      push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
      if (push && pop) return;
      push = nullptr;
      pop = nullptr;
    }
  }
};
inline Roctx &roctx() {
  static Roctx r;
  return r;
}
struct Range {  // scope = one named phase
  bool on;
  explicit Range(const char *name) : on(roctx().push != nullptr) {
    if (on) roctx().push(name);
  }
  ~Range() {
    if (on) roctx().pop();
  }
  Range(const Range &) = delete;
  Range &operator=(const Range &) = delete;
};

// (both buffers free themselves: an early return from a function that holds one as a local leaks nothing)
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  hipError_t resize(size_t count) {
    if (count <= n && p != nullptr) return hipSuccess;
    release();
    if (count == 0) return hipSuccess;
    hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

// Pinned host memory: hipHostMalloc.  NGSLD_PIN_REGISTER=1 (opt-in; the drop-in binary opts in, cli_main.cpp): buffers of two
// megabytes and more as 2 MB-aligned anonymous memory on transparent huge pages, registered with the runtime (hipHostRegister,
// mapped: the pair kernels write records through it, run_direct).  That is 4x cheaper to get and to give back -- 400 MB: 17 +
// 15 ms against 72-92 + 41-55 ms, 1.2 GB: 50 + 44 ms against 220-270 + 150-164 ms, copies and kernel writes at the same
// 56-57 GB/s (tools/probe_pin.hip, profiles/r04/probe_pin.txt) -- and takes the binary on configs[2] from 0.97-1.05 to
// 0.86-0.97 s (pin_ab.txt), configs[4] at full size from 25.9 / 19.7 to 22.7 / 16.7 s.
// Why it is not the library's default.  Its first form took the block from malloc (posix_memalign) and was the default for
// five commits: two of the three runs of the whole GPU suite made with it -- one process that lives nine minutes, creates
// hundreds of contexts and forks children -- died of "Memory access fault by GPU node-2 ... on address 0x56bd21b36000", an
// address on the process' brk heap, a few tests after one that forks (profiles/r04/late3/).  A fork() write-protects the
// parent's private pages for copy-on-write under the device's mapping, and a freed heap block is handed out again to
// anybody.  The block is now a mapping of its own with MADV_DONTFORK (what RDMA libraries do to registered memory): four
// whole-suite runs since, two with it on in the test process, none died (pin_dontfork_suite_runs.txt).  Registered memory still
// is ordinary anonymous memory whose pages the kernel may migrate under the driver's notifier, hipHostMalloc memory is the
// driver's own: a host application gets the latter unless it asks.
template <typename T>
struct PinBuf {
  T *p = nullptr;
  size_t n = 0;
  void *map_base = nullptr;  // registered variant: the anonymous mapping the buffer sits in (null: hipHostMalloc memory)
  size_t map_len = 0;
  PinBuf() = default;
  PinBuf(const PinBuf &) = delete;
  PinBuf &operator=(const PinBuf &) = delete;
  ~PinBuf() { release(); }
  hipError_t resize(size_t count) {
    if (count <= n && p != nullptr) return hipSuccess;
    release();
    if (count == 0) return hipSuccess;
    const size_t huge = (size_t)2 << 20, want = count * sizeof(T);
    static const bool use_register = [] {
      const char *e = std::getenv("NGSLD_PIN_REGISTER");
      return e != nullptr && std::strcmp(e, "1") == 0;
    }();
    if (use_register && want >= huge) {
      // a mapping of its own (never the malloc heap: a freed block there is handed out again, to anybody), 2 MB aligned, on
      // huge pages, and kept out of children (MADV_DONTFORK: a fork() would write-protect the pages for copy-on-write under
      // the device's mapping -- what registered memory of RDMA libraries is protected from the same way)
      const size_t bytes = (want + huge - 1) / huge * huge, len = bytes + huge;
      void *m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (m != MAP_FAILED) {
        void *q = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(m) + huge - 1) / huge * huge);
        (void)madvise(q, bytes, MADV_HUGEPAGE);
        if (madvise(q, bytes, MADV_DONTFORK) == 0 && hipHostRegister(q, bytes, hipHostRegisterMapped) == hipSuccess) {
          p = static_cast<T *>(q);
          n = count;
          map_base = m;
          map_len = len;
          return hipSuccess;
        }
        (void)hipGetLastError();
        (void)munmap(m, len);
      }
    }
    hipError_t e = hipHostMalloc((void **)&p, want, hipHostMallocDefault);
    if (e == hipSuccess) n = count;
    return e;
  }
  void release() {
    if (p && map_base) {
      (void)hipHostUnregister(p);
      (void)munmap(map_base, map_len);
    } else if (p) {
      (void)hipHostFree(p);
    }
    p = nullptr;
    n = 0;
    map_base = nullptr;
    map_len = 0;
  }
};
// A few parked host threads for the exact-order replay: a launch of 1e8 pairs flags a few dozen pairs, 0.2 ms of arithmetic
// each -- spawning a thread per pair cost more than the pairs (0.6 ms of a 1.1 ms ngsld_finish_device).  Threads are created
// on first use and live as long as the context; run(T, fn) executes fn(0 .. T-1), fn(0) on the calling thread.
class ReplayPool {
 public:
  ~ReplayPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
  }
  template <typename F>
  void run(int T, F &&fn) {
    if (T <= 1) {
      fn(0);
      return;
    }
    while ((int)threads_.size() < T - 1) {
      const int id = (int)threads_.size() + 1;
      threads_.emplace_back([this, id] { loop(id); });
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = [&fn](int t) { fn(t); };
      n_ = T;
      left_ = T - 1;
      ++epoch_;
    }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return left_ == 0; });
    job_ = nullptr;
  }

 private:
  void loop(int id) {
    uint64_t seen = 0;
    for (;;) {
      std::function<void(int)> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
        if (stop_) return;
        seen = epoch_;
        if (id >= n_) continue;  // (this job uses fewer threads)
        job = job_;
      }
      job(id);
      std::lock_guard<std::mutex> lk(mu_);
      if (--left_ == 0) done_.notify_all();
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::function<void(int)> job_;
  int n_ = 0, left_ = 0;
  uint64_t epoch_ = 0;
  bool stop_ = false;
};

}  // namespace eng
}  // namespace ngsld

using namespace ngsld;       // (an internal header: only the engine_*.hip units include it)
using namespace ngsld::eng;

// One pair a text batch leaves to the host's exact-order replay, written by the device right behind the batch's row lengths
// (engine_replay.hip: flag_rows_to_host_kernel): with it the host replays the pair and overwrites the row's value columns in
// the text it has received -- nothing goes back to the device, no kernel is submitted when the batch is consumed.
struct FlagRow {
  uint64_t rec, off;  // record index within the batch; first byte of its row in the batch's text
  uint32_t len, s1, s2, pad;
};
constexpr uint32_t kFlagRowsCap = 1024;

struct ngsld_ctx {
  int device = 0;
  hipStream_t stream = nullptr, stream2 = nullptr, copy_stream = nullptr;
  hipStream_t text_stream = nullptr;  // the row-writing kernels of text batches (made on first use: ngsld_run with text output)
  hipEvent_t ev_scan_done[3] = {nullptr, nullptr, nullptr};  // a text batch's lengths and prefix sums are there
  std::string err;

  // data
  uint64_t n_sites = 0, n_ind = 0;
  uint32_t np = 0;
  PairConfig cfg{};
  bool have_geno = false;
  DevBuf<double> d_planes, d_maf, d_mean, d_rsx, d_sc4;
  DevBuf<int> d_status;
  std::vector<double> h_maf, h_pos_dist;
  // hard-called matrices (kHard): per-site genotype bit sets
  DevBuf<uint64_t> d_hard_masks;
  DevBuf<double> d_hard_u;
  DevBuf<int> d_all_hard, d_odd_missing;
  int h_odd_missing = 0;
  bool missing_canonical = false;  // text genotypes: every individual without data is the reader's own triple (PrepArgs::odd_missing)
  int h_all_hard = 0, h_prep_status = 0;
  uint32_t mask_words = 0;
  // degenerate sites (ld_prep.hip, site_skip_kernel): marked per site, in sc4[.][3]; how many there are
  DevBuf<uint8_t> d_skip;
  DevBuf<uint32_t> d_skip_count;
  uint32_t h_skip_count = 0;
  bool skip_on = true;  // NGSLD_REPLAY_SKIP=0: the pair kernels run the EM of every pair (A/B, tests)
  bool skip_kernels = false;  // this cohort's pair kernel has a variant that leaves the marked sites' EM out (engine.hip)

  // plan
  bool planned = false;
  ngsld_params params{};
  std::vector<uint64_t> h_row_off, h_item_off;
  std::vector<uint32_t> h_row_end;
  std::vector<uint8_t> h_keep;
  std::vector<Item> h_items;  // host copy for the sink (which pairs each record belongs to)
  std::vector<uint64_t> h_run_off;  // run kernel: runs before each row
  uint64_t run_len = 0;             // items per run the list was cut with (0: no list)
  std::vector<uint64_t> run_ends;   // ... and the launch boundaries (rows) whose tails it was shaped for
  int n_cus = 256;                  // compute units of the device (hipDeviceProp_t::multiProcessorCount)
  DevBuf<Run> d_runs;
  DevBuf<uint64_t> d_row_off, d_item_off, d_row_seed, d_row_count;
  DevBuf<uint32_t> d_row_end;
  DevBuf<uint8_t> d_keep;
  DevBuf<Item> d_items;
  uint64_t n_items = 0;

  // tuning
  // kernel family selection for tests and A/B runs: NGSLD_PAIR_KERNEL=multi | ab (PairChoice)
  int kernel_choice = kChooseAuto;
  uint32_t pairs_per_item = 16;
  uint64_t batch_pairs = 1ull << 23;
  bool batch_pairs_set = false;  // by the caller (ngsld_set_tuning / NGSLD_TEST_BATCH_PAIRS): taken as it is

  // batch pipeline: three slots for record batches, two of them for text batches
  static constexpr int kSlots = 3;
  DevBuf<ngsld_rec_std> d_std[kSlots];
  DevBuf<ngsld_rec_ext> d_ext[kSlots];
  PinBuf<ngsld_rec_std> h_std[kSlots];
  PinBuf<ngsld_rec_ext> h_ext[kSlots];
  hipEvent_t ev_kernel_done[kSlots] = {nullptr, nullptr, nullptr}, ev_copy_done[kSlots] = {nullptr, nullptr, nullptr};
  // how record batches reach the host (ngsld_run without text output; NGSLD_TEST_RUN_DIRECT / NGSLD_TEST_RUN_TAPER):
  //   run_direct   the pair kernels write the records straight into the batch's pinned host buffers over the host link
  //                (72 B per pair at 2e8 pairs/s is 15 GB/s of posted writes; same-box A/B, profiles/r04/sink_ab.txt: the
  //                kernels take the same time) -- there is no device copy of the records and no D2H copy behind the last
  //                kernel.  Off: device buffers + a D2H copy per batch, and
  //   run_taper    the batches shrink towards the end of a run (a third of what is left, at least 2^19 pairs), so that the
  //                copy exposed behind the last kernel is small
  //   run_streams  1: one compute stream, every batch drains alone -- its last rows cut into short runs (build_runs), which
  //                takes the loss from 1.4 to ~0.4 ms per launch.  2 (opt-in, NGSLD_TEST_RUN_STREAMS=2): consecutive record
  //                batches on two compute streams HALF A BATCH OUT OF PHASE (the first batch is half a batch), so that
  //                whenever one stream's batch drains the other is in the middle of its own and fills the slots that fall
  //                free.  Measured on four boxes (profiles/r04/sink_rr*.txt, host-resident rate over the device-resident
  //                one, round robin in one process): 1.006 / 0.984 / 0.990 at 2^22 pairs per batch, 0.982 / 0.986 at 2^23 --
  //                when the dispatcher interleaves the two queues well it beats ONE launch, when it does not it loses to
  //                one stream (0.987-0.993): not the default.  (In phase -- equal batches on both, first tried -- the device
  //                shares itself evenly, both drain together: 501 ms on two streams, 500 on one, 484 as one launch.)
  bool run_direct = true, run_taper = true;
  int run_streams = 1;
  bool timed_overlap = false;  // the launches of the last run shared the device: their time is first start .. last end

  // device-side TSV (ngsld_set_text_output)
  bool text_mode = false, have_labels = false;
  uint64_t max_label = 6;  // "(null)"
  DevBuf<char> d_labels, d_text[kSlots], d_scan_tmp, d_scan_tmp_b;  // (_b: the second compute stream's scan space)
  DevBuf<uint64_t> d_label_off, d_lens[kSlots], d_offs[kSlots], d_text_meta[kSlots];  // meta: {total bytes, needs_host}
  DevBuf<double> d_cum;
  DevBuf<uint32_t> d_infc;
  PinBuf<char> h_text[kSlots];
  PinBuf<uint64_t> h_text_meta[kSlots];
  std::thread reserve_thread;  // ngsld_reserve_text_buffers: pins h_text[0..1] in the background; joined before their first use

  // exact-order replay of the pairs the kernels flag (replay.h)
  bool replay_on = true;
  ngsld_read_sites_fn replay_read = nullptr;  // the caller's raw values again (null: the device's planes are read back)
  void *replay_user = nullptr;
  const double *replay_matrix = nullptr;      // ... or the caller's own host array, read in place (ngsld_set_replay_matrix)
  std::mutex replay_mu;                       // serialises the source callback / the plane read-back
  // non-blocking: read-backs must not wait for the next batch's kernel.  Made on FIRST USE (replay_stream_of), not with the
  // context: it is needed by runs that flag more pairs than their list holds, or that replay without a registered source --
  // hardly ever -- while a stream costs 11 ms to create and a slot among the runtime's four hardware queues, which ALL of a
  // process' streams share (tools/probe_init.hip, profiles/r04/probe_init.txt, hw_queues_ab.txt).
  hipStream_t replay_stream = nullptr;
  std::mutex replay_stream_mu;
  ngsld_geno_opts gopts{};
  bool normalised = false;                    // data came through ngsld_set_geno_lkl
  DevBuf<uint32_t> d_flags[kSlots], d_flags_dev;   // [count, pad, list of the first flag_cap, one bit per record ...] per pipeline slot / for ngsld_run_device
  PinBuf<uint32_t> h_flags[kSlots], h_flags_dev;   // host copies of the HEAD (count + list): they travel with the batch's records / text meta
  PinBuf<uint32_t> h_flag_bits;                    // the bitmap, fetched only when a launch flagged more pairs than the list holds
  PinBuf<FlagRow> h_flag_rows[kSlots];             // text batches: where the rows of the listed pairs lie in the batch's text (send_flag_rows)
  uint32_t flag_cap[kSlots] = {0, 0, 0}, flag_cap_dev = 0;  // list entries of d_flags[k] / d_flags_dev as last reset
  bool replay_device = true;                       // flagged pairs replayed on the device where that is possible: called genotypes (ld_replay.hip), likelihoods (ld_replay_lkl.hip); NGSLD_REPLAY_DEVICE=0: host
  uint64_t replayed_on_device = 0;
  uint64_t text_rows_patched = 0;                  // rows of host-replayed pairs overwritten in the host's copy of a batch's text (last run)
  // The exact store of the device-side replay of LIKELIHOOD matrices (ld_replay_lkl.hip): normal-space likelihoods and est_maf
  // as the reference holds them when calc_pair_LD runs -- the HOST's libm, the sequential est_maf -- laid out like the planes.
  // Input that came through ngsld_set_geno_lkl is such a store already (the planes are the caller's values), and so are the
  // planes when no replay source is registered (the replay then runs on the device's own values, as the host's did); with a
  // source it is built by the replay threads from the caller's raw values, the first time a run flags more pairs than the host
  // should replay (exact_store_wanted), and kept until the matrix or its source changes.
  DevBuf<double> d_xplanes, d_xmaf;
  DevBuf<double> d_xT;                 // the store once more, individual-major, for the lane-per-pair kernel (ld_replay_lkl.hip)
  std::atomic<bool> xT_ready{false};
  DevBuf<uint32_t> d_xperm;            // ... where each site stands in it (rare sites first: ReplayLklArgs::xperm)
  DevBuf<uint32_t> d_xdepth;              // ... and how small its sites' likelihoods get (ReplayLklArgs::xdepth)
  struct LaneScratch {   // what the lane-per-pair replay of one launch needs: the located pairs and their sorted order
    DevBuf<ReplayEntry> list;
    DevBuf<uint64_t> keys_a, keys_b;
    DevBuf<uint32_t> vals_a, vals_b;
    DevBuf<char> temp;
    void release() { list.release(); keys_a.release(); keys_b.release(); vals_a.release(); vals_b.release(); temp.release(); }
  } lane_scratch[kSlots], lane_scratch_dev;  // per pipeline slot / for ngsld_run_device
  PinBuf<double> h_xstage[2];
  // A store that has to be BUILT is built by a thread of its own, in site order, while the run that asked for it goes on: a
  // launch's device-side replay needs the sites up to the end of its last row's window only (exact_frontier), and the build
  // (~0.25 us per individual and site) stays ahead of the pipeline (pair kernel + replay of the rows of those sites).  A run
  // that started the build waits for its end before it returns: the registered source is read during runs only.
  std::thread exact_thread;
  std::atomic<int> exact_state{0};          // 0 no store, 1 being built, 2 complete (every site + the lane kernel's copy where it fits), -1 the build failed
  std::atomic<uint64_t> exact_frontier{0};  // sites [0, exact_frontier) of d_xplanes / d_xmaf are on the device
  std::atomic<bool> exact_cancel{false};
  std::mutex exact_mu;                      // exact_cv, exact_rc, exact_msg
  std::condition_variable exact_cv;
  int exact_rc = 0;
  std::string exact_msg;
  hipStream_t exact_stream = nullptr;       // the builder's uploads
  ReplayPool exact_pool;                    // the builder's threads (replay_pool is the run's: host-only pairs beside the build)
  bool exact_ready = false, exact_alias = false;  // exact_ready: the run's view -- complete, errors collected
  bool exact_failed = false;           // the device had no room for this matrix' store: host replay
  int exact_mode = 1;                  // NGSLD_EXACT_STORE / ngsld_set_exact_store: 0 never (host replay only), 1 when it pays (default), 2 at the first flagged pair
  double exact_build_s = 0.0;          // host seconds the store of this matrix took to build (0: an alias, or not built)
  uint64_t host_replayed_total = 0;    // pairs the host threads replayed since the matrix was set (what the decision to build looks at)
  uint64_t flagged_pairs = 0;          // pairs the kernels of the last run flagged
  bool slot_dev_applied[kSlots] = {false, false, false};  // this slot's launch had the device-side replay right behind its pair kernels
  PinBuf<double> h_site_stage;                // plane read-back of one site (no source registered)
  DevBuf<uint64_t> d_patch_idx;
  DevBuf<ngsld_rec_std> d_patch_std;
  DevBuf<ngsld_rec_ext> d_patch_ext;
  DevBuf<char> d_scan_tmp2;                   // prefix sums taken again after a patch changed a row's length, beside the next batch's scan
  DevBuf<uint32_t> d_patch_s1, d_patch_s2;    // sites of the patched records (their rows' lengths are derived again)
  uint64_t replayed_pairs = 0, replayed_sites = 0;
  int replay_threads = 0;                     // 0 = min(32, the threads the process may really use)
  ReplayPool replay_pool;
  struct {
    bool pending = false;
    uint64_t s1_begin = 0, s1_end = 0;
    ngsld_rec_std *d_std = nullptr;
    ngsld_rec_ext *d_ext = nullptr;
    hipStream_t st = nullptr;
    bool dev_applied = false;
  } dev_run;                                  // the last ngsld_run_device, until ngsld_finish_device has looked at its flags
  bool dev_run_flag_text = false;             // ngsld_run_device's launches also flag what text output needs flagged (engine_run.hip, run_grouped)
  DevBuf<ngsld_rec_std> d_group_std[2];       // run_grouped: the records of a group of text batches, two groups in turn
  DevBuf<ngsld_rec_ext> d_group_ext[2];
  DevBuf<uint64_t> d_group_lens, d_group_offs, d_group_meta, d_group_first;  // ... its rows' lengths, their prefix sums, {total, .., .., overflow}, the batches' first records
  DevBuf<int> d_group_needs;                  // ... per batch: a value beyond the device formatter's range (the batch goes out as records)
  DevBuf<char> d_group_text;                  // ... its text
  PinBuf<uint64_t> h_group_bounds;            // ... where each batch's text begins (+ the total)
  PinBuf<int> h_group_needs;

  // timing of pair-kernel launches
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  hipStream_t timed_stream = nullptr;
  uint64_t timed_pairs = 0;
};

namespace ngsld {
namespace eng {


inline int fail(ngsld_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  return code;
}

inline int hip_fail(ngsld_ctx *c, hipError_t e, const char *what) {
  return fail(c, e == hipErrorOutOfMemory ? NGSLD_ERR_NOMEM : NGSLD_ERR_DEVICE,
              std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(c, call)                                \
  do {                                                  \
    hipError_t e_ = (call);                             \
    if (e_ != hipSuccess) return hip_fail(c, e_, #call); \
  } while (0)

// No exception crosses the C-ABI (include/ngsld.h): every entry point that allocates host memory is a function-try-block
// ending in this handler.  Work still in flight is waited for, so that buffers the caller owns are quiet on return.
inline int caught(ngsld_ctx *c, bool nomem) {
  if (c) {
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    c->planned = false;
  }
  return nomem ? fail(c, NGSLD_ERR_NOMEM, "out of host memory") : fail(c, NGSLD_ERR_INVALID, "unexpected C++ exception");
}
#define NGSLD_CATCH(ctx)                                        \
  catch (const std::bad_alloc &) { return caught(ctx, true); }  \
  catch (...) { return caught(ctx, false); }

inline int check_status(ngsld_ctx *c) {
  int status = 0;
  HIP_TRY(c, hipMemcpy(&status, c->d_status.p, sizeof(int), hipMemcpyDeviceToHost));
  if (status == NGSLD_ERR_MAF_RANGE) return fail(c, NGSLD_ERR_MAF_RANGE, "invalid allele frequencies");
  return NGSLD_OK;
}

// Rows per text batch (ngsld_run; a smaller NGSLD_TEST_BATCH_PAIRS / ngsld_set_tuning wins).  Round 4, configs[2] end to end on one
// box (profiles/r04/e2e_batch_size.txt): 2^21 1.42-1.46 s, 2^20 1.25-1.35 s, 2^19 1.23-1.25 s -- the loop itself takes the
// same 0.62 s whatever the count (a batch costs ~0.3 ms since its last rows go out as short runs and a replayed row no longer
// has every length derived again), while the two pinned buffers (2 x 400 MB at 2^21) cost 0.1 s to pin -- beside the matrix
// upload, which they slow -- and 0.06 s to give back.
constexpr uint64_t kTextBatchPairs = 1ull << 19;

// List entries of a launch of n records: a 256th of them (a called-genotype matrix flags one pair in ~4,000, a likelihood
// matrix one in 10^6), at least 4,096, at most 2^20 (8 MB of head to read back).
inline uint32_t flag_cap_for(uint64_t n) { return (uint32_t)std::min<uint64_t>(1ull << 20, std::max<uint64_t>(4096, n / 256)); }
inline size_t flag_head_bytes(uint32_t cap) { return (size_t)flag_head_words(cap) * sizeof(uint32_t); }
inline size_t flag_bitmap_words(uint64_t n) { return (size_t)((n + 31) / 32); }
inline size_t flag_words(uint64_t n, uint32_t cap) { return (size_t)flag_head_words(cap) + 2 * flag_bitmap_words(n); }

// ---- engine_plan.hip ----
void plan_rows(const std::vector<double> &pos_dist, const std::vector<double> &maf, const ngsld_params &p, uint64_t n,
               std::vector<uint32_t> &row_end);
int build_runs(ngsld_ctx *c, uint64_t run_len, const std::vector<uint64_t> &launch_ends);

// ---- engine_run.hip ----
hipError_t timed_launch(ngsld_ctx *c, const PairArgs &a, hipStream_t stream);
// flag_n: records the flag buffer was laid out for (its two bitmaps follow the list: ld_device.h)
PairArgs make_args(ngsld_ctx *c, uint64_t r0, uint64_t r1, ngsld_rec_std *d_std, ngsld_rec_ext *d_ext,
                   uint32_t *d_flags = nullptr, uint32_t flag_cap = 0, uint64_t flag_n = 0);

// ---- engine_replay.hip ----
// send_flag_rows: for the pairs a text batch leaves to the host (its flag list, or its host-only list behind a device-side
// replay -- at most kFlagRowsCap of them), which pair each is and where its row lies in the batch's text
int send_flag_rows(ngsld_ctx *c, const uint32_t *d_flags, uint32_t cap, bool dev_applied, const uint64_t *d_offs, const uint64_t *d_lens,
                   uint64_t n_records, uint64_t base, FlagRow *h_rows, hipStream_t st);
// the pairs (s1[k], s2[k]) in the reference's operation order on the host's threads -> out_std[k] / out_ext[k] (null: not wanted)
int replay_pairs_on_host(ngsld_ctx *c, const uint32_t *s1, const uint32_t *s2, size_t n, ngsld_rec_std *out_std, ngsld_rec_ext *out_ext);
unsigned usable_threads();
hipStream_t replay_stream_of(ngsld_ctx *c);
int ensure_host_items(ngsld_ctx *c);
int fetch_replay_site(ngsld_ctx *c, uint64_t s, std::vector<double> &tmp, ReplaySite *out);
// dev_applied: the device-side replay of likelihood matrices ran behind the launch -- what is left are its host-only pairs
int flagged_records(ngsld_ctx *c, const uint32_t *h_head, const uint32_t *d_flags, uint32_t cap, uint64_t n,
                    std::vector<uint64_t> &recs, bool dev_applied = false);
// likelihood matrices: can the flagged pairs of this context's runs be replayed on the device at all / right now?
// room on the device for `need_bytes` more: hipMemGetInfo's free memory (less device_margin) and, where a cap is set
// (ngsld_set_memory_budget), what the process has taken since against the cap (less budget_margin)
bool room_for(uint64_t need_bytes, uint64_t device_margin, uint64_t budget_margin);
bool lkl_device_eligible(const ngsld_ctx *c);
bool exact_store_is_free(const ngsld_ctx *c);
// the store there or on its way (device_replay_lkl may be asked for launches whose sites it covers)
inline bool exact_store_started(const ngsld_ctx *c) { return c->exact_state.load() >= 1; }
// starts the build where there is something to build (idempotent); the alias forms are complete at once
int start_exact_store(ngsld_ctx *c);
int reserve_device_run(ngsld_ctx *c, uint64_t n_records);  // ngsld_run_device's own buffers for launches of up to n records, once
// waits until sites [0, need_sites) are on the device (need_sites >= n_sites: until the store is complete); *have = false when
// there is no store to be had (no room on the device: host replay); an error of the build comes back as the return value
int wait_exact_store(ngsld_ctx *c, uint64_t need_sites, bool *have);
int ensure_exact_store(ngsld_ctx *c);  // start + wait for all of it
void stop_exact_store(ngsld_ctx *c);   // the matrix or its source changes, the context goes: the builder is cancelled and joined
// sites a launch over rows [r0, r1) reads: up to the furthest window end among its rows (a row a filter emptied ends at itself)
inline uint64_t exact_sites_needed(const ngsld_ctx *c, uint64_t r0, uint64_t r1) {
  uint64_t need = r1;
  for (uint64_t r = r0; r < r1 && r < c->h_row_end.size(); ++r) need = std::max<uint64_t>(need, c->h_row_end[r]);
  return std::min<uint64_t>(need, c->n_sites);
}
// should a run that has `pending` flagged pairs for the host build the store instead?
bool exact_store_wanted(const ngsld_ctx *c, uint64_t pending);
// flag_text: the launch's records become text (PairArgs::flag_text)
// start + wait for the launch's sites + device_replay_lkl; *applied says whether the replay was launched (false: no store to be had)
int try_device_replay_lkl(ngsld_ctx *c, uint32_t *d_flags, uint32_t cap, uint64_t out_base, uint64_t n, ngsld_rec_std *d_std,
                          ngsld_rec_ext *d_ext, hipStream_t st, bool flag_text, int slot, bool *applied, uint64_t need_sites);
// slot: the pipeline slot whose pair list the launch uses (-1: ngsld_run_device's)
int device_replay_lkl(ngsld_ctx *c, uint32_t *d_flags, uint32_t cap, uint64_t out_base, uint64_t n, ngsld_rec_std *d_std,
                      ngsld_rec_ext *d_ext, hipStream_t st, bool flag_text, int slot);
int replay_flagged(ngsld_ctx *c, const std::vector<uint64_t> &recs, uint64_t base, ngsld_rec_std *h_std,
                   ngsld_rec_ext *h_ext, ngsld_rec_std *d_std, ngsld_rec_ext *d_ext, hipStream_t st,
                   std::vector<uint32_t> *sites1 = nullptr, std::vector<uint32_t> *sites2 = nullptr);
int reset_flags(ngsld_ctx *c, DevBuf<uint32_t> &buf, uint64_t n, uint32_t cap, hipStream_t stream);
// the head of a flag buffer (counters + lists) into the pinned host buffer h_head, by a small kernel on `st`
// with_list == false: the launch had the device-side replay of likelihood matrices behind it (the listed pairs are settled)
int send_flag_head(ngsld_ctx *c, const uint32_t *d_flags, uint32_t *h_head, uint32_t cap, hipStream_t st, bool with_list = true);
int device_replay(ngsld_ctx *c, uint32_t *d_flags, uint32_t cap, uint64_t out_base, uint64_t n, ngsld_rec_std *d_std,
                  ngsld_rec_ext *d_ext, hipStream_t st, int slot);
int finish_device_run(ngsld_ctx *c);  // waits for a run left on a caller's stream and replays what it flagged

}  // namespace eng
}  // namespace ngsld
