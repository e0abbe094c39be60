// engine.hip -- host side of the C-ABI in include/ngsld.h: device state, the pair-space plan (the
// batched replacement of the reference's per-s1 thread-pool dispatch, ngsLD.cpp:153-198) and the
// batch pipeline kernel -> async D2H -> sink.
#include <hip/hip_runtime.h>
#include <sched.h>
#include <dlfcn.h>
#include <sys/mman.h>

#include <algorithm>
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
#include "ld_device.h"
#include "ld_prep.h"
#include "ld_replay.h"
#include "ld_text.h"
#include "replay.h"
#include "taus.h"

using namespace ngsld;

namespace {
thread_local std::string g_create_error;

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
Roctx &roctx() {
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
}  // namespace

struct ngsld_ctx {
  int device = 0;
  hipStream_t stream = nullptr, stream2 = nullptr, copy_stream = nullptr;
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
  DevBuf<int> d_all_hard;
  int h_all_hard = 0, h_prep_status = 0;
  uint32_t mask_words = 0;

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
  bool batch_pairs_set = false;  // by the caller (ngsld_set_tuning / NGSLD_BATCH_PAIRS): taken as it is

  // batch pipeline: three slots for record batches, two of them for text batches
  static constexpr int kSlots = 3;
  DevBuf<ngsld_rec_std> d_std[kSlots];
  DevBuf<ngsld_rec_ext> d_ext[kSlots];
  PinBuf<ngsld_rec_std> h_std[kSlots];
  PinBuf<ngsld_rec_ext> h_ext[kSlots];
  hipEvent_t ev_kernel_done[kSlots] = {nullptr, nullptr, nullptr}, ev_copy_done[kSlots] = {nullptr, nullptr, nullptr};
  // how record batches reach the host (ngsld_run without text output; NGSLD_RUN_DIRECT / NGSLD_RUN_TAPER):
  //   run_direct   the pair kernels write the records straight into the batch's pinned host buffers over the host link
  //                (72 B per pair at 2e8 pairs/s is 15 GB/s of posted writes; same-box A/B, profiles/r04/sink_ab.txt: the
  //                kernels take the same time) -- there is no device copy of the records and no D2H copy behind the last
  //                kernel.  Off: device buffers + a D2H copy per batch, and
  //   run_taper    the batches shrink towards the end of a run (a third of what is left, at least 2^19 pairs), so that the
  //                copy exposed behind the last kernel is small
  //   run_streams  1: one compute stream, every batch drains alone -- its last rows cut into short runs (build_runs), which
  //                takes the loss from 1.4 to ~0.4 ms per launch.  2 (opt-in, NGSLD_RUN_STREAMS=2): consecutive record
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
  uint32_t flag_cap[kSlots] = {0, 0, 0}, flag_cap_dev = 0;  // list entries of d_flags[k] / d_flags_dev as last reset
  bool replay_device = true;                       // called-genotype matrices: flagged pairs replayed by ld_replay.hip (NGSLD_REPLAY_DEVICE=0: host)
  uint64_t replayed_on_device = 0;
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
  } dev_run;                                  // the last ngsld_run_device, until ngsld_finish_device has looked at its flags

  // timing of pair-kernel launches
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  hipStream_t timed_stream = nullptr;
  uint64_t timed_pairs = 0;
};

namespace {

int finish_device_run(ngsld_ctx *c);  // (defined below: waits for a run left on a caller's stream and replays what it flagged)

int fail(ngsld_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  return code;
}

int hip_fail(ngsld_ctx *c, hipError_t e, const char *what) {
  return fail(c, e == hipErrorOutOfMemory ? NGSLD_ERR_NOMEM : NGSLD_ERR_DEVICE,
              std::string(what) + ": " + hipGetErrorString(e));
}

// The read-back stream of the exact-order replay (see ngsld_ctx::replay_stream); the copy stream where it cannot be had.
hipStream_t replay_stream_of(ngsld_ctx *c) {
  std::lock_guard<std::mutex> g(c->replay_stream_mu);
  if (c->replay_stream == nullptr && hipStreamCreateWithFlags(&c->replay_stream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    c->replay_stream = nullptr;
    return c->copy_stream;
  }
  return c->replay_stream;
}

#define HIP_TRY(c, call)                                \
  do {                                                  \
    hipError_t e_ = (call);                             \
    if (e_ != hipSuccess) return hip_fail(c, e_, #call); \
  } while (0)

// No exception crosses the C-ABI (include/ngsld.h): every entry point that allocates host memory is a function-try-block
// ending in this handler.  Work still in flight is waited for, so that buffers the caller owns are quiet on return.
int caught(ngsld_ctx *c, bool nomem) {
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
  c->have_geno = false;
  c->planned = false;
  c->text_mode = false;  // labels belong to a matrix
  c->replay_read = nullptr;  // and so does the replay source
  c->replay_user = nullptr;
  c->replay_matrix = nullptr;
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
    const char *pe = std::getenv("NGSLD_PREP_EXACT");  // tests: every triple through the reference's log / exp chain
    a.exact_chain = (pe != nullptr && std::strcmp(pe, "0") != 0) ? 1 : 0;
  }
  a.N_thresh = o.N_thresh;
  a.call_thresh = o.call_thresh;
  a.maf = c->d_maf.p;
  a.mean_e = c->d_mean.p;
  a.rsx = c->d_rsx.p;
  a.status = c->d_status.p;
  if (on_device) {
    a.raw = gl;
    a.maf_in = maf;
    a.site0 = 0;
    a.n_sites = n_sites;
    HIP_TRY(c, launch_prep(a, c->stream));
  } else {
    const uint64_t site_bytes = n_ind * 3 * sizeof(double);
    uint64_t stage_bytes = 64ull << 20;
    if (const char *e = std::getenv("NGSLD_STAGE_BYTES")) stage_bytes = std::strtoull(e, nullptr, 10);  // tests: force many chunks
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
  HIP_TRY(c, launch_pack_scalars(c->d_maf.p, c->d_mean.p, c->d_rsx.p, c->d_sc4.p, n_sites, c->stream));
  // Is every likelihood triple a called genotype or "no data" (text genotypes, --call_geno)?  Then the pairs run on the
  // 16 genotype-combination counts instead of the individuals (ld_pair_hard.hip).  NGSLD_HARD_KERNEL=0: never (A/B, tests).
  int &all_hard = c->h_all_hard;  // (ctx-owned: the asynchronous copies below must not target a stack frame an early return leaves)
  all_hard = 0;
  const char *hk = std::getenv("NGSLD_HARD_KERNEL");
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
  HIP_TRY(c, hipStreamSynchronize(c->stream));
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

PairArgs make_args(ngsld_ctx *c, uint64_t r0, uint64_t r1, ngsld_rec_std *d_std, ngsld_rec_ext *d_ext,
                   uint32_t *d_flags = nullptr, uint32_t flag_cap = 0) {
  PairArgs a{};
  a.flags = d_flags;
  a.flag_cap = flag_cap;
  a.flag_text = 1;
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

// Host threads this process may really run on: the affinity mask cut by the cgroup CPU quota (a lease that shows 256 CPUs
// and grants 16 is common); what the exact-order replay spreads its pairs over when nothing else was asked for.
unsigned usable_threads() {
  unsigned n = std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    const int k = CPU_COUNT(&set);
    if (k > 0) n = (unsigned)k;
  }
  if (FILE *fh = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char a[64] = "";
    double period = 0;
    if (std::fscanf(fh, "%63s %lf", a, &period) == 2 && std::strcmp(a, "max") != 0 && period > 0) {
      const double q = std::atof(a) / period;
      if (q >= 1.0 && q < (double)n) n = (unsigned)(q + 0.5);
    }
    std::fclose(fh);
  }
  return n ? n : 1u;
}

int check_status(ngsld_ctx *c) {
  int status = 0;
  HIP_TRY(c, hipMemcpy(&status, c->d_status.p, sizeof(int), hipMemcpyDeviceToHost));
  if (status == NGSLD_ERR_MAF_RANGE) return fail(c, NGSLD_ERR_MAF_RANGE, "invalid allele frequencies");
  return NGSLD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Exact-order replay (replay.h): the pairs the kernels flagged are re-evaluated on the host in the reference's own
// operation order and their records overwritten -- in the host buffers of a record batch, or on the device (text
// batches, ngsld_run_device) through a small scatter kernel.
// ---------------------------------------------------------------------------------------------------------------
__global__ void patch_records_kernel(const uint64_t *idx, uint64_t n, const ngsld_rec_std *src_std,
                                     const ngsld_rec_ext *src_ext, ngsld_rec_std *dst_std, ngsld_rec_ext *dst_ext) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  dst_std[idx[k]] = src_std[k];
  if (dst_ext != nullptr) dst_ext[idx[k]] = src_ext[k];
}

// (s1, s2) of plan records, on the device: what locate_record below does on the host copy of the items -- which a run that
// leaves its records on the device never needs otherwise (configs[3]: 7.8e7 items, 2.5 GB to copy and hold for a few
// hundred flagged pairs: 155 ms of its one 12 s step)
__global__ void locate_records_kernel(const uint64_t *rec, uint64_t n, uint64_t base, const uint64_t *row_off,
                                      const uint64_t *item_off, const Item *items, uint32_t n_sites, uint32_t *s1, uint32_t *s2) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  s1[t] = s2[t] = 0xffffffffu;
  const uint64_t r = base + rec[t];
  uint32_t lo = 0, hi = n_sites;  // largest row with row_off[row] <= r
  while (lo + 1 < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (row_off[mid] <= r) lo = mid; else hi = mid;
  }
  uint64_t il = item_off[lo], ih = item_off[lo + 1];
  if (il >= ih) return;
  while (il + 1 < ih) {
    const uint64_t mid = il + (ih - il) / 2;
    if (items[mid].first_record <= r) il = mid; else ih = mid;
  }
  const Item it = items[il];
  uint64_t k = r - it.first_record, m = it.mask;
  if (k >= (uint64_t)__popcll(m)) return;
  while (k--) m &= m - 1;
  s1[t] = it.s1;
  s2[t] = it.s2_begin + (uint32_t)(__ffsll((unsigned long long)m) - 1);
}

int ensure_host_items(ngsld_ctx *c) {  // the host copy of the plan's items, fetched on first use
  if (c->h_items.size() != c->n_items) {
    c->h_items.resize(c->n_items);
    if (c->n_items)
      HIP_TRY(c, hipMemcpy(c->h_items.data(), c->d_items.p, c->n_items * sizeof(Item), hipMemcpyDeviceToHost));
  }
  return NGSLD_OK;
}

// (s1, s2) of the plan's record `rec` (h_items must be present)
bool locate_record(const ngsld_ctx *c, uint64_t rec, uint32_t *s1, uint32_t *s2) {
  const auto &off = c->h_row_off;
  const uint64_t row = (uint64_t)(std::upper_bound(off.begin(), off.end(), rec) - off.begin()) - 1;
  if (row >= c->n_sites) return false;
  uint64_t lo = c->h_item_off[row], hi = c->h_item_off[row + 1];
  while (lo + 1 < hi) {  // last item of the row whose first record is <= rec
    const uint64_t mid = (lo + hi) / 2;
    if (c->h_items[mid].first_record <= rec) lo = mid; else hi = mid;
  }
  if (lo >= hi) return false;
  const Item &it = c->h_items[lo];
  uint64_t k = rec - it.first_record, m = it.mask;
  if (k >= (uint64_t)__builtin_popcountll(m)) return false;
  while (k--) m &= m - 1;  // drop the k lowest set bits
  *s1 = it.s1;
  *s2 = it.s2_begin + (uint32_t)__builtin_ctzll(m);
  return true;
}

// One site in the reference's arithmetic: from the caller's raw values when a source is registered, otherwise from the
// device's own planes (already normalised normal-space values; exact for ngsld_set_geno_lkl input).
int fetch_replay_site(ngsld_ctx *c, uint64_t s, std::vector<double> &tmp, ReplaySite *out) {
  const uint64_t n = c->n_ind;
  if (c->replay_matrix != nullptr) {  // the caller's own array, read in place
    const double *v = c->replay_matrix + s * 3 * n;
    if (c->normalised)
      replay_site_from_lkl(v, c->h_maf[s], n, out);
    else
      replay_site_from_raw(v, n, c->gopts, out);
    return NGSLD_OK;
  }
  if (c->replay_read != nullptr) {
    tmp.resize(3 * n);
    {
      std::lock_guard<std::mutex> g(c->replay_mu);
      if (c->replay_read(c->replay_user, s, 1, tmp.data()) != 0) return NGSLD_ERR_SINK;
    }
    if (c->normalised)
      replay_site_from_lkl(tmp.data(), c->h_maf[s], n, out);
    else
      replay_site_from_raw(tmp.data(), n, c->gopts, out);
    return NGSLD_OK;
  }
  // (pinned staging: a pageable copy would be staged by the runtime; while a pair kernel of the next batch has the device
  // this read-back can still wait for it -- callers that care register a source)
  tmp.resize(3 * n);
  double *lkl = tmp.data();
  {
    std::lock_guard<std::mutex> g(c->replay_mu);
    if (hipSetDevice(c->device) != hipSuccess || c->h_site_stage.resize(3ull * c->np) != hipSuccess) return NGSLD_ERR_DEVICE;
    hipStream_t rs = replay_stream_of(c);
    if (hipMemcpyAsync(c->h_site_stage.p, c->d_planes.p + s * 3ull * c->np, 3ull * c->np * sizeof(double),
                       hipMemcpyDeviceToHost, rs) != hipSuccess ||
        hipStreamSynchronize(rs) != hipSuccess)
      return NGSLD_ERR_DEVICE;
    const double *planes = c->h_site_stage.p;
    for (uint64_t i = 0; i < n; ++i)
      for (int g = 0; g < 3; ++g) lkl[3 * i + g] = planes[(uint64_t)g * c->np + i];
  }
  replay_site_from_lkl(lkl, c->h_maf[s], n, out);
  return NGSLD_OK;
}

// Rows per text batch (ngsld_run; a smaller NGSLD_BATCH_PAIRS / ngsld_set_tuning wins).  Round 4, configs[2] end to end on one
// box (profiles/r04/e2e_batch_size.txt): 2^21 1.42-1.46 s, 2^20 1.25-1.35 s, 2^19 1.23-1.25 s -- the loop itself takes the
// same 0.62 s whatever the count (a batch costs ~0.3 ms since its last rows go out as short runs and a replayed row no longer
// has every length derived again), while the two pinned buffers (2 x 400 MB at 2^21) cost 0.1 s to pin -- beside the matrix
// upload, which they slow -- and 0.06 s to give back.
constexpr uint64_t kTextBatchPairs = 1ull << 19;

// List entries of a launch of n records: a 256th of them (a called-genotype matrix flags one pair in ~4,000, a likelihood
// matrix one in 10^6), at least 4,096, at most 2^20 (8 MB of head to read back).
inline uint32_t flag_cap_for(uint64_t n) { return (uint32_t)std::min<uint64_t>(1ull << 20, std::max<uint64_t>(4096, n / 256)); }
inline size_t flag_head_bytes(uint32_t cap) { return (size_t)flag_head_words(cap) * sizeof(uint32_t); }
inline size_t flag_words(uint64_t n, uint32_t cap) { return (size_t)flag_head_words(cap) + (size_t)((n + 31) / 32); }

// The flagged records of a launch of n records, in increasing order.  h_head: the head of its flag buffer (counter + the
// first `cap` record indices) in host memory -- it travels with the batch, or is copied on the launch's own stream
// right behind the kernels (a copy issued later, while the next batch's pair kernel has the device, can wait for that
// kernel: measured 43 ms).  Only a launch that flagged more pairs than the list holds has its bitmap fetched from d_flags,
// on the replay stream (the kernels that set it are complete when this is called).
int flagged_records(ngsld_ctx *c, const uint32_t *h_head, const uint32_t *d_flags, uint32_t cap, uint64_t n,
                    std::vector<uint64_t> &recs) {
  recs.clear();
  const uint32_t count = h_head[0];
  if (count == 0) return NGSLD_OK;
  if (count <= cap) {
    const uint64_t *list = reinterpret_cast<const uint64_t *>(h_head + 2);
    recs.reserve(count);
    uint64_t on_device = 0;
    for (uint32_t k = 0; k < count; ++k) {
      if (list[k] & kFlagDone) {  // the device-side replay (ld_replay.hip) has rewritten this record already
        ++on_device;
        continue;
      }
      recs.push_back(list[k] & kFlagIndexMask);
    }
    c->replayed_on_device += on_device;
    c->replayed_pairs += on_device;
    std::sort(recs.begin(), recs.end());  // (the order the atomics landed in is not the record order)
    while (!recs.empty() && recs.back() >= n) recs.pop_back();
    return NGSLD_OK;
  }
  const size_t words = (size_t)((n + 31) / 32);
  HIP_TRY(c, c->h_flag_bits.resize(words ? words : 1));
  hipStream_t rs = replay_stream_of(c);
  HIP_TRY(c, hipMemcpyAsync(c->h_flag_bits.p, d_flags + flag_head_words(cap), words * sizeof(uint32_t), hipMemcpyDeviceToHost, rs));
  HIP_TRY(c, hipStreamSynchronize(rs));
  const uint32_t *bits = c->h_flag_bits.p;
  recs.reserve(count);
  for (uint64_t w = 0; w < words; ++w)
    for (uint32_t m = bits[w]; m; m &= m - 1) {
      const uint64_t r = w * 32 + (uint64_t)__builtin_ctz(m);
      if (r < n) recs.push_back(r);
    }
  return NGSLD_OK;
}

// Records `recs` (indices into a launch whose record 0 is the plan's record `base`, increasing) are replayed; the new
// records go to h_std / h_ext (host buffers of the batch) or, when those are null, to d_std / d_ext on stream st (synchronised).
int replay_flagged(ngsld_ctx *c, const std::vector<uint64_t> &recs, uint64_t base, ngsld_rec_std *h_std,
                   ngsld_rec_ext *h_ext, ngsld_rec_std *d_std, ngsld_rec_ext *d_ext, hipStream_t st,
                   std::vector<uint32_t> *sites1 = nullptr, std::vector<uint32_t> *sites2 = nullptr) {
  Range range_("ngsld:exact-order replay (host)");
  if (recs.empty()) return NGSLD_OK;
  // which pairs these records are: from the host copy of the plan's items where the run has one anyway (the sink path), from
  // the device's otherwise
  const bool have_items = c->h_items.size() == c->n_items;
  std::vector<uint32_t> loc_s1, loc_s2;
  if (!have_items) {
    hipStream_t ls = st != nullptr ? st : replay_stream_of(c);
    loc_s1.resize(recs.size());
    loc_s2.resize(recs.size());
    HIP_TRY(c, c->d_patch_idx.resize(recs.size()));
    HIP_TRY(c, c->d_patch_s1.resize(recs.size()));
    HIP_TRY(c, c->d_patch_s2.resize(recs.size()));
    HIP_TRY(c, hipMemcpyAsync(c->d_patch_idx.p, recs.data(), recs.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ls));
    hipLaunchKernelGGL(locate_records_kernel, dim3((unsigned)((recs.size() + 63) / 64)), dim3(64), 0, ls, c->d_patch_idx.p,
                       (uint64_t)recs.size(), base, c->d_row_off.p, c->d_item_off.p, c->d_items.p, (uint32_t)c->n_sites,
                       c->d_patch_s1.p, c->d_patch_s2.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(loc_s1.data(), c->d_patch_s1.p, recs.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ls));
    HIP_TRY(c, hipMemcpyAsync(loc_s2.data(), c->d_patch_s2.p, recs.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ls));
    HIP_TRY(c, hipStreamSynchronize(ls));
  }
  const bool ext = (h_std != nullptr ? (void *)h_ext : (void *)d_ext) != nullptr;
  const bool ign = c->params.ignore_miss_data != 0;
  std::vector<ngsld_rec_std> out_std(recs.size());
  std::vector<ngsld_rec_ext> out_ext(ext ? recs.size() : 0);
  if (sites1) sites1->assign(recs.size(), 0);  // (the pairs' sites, for callers that format the replayed rows again)
  if (sites2) sites2->assign(recs.size(), 0);
  int T = c->replay_threads > 0 ? c->replay_threads : (int)std::min<unsigned>(32u, usable_threads());
  if ((uint64_t)T > recs.size()) T = (int)recs.size();  // (a launch of 1e8 pairs flags a few dozen: sixteen per thread left them to two threads, 2.6 ms)
  std::vector<int> rcs((size_t)T, NGSLD_OK), stats((size_t)T, NGSLD_OK);
  std::vector<uint64_t> sites_done((size_t)T, 0);
  auto work = [&](int t) {
    const size_t k0 = recs.size() * (size_t)t / (size_t)T, k1 = recs.size() * (size_t)(t + 1) / (size_t)T;
    // records come in (s1, s2) order: the row's site is kept, the partners go through a bounded cache
    const size_t cache_cap = std::max<size_t>(64, (256ull << 20) / (32 * c->n_ind + 64));
    std::unordered_map<uint32_t, ReplaySite> cache;
    std::vector<double> tmp;
    ReplaySite row;
    uint32_t row_site = 0xffffffffu;
    try {
      for (size_t k = k0; k < k1; ++k) {
        uint32_t s1 = 0, s2 = 0;
        if (have_items ? !locate_record(c, base + recs[k], &s1, &s2)
                       : ((s1 = loc_s1[k]) == 0xffffffffu || (s2 = loc_s2[k]) == 0xffffffffu)) {
          rcs[(size_t)t] = NGSLD_ERR_INVALID;
          return;
        }
        if (sites1) (*sites1)[k] = s1;
        if (sites2) (*sites2)[k] = s2;
        if (s1 != row_site) {
          const int rc = fetch_replay_site(c, s1, tmp, &row);
          if (rc != NGSLD_OK) { rcs[(size_t)t] = rc; return; }
          row_site = s1;
          ++sites_done[(size_t)t];
        }
        auto hit = cache.find(s2);
        if (hit == cache.end()) {
          if (cache.size() >= cache_cap) cache.clear();
          hit = cache.emplace(s2, ReplaySite()).first;
          const int rc = fetch_replay_site(c, s2, tmp, &hit->second);
          if (rc != NGSLD_OK) { rcs[(size_t)t] = rc; return; }
          ++sites_done[(size_t)t];
        }
        replay_pair(row, hit->second, c->n_ind, ign, &out_std[k], ext ? &out_ext[k] : nullptr, &stats[(size_t)t]);
      }
    } catch (...) {
      rcs[(size_t)t] = NGSLD_ERR_NOMEM;
    }
  };
  if (T < 1) T = 1;
  c->replay_pool.run(T, work);
  for (int t = 0; t < T; ++t) {
    if (rcs[(size_t)t] != NGSLD_OK)
      return fail(c, rcs[(size_t)t], rcs[(size_t)t] == NGSLD_ERR_SINK ? "the replay source callback failed"
                                                                      : "exact-order replay failed");
    if (stats[(size_t)t] == NGSLD_ERR_MAF_RANGE) {
      const int v = NGSLD_ERR_MAF_RANGE;
      HIP_TRY(c, hipMemcpy(c->d_status.p, &v, sizeof(int), hipMemcpyHostToDevice));
    }
    c->replayed_sites += sites_done[(size_t)t];
  }
  c->replayed_pairs += recs.size();
  if (h_std != nullptr) {
    for (size_t k = 0; k < recs.size(); ++k) {
      h_std[recs[k]] = out_std[k];
      if (ext && h_ext != nullptr) h_ext[recs[k]] = out_ext[k];
    }
    return NGSLD_OK;
  }
  HIP_TRY(c, c->d_patch_idx.resize(recs.size()));
  HIP_TRY(c, c->d_patch_std.resize(recs.size()));
  if (ext) HIP_TRY(c, c->d_patch_ext.resize(recs.size()));
  HIP_TRY(c, hipMemcpyAsync(c->d_patch_idx.p, recs.data(), recs.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemcpyAsync(c->d_patch_std.p, out_std.data(), recs.size() * sizeof(ngsld_rec_std), hipMemcpyHostToDevice, st));
  if (ext)
    HIP_TRY(c, hipMemcpyAsync(c->d_patch_ext.p, out_ext.data(), recs.size() * sizeof(ngsld_rec_ext), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(patch_records_kernel, dim3((unsigned)((recs.size() + 255) / 256)), dim3(256), 0, st, c->d_patch_idx.p,
                     (uint64_t)recs.size(), c->d_patch_std.p, ext ? c->d_patch_ext.p : nullptr, d_std, ext ? d_ext : nullptr);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(st));  // the pageable source vectors go out of scope
  return NGSLD_OK;
}

// A flag buffer for n records with `cap` list entries, zeroed on `stream`.
int reset_flags(ngsld_ctx *c, DevBuf<uint32_t> &buf, uint64_t n, uint32_t cap, hipStream_t stream) {
  const size_t words = flag_words(n, cap), head = flag_head_words(cap);
  HIP_TRY(c, buf.resize(words));
  HIP_TRY(c, hipMemsetAsync(buf.p, 0, 2 * sizeof(uint32_t), stream));  // the counter (the list behind it needs no clearing)
  if (words > head)
    HIP_TRY(c, hipMemsetAsync(buf.p + head, 0, (words - head) * sizeof(uint32_t), stream));
  return NGSLD_OK;
}

// Called-genotype matrices: the flagged pairs of a launch replayed on the device (ld_replay.hip), right behind the pair
// kernels on their stream -- before the head of the flag buffer travels to the host, before text rows are formatted.
// out_base: plan index of the launch's record 0; d_std / d_ext: where the launch wrote (device, or pinned host memory).
int device_replay(ngsld_ctx *c, uint32_t *d_flags, uint32_t cap, uint64_t out_base, uint64_t n, ngsld_rec_std *d_std,
                  ngsld_rec_ext *d_ext, hipStream_t st) {
  if (!c->replay_on || !c->replay_device || c->cfg.kernel != kHard || d_flags == nullptr || n == 0) return NGSLD_OK;
  ReplayHardArgs a{};
  a.flags = d_flags;
  a.flag_cap = cap;
  a.row_off = c->d_row_off.p;
  a.item_off = c->d_item_off.p;
  a.items = c->d_items.p;
  a.n_sites = (uint32_t)c->n_sites;
  a.rec_base = out_base;
  a.masks = c->d_hard_masks.p;
  a.words = c->mask_words;
  a.n_ind = (uint32_t)c->n_ind;
  a.ignore_miss = c->params.ignore_miss_data;
  // "no data" individuals: only call_geno's triple is the same arithmetic on every individual (gen_func.cpp:903-905); a
  // matrix that came called from elsewhere may hold any three equal values -- its pairs at sites with missing data stay
  // with the host, which has the caller's raw values
  a.miss_ok = c->gopts.call_geno && !c->normalised ? 1 : 0;
  replay_missing_constants(&a.u_lkl, &a.u_pp);
  a.out_std = d_std;
  a.out_ext = d_ext;
  a.status = c->d_status.p;
  HIP_TRY(c, launch_replay_hard(a, n, st));
  return NGSLD_OK;
}

int finish_device_run(ngsld_ctx *c) {
  if (!c->dev_run.pending) return NGSLD_OK;
  c->dev_run.pending = false;
  const bool trace = std::getenv("NGSLD_TRACE") != nullptr;  // dev: where ngsld_finish_device's time goes, on stderr
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  HIP_TRY(c, hipStreamSynchronize(c->dev_run.st));  // (the head of the flag buffer came over behind the kernels, ngsld_run_device)
  const double t_sync = ms();
  if (!c->replay_on || c->d_flags_dev.p == nullptr || c->h_flags_dev.p == nullptr) return NGSLD_OK;
  if (c->h_flags_dev.p[0] == 0) return NGSLD_OK;
  const uint64_t base = c->h_row_off[c->dev_run.s1_begin], n = c->h_row_off[c->dev_run.s1_end] - base;
  std::vector<uint64_t> recs;
  const int rcf = flagged_records(c, c->h_flags_dev.p, c->d_flags_dev.p, c->flag_cap_dev, n, recs);
  if (rcf != NGSLD_OK) return rcf;
  const double t_list = ms();
  const int rcr = replay_flagged(c, recs, base, nullptr, nullptr, c->dev_run.d_std, c->dev_run.d_ext, c->dev_run.st);
  if (trace)
    std::fprintf(stderr, "[trace] finish_device: waited for the kernels %.2f ms, flag list %.2f ms (%u flagged, %zu for the host), "
                         "host replay + patch %.2f ms\n", t_sync, t_list - t_sync, c->h_flags_dev.p[0], recs.size(), ms() - t_list);
  return rcr;
}

}  // namespace

extern "C" {

const char *ngsld_version(void) { return "ngsld-amd 0.2.0 (gfx950; reference ngsLD 1.2.1)"; }

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
  if (const char *k = std::getenv("NGSLD_BATCH_PAIRS")) {  // tests: many small batches through ngsld_run
    const uint64_t v = std::strtoull(k, nullptr, 10);
    if (v > 0) {
      c->batch_pairs = v;
      c->batch_pairs_set = true;
    }
  }
  if (const char *k = std::getenv("NGSLD_REPLAY")) c->replay_on = std::strcmp(k, "0") != 0;  // A/B, tests
  if (const char *k = std::getenv("NGSLD_REPLAY_THREADS")) c->replay_threads = std::atoi(k);
  if (const char *k = std::getenv("NGSLD_REPLAY_DEVICE")) c->replay_device = std::strcmp(k, "0") != 0;
  if (const char *k = std::getenv("NGSLD_RUN_DIRECT")) c->run_direct = std::strcmp(k, "0") != 0;  // A/B, tests (see ngsld_ctx)
  if (const char *k = std::getenv("NGSLD_RUN_TAPER")) c->run_taper = std::strcmp(k, "0") != 0;
  if (const char *k = std::getenv("NGSLD_RUN_STREAMS")) c->run_streams = std::atoi(k) >= 2 ? 2 : 1;
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
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  c->d_planes.release(); c->d_maf.release(); c->d_mean.release(); c->d_rsx.release(); c->d_sc4.release(); c->d_runs.release();
  c->d_hard_masks.release(); c->d_hard_u.release(); c->d_all_hard.release();
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

// Runs: a row's items cut into ceil(items / run_len) runs of near-equal length, one workgroup each.  run_len = kRunItems
// (16 items = 1,024 candidates: a whole 100 kb row) is what the pair kernel likes best in one big launch; a run that goes
// out in SMALL batches -- text batches are 2^21 pairs, i.e. only four rounds of such workgroups on 512 slots, each batch
// ending in a ragged tail -- is cut finer (ngsld_run).  NGSLD_RUN_LEN overrides (tuning / A-B).
//
// Tails.  A launch of equal workgroups of length L ends in a drain: the device's 2 x CUs workgroup slots finish evenly over
// the last L (2.5 ms for whole-row runs at n_ind 500), i.e. L / 2 of the whole device is lost per launch -- 1.4 ms,
// measured: 12 launches of configs[2] take 500 ms, one launch 484 (profiles/r04/sink_ab.txt).  The rows at the END of every
// launch are therefore cut into short runs (run_len / 8): as many of them as fill that triangle (CUs x one full run of
// pairs), so that every slot that falls free during the drain still finds work and all of them end within one short
// workgroup of each other.  `launch_ends` = the rows (exclusive, increasing) at which the launches this list is for end.
// NGSLD_TAIL_LEN=0 turns the shaping off, NGSLD_TAIL_PAIRS / NGSLD_TAIL_LEN override its two numbers (A/B).
// (Also tried, round 4: the FIRST rows of a launch in runs of mixed lengths, so that the workgroups that start together do
// not turn over together for their first generations -- no gain, 0.9884 against 0.9894 of the device-resident rate,
// profiles/r04/sink_rr3.txt: dropped.)
static int build_runs(ngsld_ctx *c, uint64_t run_len, const std::vector<uint64_t> &launch_ends) {
  if (const char *e = getenv("NGSLD_RUN_LEN")) {
    const long v = atol(e);
    if (v >= 1) run_len = (uint64_t)v;
  }
  run_len = std::max<uint64_t>(1, std::min<uint64_t>(run_len, kRunItems));
  if (c->run_len == run_len && c->run_ends == launch_ends) return NGSLD_OK;
  const uint64_t n = c->n_sites;
  uint64_t tail_len = std::max<uint64_t>(1, run_len / 8);
  uint64_t tail_pairs = (uint64_t)c->n_cus * run_len * item_span(c->cfg, c->pairs_per_item);
  if (const char *e = getenv("NGSLD_TAIL_LEN")) tail_len = (uint64_t)std::max(0l, atol(e));
  if (const char *e = getenv("NGSLD_TAIL_PAIRS")) tail_pairs = std::strtoull(e, nullptr, 10);
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
      HIP_TRY(c, launch_pack_scalars(c->d_maf.p, c->d_mean.p, c->d_rsx.p, c->d_sc4.p, n, c->stream));
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

int ngsld_set_replay_source(ngsld_ctx *c, ngsld_read_sites_fn read, void *user) {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  if (!c->have_geno) return fail(c, NGSLD_ERR_INVALID, "set the genotype data before its replay source");
  c->replay_matrix = nullptr;
  c->replay_read = read;
  c->replay_user = user;
  c->planned = false;  // a --min_maf tie is settled at plan time
  return NGSLD_OK;
}

int ngsld_set_replay_matrix(ngsld_ctx *c, const double *values) {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  if (!c->have_geno) return fail(c, NGSLD_ERR_INVALID, "set the genotype data before its replay source");
  c->replay_matrix = values;
  c->replay_read = nullptr;
  c->replay_user = nullptr;
  c->planned = false;  // a --min_maf tie is settled at plan time
  return NGSLD_OK;
}

int ngsld_set_replay(ngsld_ctx *c, int enable) {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  c->replay_on = enable != 0;
  c->planned = false;
  return NGSLD_OK;
}

int ngsld_replay_stats(ngsld_ctx *c, uint64_t *pairs, uint64_t *sites) {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  if (pairs) *pairs = c->replayed_pairs;
  if (sites) *sites = c->replayed_sites;
  return NGSLD_OK;
}

int ngsld_finish_device(ngsld_ctx *c) try {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  (void)hipGetLastError();  // (a failure some earlier call already reported must not surface as a launch's "last error")
  const int rc = finish_device_run(c);
  if (rc != NGSLD_OK) return rc;
  return check_status(c);
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
                           c->flag_cap_dev);
    a.out_base = c->h_row_off[s1_begin];
    a.flag_text = 0;  // these records stay on the device: only numerically ill-conditioned pairs are replayed
    HIP_TRY(c, timed_launch(c, a, st));
    r0 = r1;
  }
  if (c->replay_on) {
    const int rcd = device_replay(c, c->d_flags_dev.p, c->flag_cap_dev, c->h_row_off[s1_begin], c->timed_pairs,
                                  (ngsld_rec_std *)d_std, (ngsld_rec_ext *)d_ext, st);
    if (rcd != NGSLD_OK) return rcd;
    // which pairs the kernels flagged (and the device has not settled itself): the counter and the list come over behind
    // them, on their stream
    HIP_TRY(c, hipMemcpyAsync(c->h_flags_dev.p, c->d_flags_dev.p, flag_head_bytes(c->flag_cap_dev), hipMemcpyDeviceToHost, st));
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
  const bool ext = c->params.extend_out != 0;
  c->ev_used = 0;
  c->timed_stream = c->stream;
  c->timed_pairs = c->h_row_off[s1_end] - c->h_row_off[s1_begin];
  c->replayed_pairs = 0;
  c->replayed_on_device = 0;
  const bool replay = c->replay_on;
  if (c->reserve_thread.joinable()) c->reserve_thread.join();  // (ngsld_reserve_text_buffers: h_text[] is this thread's again)

  // Device-side TSV: the dist column needs prefix sums of pos_dist that are EXACT (the host writer adds the gaps one
  // by one, ngsLD.cpp:241), i.e. integer gaps as read_dist produces them; otherwise the batches go out as records.
  bool text = c->text_mode;
  if (text) {
    const uint64_t n = c->n_sites;
    std::vector<double> cum(n);
    std::vector<uint32_t> infc(n);
    double run = 0.0;
    uint32_t ic = 0;
    for (uint64_t s = 0; s < n && text; ++s) {
      const double g = c->h_pos_dist[s];
      if (std::isinf(g) && g > 0) {
        ++ic;
      } else {
        if (!(g >= 0.0) || g != std::floor(g) || run + g > 9.0e15) text = false;
        run += g;
      }
      cum[s] = run;
      infc[s] = ic;
    }
    if (text) {
      HIP_TRY(c, c->d_cum.resize(n));
      HIP_TRY(c, c->d_infc.resize(n));
      HIP_TRY(c, hipMemcpy(c->d_cum.p, cum.data(), n * sizeof(double), hipMemcpyHostToDevice));
      HIP_TRY(c, hipMemcpy(c->d_infc.p, infc.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
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
  // NGSLD_RUN_STREAMS=2: three slots, two compute streams half a batch out of phase (see ngsld_ctx).
  const bool direct = !text && c->run_direct;
  // Text batches are small (2^19 rows: a 2.8 ms pair kernel, a tenth of it ramp and drain) and many: for them the two compute
  // streams half a batch out of phase DO pay, on every box -- while one stream's kernel drains the other's is in full
  // flight: configs[2]'s loop 0.58-0.63 -> 0.546-0.551 s (profiles/r04/e2e_text_streams.txt).  NGSLD_TEXT_STREAMS=1: one stream.
  bool text_two = true;
  if (const char *e = std::getenv("NGSLD_TEXT_STREAMS")) text_two = std::atoi(e) != 1;
  const bool two_streams = text ? text_two : c->run_streams == 2;
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
  uint64_t batch_pairs = text ? std::min<uint64_t>(c->batch_pairs, kTextBatchPairs)
                              : ((direct && !c->batch_pairs_set) ? 2 * c->batch_pairs : c->batch_pairs);
  const bool taper = !text && !direct && c->run_taper;
  uint64_t cap = 1;
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
    if (const char *lim = std::getenv("NGSLD_PIN_LIMIT_BYTES"))  // tests: a host that cannot pin more than this per buffer
      if (cap * sizeof(ngsld_rec_std) > std::strtoull(lim, nullptr, 10)) e = hipErrorOutOfMemory;
    for (int k = 0; k < S && e == hipSuccess; ++k) {
      e = c->h_std[k].resize(cap);
      if (e == hipSuccess && ext) e = c->h_ext[k].resize(cap);
    }
    if (e == hipSuccess) break;
    (void)hipGetLastError();
    if (cap <= (1ull << 16) || batches.size() >= (1u << 20)) return hip_fail(c, e, "pinned host buffers of a record batch");
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
      HIP_TRY(c, c->d_text_meta[k].resize(3));  // {total bytes, needs_host, a replayed row changed its length}
      HIP_TRY(c, c->h_text_meta[k].resize(3));
    }
    if (replay) {
      c->flag_cap[k] = flag_cap_for(cap);
      HIP_TRY(c, c->d_flags[k].resize(flag_words(cap, c->flag_cap[k])));
      HIP_TRY(c, c->h_flags[k].resize(flag_head_words(c->flag_cap[k])));
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
    PairArgs a = make_args(c, b.r0, b.r1, dev_std[k], dev_ext[k], replay ? c->d_flags[k].p : nullptr, c->flag_cap[k]);
    HIP_TRY(c, timed_launch(c, a, st));
    if (replay) {  // (called genotypes: the flagged pairs settled on the device, before anything reads the records)
      const int rcd = device_replay(c, c->d_flags[k].p, c->flag_cap[k], c->h_row_off[b.r0], b.n, dev_std[k], dev_ext[k], st);
      if (rcd != NGSLD_OK) return rcd;
    }
    // which pairs the kernel flagged for the exact-order replay (counter + list, 32 KB): known to the host with the batch.
    // On the kernel's own stream, right behind it: on the copy stream, behind the records, this small copy took 9 ms per
    // batch -- it goes through a copy kernel, and that waited for the next batch's pair kernel to leave it a CU
    if (replay)
      HIP_TRY(c, hipMemcpyAsync(c->h_flags[k].p, c->d_flags[k].p, flag_head_bytes(c->flag_cap[k]), hipMemcpyDeviceToHost, st));
    if (text) {  // row lengths and their prefix sums right behind the pair kernel; the rows are written at consume time
      HIP_TRY(c, hipMemsetAsync(c->d_text_meta[k].p, 0, 2 * sizeof(uint64_t), st));
      const TextArgs t = text_args(b, k);
      HIP_TRY(c, launch_text_lengths(t, st));
      HIP_TRY(c, text_scan(st == c->stream ? c->d_scan_tmp.p : c->d_scan_tmp_b.p, scan_bytes, c->d_lens[k].p, c->d_offs[k].p, b.n,
                           c->d_text_meta[k].p, st));
      HIP_TRY(c, hipMemcpyAsync(c->h_text_meta[k].p, c->d_text_meta[k].p, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(c, hipEventRecord(c->ev_kernel_done[k], st));
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
  int rc = NGSLD_OK;
  const bool trace = std::getenv("NGSLD_TRACE") != nullptr;  // dev: per-batch host timeline on stderr
  const auto t_run = std::chrono::steady_clock::now();
  auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run).count(); };
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
    if (text) {
      // the batch's text: its length is known now; the rows are written and copied on the copy stream while the pair
      // kernel of the next batch (already enqueued) runs on the compute stream
      HIP_TRY(c, hipEventSynchronize(c->ev_kernel_done[k]));
      if (replay && c->h_flags[k].p[0] != 0) {
        // flagged pairs: replayed on the host, patched into the device records, and the row lengths derived again --
        // all on the copy stream, beside the next batch's pair kernel
        int rcr = flagged_records(c, c->h_flags[k].p, c->d_flags[k].p, c->flag_cap[k], b.n, recs);
        if (rcr == NGSLD_OK)
          rcr = replay_flagged(c, recs, c->h_row_off[b.r0], nullptr, nullptr, c->d_std[k].p, ext ? c->d_ext[k].p : nullptr,
                               c->copy_stream, &rep_s1, &rep_s2);
        if (rcr != NGSLD_OK) return rcr;
        // Only the replayed rows' lengths are derived again (replay_flagged left their record indices in d_patch_idx); the
        // prefix sums are taken again only if one of them changed -- a full length pass + scan beside the next batch's pair
        // kernel cost that kernel ~1 ms of every 11 (profiles/r04/e2e_timeline.txt)
        if (!recs.empty()) {
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
      }
      const uint64_t total = c->h_text_meta[k].p[0];
      bool needs_host = (c->h_text_meta[k].p[1] & 0xffffffffull) != 0;
      if (const char *e = std::getenv("NGSLD_TEXT_FALLBACK_EVERY")) {  // tests: every n-th batch takes the record path
        const uint64_t every = std::strtoull(e, nullptr, 10);
        if (every > 0 && bi % every == every - 1) needs_host = true;
      }
      if (needs_host) {
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
        if (total > c->d_text[k].n) HIP_TRY(c, c->d_text[k].resize(total + total / 8));
        if (total > c->h_text[k].n) HIP_TRY(c, c->h_text[k].resize(total + total / 8));
        if (total) {
          const TextArgs t = text_args(b, k);
          HIP_TRY(c, launch_text_write(t, c->copy_stream));
          HIP_TRY(c, hipMemcpyAsync(c->h_text[k].p, c->d_text[k].p, total, hipMemcpyDeviceToHost, c->copy_stream));
        }
        HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
        out.text = c->h_text[k].p;
        out.text_len = total;
      }
    } else {
      HIP_TRY(c, hipEventSynchronize(direct ? c->ev_kernel_done[k] : c->ev_copy_done[k]));
      if (trace) std::fprintf(stderr, "[trace] batch %zu (%llu pairs): issue next %.2f..%.2f, records on the host %.2f, flagged %u\n", bi, (unsigned long long)b.n, t_a, t_b, now_ms(), replay ? c->h_flags[k].p[0] : 0u);
      if (replay && c->h_flags[k].p[0] != 0) {  // flagged pairs: replayed on the host, patched into the batch's buffers
        int rcr = flagged_records(c, c->h_flags[k].p, c->d_flags[k].p, c->flag_cap[k], b.n, recs);
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
    if (trace) std::fprintf(stderr, "[trace] batch %zu: replay done %.2f\n", bi, now_ms());
    Range range_sink("ngsld:sink");
    if (sink(user, &out) != 0) rc = fail(c, NGSLD_ERR_SINK, "sink callback failed");
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream2));
  HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
  if (rc != NGSLD_OK) return rc;
  return check_status(c);
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

uint64_t ngsld_slab_sites_for_budget(uint64_t n_ind, uint64_t budget_bytes) {
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
  return (per_ctx - fixed) / (24ull * cfg.np + 64ull);
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
