// multi.hip -- one job over several GPUs of one node, inside ONE process: the replacement of the reference's thread-pool
// section (ngsLD.cpp:153-198: one calc_pair_LD job per s1 dealt to --n_threads workers) at the scale of devices.
//
// Every SNP pair is independent given the read-only GL matrix (the reference already exploits this per s1), so the rows
// are cut into as many contiguous parts as there are devices, balanced by candidate-pair count, and each part runs on
// its own device from its own host thread through the public C-ABI (include/ngsld.h): create -> set_geno -> plan ->
// run.  Part k holds only the sites its rows pair with -- [row_begin, site_end): its rows plus the window halo, or
// everything from row_begin on for an all-pairs run -- and never talks to the other parts while computing.  Per-site
// quantities (est_maf, expected-genotype moments) do not depend on the part and --rnd_sample row seeds are taken at the
// row's GLOBAL index (ngsld_params.first_row), so the parts' records concatenate to the single-device run bit for bit.
//
// Distribution of the matrix:
//   * windowed runs, or no RCCL: every part uploads its own slab from host memory over its own PCIe link -- no collective;
//   * all-pairs runs on >= 2 distinct devices: the matrix goes to the first device once and ONE ncclBroadcast (RCCL over
//     xGMI; librccl is loaded on demand, it is not a link-time dependency of this library) hands it to the others.
//     NGSLD_TEST_MULTI_DIST=upload / broadcast overrides the choice.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the five entry points used are resolved with dlsym

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "knobs.h"
#include <thread>
#include <vector>

#include "../../include/ngsld.h"
#include "host_buf.h"

#include <atomic>

namespace {

// how the last ngsld_run_multi of this process distributed the matrix (ngsld_multi_last_distribution)
std::atomic<int> g_last_distribution{NGSLD_DIST_NONE};

void set_err(char *err, size_t errlen, const std::string &msg) {
  if (err != nullptr && errlen > 0) std::snprintf(err, errlen, "%s", msg.c_str());
}

struct Barrier {  // all parts meet here (C++17 has no std::barrier)
  std::mutex mu;
  std::condition_variable cv;
  int n, waiting = 0, phase = 0;
  explicit Barrier(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    const int ph = phase;
    if (++waiting == n) {
      waiting = 0;
      ++phase;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return phase != ph; });
    }
  }
};

struct PartSink {  // same records, site indices moved from the part's frame to the global one
  uint64_t base;
  int part;
  ngsld_multi_sink_fn sink;
  void *user;
  std::vector<ngsld_item> items;
};
int part_sink(void *user, const ngsld_batch *b) {
  PartSink *r = static_cast<PartSink *>(user);
  ngsld_batch g = *b;
  if (b->items != nullptr) {
    r->items.assign(b->items, b->items + b->n_items);
    for (ngsld_item &it : r->items) {
      it.s1 += (uint32_t)r->base;
      it.s2_begin += (uint32_t)r->base;
    }
    g.items = r->items.data();
  }
  g.s1_begin += r->base;
  g.s1_end += r->base;
  return r->sink(r->user, r->part, &g);
}

// The five RCCL entry points, resolved on demand.
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool load() {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib != nullptr) break;
    }
    if (lib == nullptr) return false;
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(lib, "ncclBroadcast"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast && GetErrorString;
  }
};

// The whole raw matrix on every device: H2D to devices[0], one ncclBroadcast to the others.  d_raw[k] are allocated here
// and owned by the caller.  Returns an empty string on success, the reason otherwise (the caller falls back to uploads).
std::string broadcast_matrix(const int *devices, int n, const double *gl_raw, size_t bytes, std::vector<void *> &d_raw,
                             bool *used_rccl) {
  *used_rccl = false;
  d_raw.assign((size_t)n, nullptr);
  std::vector<hipStream_t> st((size_t)n, nullptr);
  std::string why;
  auto cleanup = [&]() {
    for (int k = 0; k < n; ++k)
      if (st[(size_t)k]) {
        (void)hipSetDevice(devices[k]);
        (void)hipStreamDestroy(st[(size_t)k]);
      }
  };
  for (int k = 0; k < n && why.empty(); ++k) {
    hipError_t e = hipSetDevice(devices[k]);
    if (e == hipSuccess) e = hipMalloc(&d_raw[(size_t)k], bytes);
    if (e == hipSuccess) e = hipStreamCreate(&st[(size_t)k]);
    if (e != hipSuccess) why = std::string("device buffer for the broadcast: ") + hipGetErrorString(e);
  }
  if (why.empty()) {
    hipError_t e = hipSetDevice(devices[0]);
    if (e == hipSuccess) e = hipMemcpy(d_raw[0], gl_raw, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) why = std::string("upload to the first device: ") + hipGetErrorString(e);
  }
  bool distinct = true;
  for (int a = 0; a < n; ++a)
    for (int b = a + 1; b < n; ++b) distinct = distinct && devices[a] != devices[b];
  Rccl r;
  if (why.empty() && distinct && r.load()) {
    std::vector<ncclComm_t> comm((size_t)n, nullptr);
    ncclResult_t rc = r.CommInitAll(comm.data(), n, devices);
    if (rc == ncclSuccess) {
      rc = r.GroupStart();
      for (int k = 0; k < n && rc == ncclSuccess; ++k) {
        (void)hipSetDevice(devices[k]);
        rc = r.Broadcast(d_raw[(size_t)k], d_raw[(size_t)k], bytes, ncclChar, 0, comm[(size_t)k], st[(size_t)k]);  // in place; the send side counts at the root only
      }
      const ncclResult_t rc2 = r.GroupEnd();
      if (rc == ncclSuccess) rc = rc2;
      for (int k = 0; k < n; ++k) {
        (void)hipSetDevice(devices[k]);
        if (hipStreamSynchronize(st[(size_t)k]) != hipSuccess && rc == ncclSuccess) rc = ncclUnhandledCudaError;
      }
      for (int k = 0; k < n; ++k)
        if (comm[(size_t)k]) (void)r.CommDestroy(comm[(size_t)k]);
    }
    if (rc == ncclSuccess) {
      *used_rccl = true;
      cleanup();
      return why;
    }
    std::fprintf(stderr, "ngsld: RCCL broadcast failed (%s), copying device to device instead\n", r.GetErrorString(rc));
  }
  if (why.empty()) {  // no RCCL (or the same device more than once): plain device-to-device copies from the first
    for (int k = 1; k < n && why.empty(); ++k) {
      hipError_t e = hipSetDevice(devices[k]);
      if (e == hipSuccess)
        e = devices[k] == devices[0] ? hipMemcpy(d_raw[(size_t)k], d_raw[0], bytes, hipMemcpyDeviceToDevice)
                                     : hipMemcpyPeer(d_raw[(size_t)k], devices[k], d_raw[0], devices[0], bytes);
      if (e != hipSuccess) why = std::string("device to device copy: ") + hipGetErrorString(e);
    }
  }
  cleanup();
  return why;
}

}  // namespace

extern "C" {

int ngsld_plan_parts(const double *pos_dist, uint64_t n_sites, const ngsld_params *params, int n_parts, ngsld_slab *parts) {
  if (params == nullptr || parts == nullptr || n_sites == 0 || n_parts < 1) return NGSLD_ERR_INVALID;
  std::vector<uint32_t> row_end;
  try {
    row_end.resize(n_sites);
  } catch (const std::bad_alloc &) {
    return NGSLD_ERR_NOMEM;
  }
  const int rc = ngsld_window_ends(pos_dist, n_sites, params, row_end.data());
  if (rc != NGSLD_OK) return rc;
  // candidate pairs before each row (the maf / sub-sampling filters thin all rows alike: balanced by candidates)
  std::vector<uint64_t> cum(n_sites + 1, 0);
  for (uint64_t s = 0; s < n_sites; ++s) cum[s + 1] = cum[s] + (row_end[s] > s + 1 ? row_end[s] - (s + 1) : 0);
  const uint64_t total = cum[n_sites];
  uint64_t lo = 0;
  for (int k = 0; k < n_parts; ++k) {
    uint64_t hi = n_sites;
    if (k + 1 < n_parts) {
      const long double want = (long double)total * (long double)(k + 1) / (long double)n_parts;
      hi = (uint64_t)(std::lower_bound(cum.begin(), cum.end(), (uint64_t)want) - cum.begin());
      hi = std::min<uint64_t>(std::max<uint64_t>(hi, lo), n_sites);
    }
    uint64_t site_end = hi;
    for (uint64_t s = lo; s < hi; ++s) site_end = std::max<uint64_t>(site_end, row_end[s]);
    parts[k] = ngsld_slab{lo, hi, site_end};
    lo = hi;
  }
  return NGSLD_OK;
}

int ngsld_run_multi(const int *devices, int n_devices, uint64_t n_sites, uint64_t n_ind, const double *pos_dist,
                    const ngsld_params *params, const ngsld_geno_opts *opts, const double *gl_raw,
                    ngsld_read_sites_fn read, void *read_user, double *maf_out, ngsld_multi_sink_fn sink, void *sink_user,
                    const char *const *labels, int text_output, uint64_t *pairs_per_part, char *err, size_t errlen) try {
  if (devices == nullptr || n_devices < 1 || params == nullptr || opts == nullptr || sink == nullptr || n_sites == 0 ||
      n_ind == 0 || (gl_raw == nullptr && read == nullptr)) {
    set_err(err, errlen, "invalid argument");
    return NGSLD_ERR_INVALID;
  }
  if (opts->on_device) {
    set_err(err, errlen, "a multi-device run reads host memory");
    return NGSLD_ERR_INVALID;
  }
  const int n = n_devices;
  {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
      (void)hipGetLastError();
      set_err(err, errlen, "part 0 (device " + std::to_string(devices[0]) + "): no HIP device available; this library has no CPU fallback");
      return NGSLD_ERR_DEVICE;
    }
    for (int k = 0; k < n; ++k)
      if (devices[k] < 0 || devices[k] >= n_dev) {
        set_err(err, errlen, "part " + std::to_string(k) + " (device " + std::to_string(devices[k]) + "): device index out of range");
        return NGSLD_ERR_INVALID;
      }
  }
  std::vector<ngsld_slab> parts((size_t)n);
  int rc = ngsld_plan_parts(pos_dist, n_sites, params, n, parts.data());
  if (rc != NGSLD_OK) {
    set_err(err, errlen, "cannot cut the rows into parts");
    return rc;
  }
  if (pairs_per_part)
    for (int k = 0; k < n; ++k) pairs_per_part[k] = 0;

  // ---- all-pairs runs: the matrix once to the first device, one broadcast to the others ----
  const bool all_pairs = params->max_kb_dist == 0 && params->max_snp_dist == 0;
  const char *dist = ngsld::test_knob("MULTI_DIST");
  bool broadcast = gl_raw != nullptr && n > 1 && all_pairs;
  if (dist != nullptr && std::strcmp(dist, "upload") == 0) broadcast = false;
  if (dist != nullptr && std::strcmp(dist, "broadcast") == 0) broadcast = gl_raw != nullptr && n > 1;
  std::vector<void *> d_raw;
  bool used_rccl = false;
  if (broadcast) {
    const std::string why = broadcast_matrix(devices, n, gl_raw, (size_t)n_sites * n_ind * 3 * sizeof(double), d_raw, &used_rccl);
    if (!why.empty()) {  // e.g. not enough device memory for the whole matrix everywhere: every part uploads its own slab
      for (int k = 0; k < n; ++k)
        if (k < (int)d_raw.size() && d_raw[(size_t)k]) {
          (void)hipSetDevice(devices[k]);
          (void)hipFree(d_raw[(size_t)k]);
        }
      d_raw.clear();
      broadcast = false;
    }
    (void)hipGetLastError();  // a handled failure must not surface later as some launch's "last error"
  }
  g_last_distribution.store(broadcast ? (used_rccl ? NGSLD_DIST_RCCL : NGSLD_DIST_PEER_COPY) : NGSLD_DIST_UPLOAD);
  if (const char *v = std::getenv("NGSLD_MULTI_VERBOSE"))
    if (std::strcmp(v, "0") != 0)
      std::fprintf(stderr, "ngsld_run_multi: %d parts, matrix %s\n", n,
                   broadcast ? (used_rccl ? "broadcast over RCCL" : "copied device to device") : "uploaded slab by slab");

  Barrier barrier(n);
  std::vector<int> rcs((size_t)n, NGSLD_OK), hard((size_t)n, 0);
  std::vector<std::string> msgs((size_t)n);
  std::vector<ngsld::HostMatrix> host((size_t)n);  // (host_buf.h: a part's raw values, no zero-fill, huge pages)

  auto work = [&](int k) {
    const ngsld_slab &pt = parts[(size_t)k];
    const uint64_t m = pt.site_end - pt.row_begin, rows = pt.row_end - pt.row_begin;
    ngsld_ctx *ctx = nullptr;
    int r = NGSLD_OK, passed = 0;  // passed: barriers behind this part (an exception must not leave the others waiting)
    std::string msg;
    auto failed = [&](const char *what) {
      msg = std::string(what) + ": " + (ctx ? ngsld_last_error(ctx) : ngsld_last_error(nullptr));
    };
    try {
      r = ngsld_create(devices[k], &ctx);
      if (r != NGSLD_OK) failed("ngsld_create");
      const double *slab = nullptr;  // the part's raw values in host memory (replay source)
      ngsld_geno_opts so = *opts;
      if (r == NGSLD_OK && rows > 0) {
        if (gl_raw != nullptr) {
          slab = gl_raw + pt.row_begin * n_ind * 3;
        } else {
          if (!host[(size_t)k].alloc((size_t)(m * n_ind * 3))) throw std::bad_alloc();
          if (read(read_user, pt.row_begin, m, host[(size_t)k].data()) != 0) {
            r = NGSLD_ERR_INVALID;
            msg = "cannot read the genotype data of a part";
          }
          slab = host[(size_t)k].data();
        }
      }
      auto set_geno = [&]() {
        if (broadcast) {
          so.on_device = 1;
          return ngsld_set_geno_raw_opts(ctx, static_cast<const double *>(d_raw[(size_t)k]) + pt.row_begin * n_ind * 3, m, n_ind, &so);
        }
        return ngsld_set_geno_raw_opts(ctx, slab, m, n_ind, &so);
      };
      if (r == NGSLD_OK && rows > 0) {
        r = set_geno();
        if (r != NGSLD_OK) failed("ngsld_set_geno_raw_opts");
      }
      // One kernel family for the whole job: the genotype-combination kernel runs only if EVERY part's slab qualifies
      hard[(size_t)k] = (r == NGSLD_OK && rows > 0) ? (std::strcmp(ngsld_pair_kernel(ctx), "hard") == 0 ? 1 : 0) : -1;
      rcs[(size_t)k] = r;
      barrier.wait();
      ++passed;
      bool any_soft = false, any_failed = false;
      for (int q = 0; q < n; ++q) {
        any_soft = any_soft || hard[(size_t)q] == 0;
        any_failed = any_failed || rcs[(size_t)q] != NGSLD_OK;
      }
      if (!any_failed && rows > 0 && any_soft && hard[(size_t)k] == 1) {
        so.per_individual_only = 1;
        r = set_geno();
        if (r != NGSLD_OK) failed("ngsld_set_geno_raw_opts");
      }
      barrier.wait();  // nobody reads the broadcast buffers any more
      ++passed;
      if (broadcast && d_raw[(size_t)k]) {
        (void)hipSetDevice(devices[k]);
        (void)hipFree(d_raw[(size_t)k]);
        d_raw[(size_t)k] = nullptr;
      }
      if (!any_failed && r == NGSLD_OK && rows > 0) {
        r = ngsld_set_replay_matrix(ctx, slab);  // (the part's slab stays in host memory for the run: read in place)
        if (r == NGSLD_OK) r = ngsld_set_pos_dist(ctx, pos_dist ? pos_dist + pt.row_begin : nullptr);
        ngsld_params p = *params;
        p.first_row = params->first_row + pt.row_begin;
        uint64_t all_rows = 0;  // includes the halo rows, which the next part computes
        if (r == NGSLD_OK) r = ngsld_plan(ctx, &p, &all_rows);
        const uint64_t *row_off = nullptr;
        if (r == NGSLD_OK) r = ngsld_plan_rows(ctx, &row_off, nullptr);
        if (r == NGSLD_OK && pairs_per_part) pairs_per_part[k] = row_off[rows];
        if (r == NGSLD_OK && maf_out != nullptr) {
          // the part's own sites; entries of the halo are also written by the next parts, with the same values
          std::vector<double> maf(m);
          r = ngsld_get_maf(ctx, maf.data());
          if (r == NGSLD_OK) std::memcpy(maf_out + pt.row_begin, maf.data(), m * sizeof(double));
        }
        if (r == NGSLD_OK && text_output) r = ngsld_set_text_output(ctx, labels ? labels + pt.row_begin : nullptr, 1);
        if (r != NGSLD_OK) failed("plan");
      }
      rcs[(size_t)k] = r;
      barrier.wait();  // every maf entry a sink may read is final
      ++passed;
      bool ok_all = true;
      for (int q = 0; q < n; ++q) ok_all = ok_all && rcs[(size_t)q] == NGSLD_OK;
      if (ok_all && rows > 0) {
        PartSink ps{pt.row_begin, k, sink, sink_user, {}};
        r = ngsld_run(ctx, 0, rows, part_sink, &ps);
        if (r != NGSLD_OK) failed("ngsld_run");
      }
    } catch (...) {
      r = NGSLD_ERR_NOMEM;
      msg = "out of host memory in a part";
      rcs[(size_t)k] = r;
      for (; passed < 3; ++passed) barrier.wait();  // the other parts may be waiting for this one
    }
    rcs[(size_t)k] = r;
    msgs[(size_t)k] = msg;
    if (ctx) ngsld_destroy(ctx);
  };

  std::vector<std::thread> th;
  for (int k = 1; k < n; ++k) th.emplace_back(work, k);
  work(0);
  for (auto &t : th) t.join();
  for (int k = 0; k < n; ++k)
    if (k < (int)d_raw.size() && d_raw[(size_t)k]) {
      (void)hipSetDevice(devices[k]);
      (void)hipFree(d_raw[(size_t)k]);
    }
  for (int k = 0; k < n; ++k)
    if (rcs[(size_t)k] != NGSLD_OK) {
      set_err(err, errlen, "part " + std::to_string(k) + " (device " + std::to_string(devices[k]) + "): " + msgs[(size_t)k]);
      return rcs[(size_t)k];
    }
  return NGSLD_OK;
} catch (const std::bad_alloc &) {
  set_err(err, errlen, "out of host memory");
  return NGSLD_ERR_NOMEM;
}

int ngsld_multi_last_distribution(void) { return g_last_distribution.load(); }

int ngsld_rccl_selftest(int device, uint64_t bytes, char *err, size_t errlen) try {
  if (bytes < 8 || bytes > (1ull << 32)) {
    set_err(err, errlen, "bytes out of range");
    return NGSLD_ERR_INVALID;
  }
  const size_t n = (size_t)(bytes / 8);
  std::vector<double> host(n), back(n);
  for (size_t i = 0; i < n; ++i) host[i] = (double)(i % 1021) * 0.5 + 0.25;
  std::vector<void *> d_raw;
  bool used_rccl = false;
  const int devices[1] = {device};
  const std::string why = broadcast_matrix(devices, 1, host.data(), n * sizeof(double), d_raw, &used_rccl);
  int rc = NGSLD_OK;
  if (!why.empty()) {
    set_err(err, errlen, why);
    rc = NGSLD_ERR_DEVICE;
  } else if (!used_rccl) {
    set_err(err, errlen, "librccl could not be loaded, or one of its calls failed (see stderr)");
    rc = NGSLD_ERR_DEVICE;
  } else if (hipSetDevice(device) != hipSuccess ||
             hipMemcpy(back.data(), d_raw[0], n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
             std::memcmp(back.data(), host.data(), n * sizeof(double)) != 0) {
    set_err(err, errlen, "the buffer came back changed from the in-place broadcast");
    rc = NGSLD_ERR_DEVICE;
  }
  if (!d_raw.empty() && d_raw[0] != nullptr) {
    (void)hipSetDevice(device);
    (void)hipFree(d_raw[0]);
  }
  (void)hipGetLastError();
  return rc;
} catch (const std::bad_alloc &) {
  set_err(err, errlen, "out of host memory");
  return NGSLD_ERR_NOMEM;
}

}  // extern "C"
