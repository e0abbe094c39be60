// engine_replay.hip -- exact-order replay (replay.h): the pairs the kernels flagged are re-evaluated in the reference's own
// operation order and their records overwritten -- in the host buffers of a record batch, or on the device (text batches,
// ngsld_run_device) through a small scatter kernel.
#include "engine.h"

namespace ngsld {
namespace eng {

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
__device__ inline void locate_record_on_device(uint64_t r, const uint64_t *row_off, const uint64_t *item_off, const Item *items,
                                               uint32_t n_sites, uint32_t *s1, uint32_t *s2) {
  *s1 = *s2 = 0xffffffffu;
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
  *s1 = it.s1;
  *s2 = it.s2_begin + (uint32_t)(__ffsll((unsigned long long)m) - 1);
}

__global__ void locate_records_kernel(const uint64_t *rec, uint64_t n, uint64_t base, const uint64_t *row_off,
                                      const uint64_t *item_off, const Item *items, uint32_t n_sites, uint32_t *s1, uint32_t *s2) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  locate_record_on_device(base + rec[t], row_off, item_off, items, n_sites, &s1[t], &s2[t]);
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

// The flagged records of a launch of n records, in increasing order.  h_head: the head of its flag buffer (counter + the
// first `cap` record indices) in host memory -- it travels with the batch, or is copied on the launch's own stream
// right behind the kernels (a copy issued later, while the next batch's pair kernel has the device, can wait for that
// kernel: measured 43 ms).  Only a launch that flagged more pairs than the list holds has its bitmap fetched from d_flags,
// on the replay stream (the kernels that set it are complete when this is called).
int flagged_records(ngsld_ctx *c, const uint32_t *h_head, const uint32_t *d_flags, uint32_t cap, uint64_t n,
                    std::vector<uint64_t> &recs, bool dev_applied) {
  recs.clear();
  const uint32_t count = h_head[0];
  if (count == 0) return NGSLD_OK;
  const size_t words = flag_bitmap_words(n);
  auto from_bitmap = [&](size_t first_word, uint32_t expect) -> int {
    HIP_TRY(c, c->h_flag_bits.resize(words ? words : 1));
    hipStream_t rs = replay_stream_of(c);
    HIP_TRY(c, hipMemcpyAsync(c->h_flag_bits.p, d_flags + first_word, words * sizeof(uint32_t), hipMemcpyDeviceToHost, rs));
    HIP_TRY(c, hipStreamSynchronize(rs));
    const uint32_t *bits = c->h_flag_bits.p;
    recs.reserve(expect);
    for (uint64_t w = 0; w < words; ++w)
      for (uint32_t m = bits[w]; m; m &= m - 1) {
        const uint64_t r = w * 32 + (uint64_t)__builtin_ctz(m);
        if (r < n) recs.push_back(r);
      }
    return NGSLD_OK;
  };
  if (h_head[7] != 0) dev_applied = true;  // (called genotypes: the launch overflowed its list and the device took all of it, ld_replay.hip)
  if (dev_applied) {
    // likelihood matrices, device-side replay behind the launch (ld_replay_lkl.hip): it settled every flagged pair but the
    // host-only ones -- head[2] says how many --, and those have a bitmap of their own
    c->replayed_on_device += h_head[2];
    c->replayed_pairs += h_head[2];
    const uint32_t host_only = h_head[1];
    if (host_only == 0) return NGSLD_OK;
    if (host_only <= kFlagHostCap) {  // (their own list names them)
      const uint64_t *list = reinterpret_cast<const uint64_t *>(h_head + kFlagListAt + 2u * cap);
      recs.assign(list, list + host_only);
      std::sort(recs.begin(), recs.end());
      while (!recs.empty() && recs.back() >= n) recs.pop_back();
      return NGSLD_OK;
    }
    return from_bitmap((size_t)flag_head_words(cap) + words, host_only);
  }
  if (count <= cap) {
    const uint64_t *list = reinterpret_cast<const uint64_t *>(h_head + kFlagListAt);
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
  return from_bitmap(flag_head_words(cap), count);
}

// The pairs (s1[k], s2[k]), k < n, in the reference's operation order on the host's threads.
int replay_pairs_on_host(ngsld_ctx *c, const uint32_t *s1v, const uint32_t *s2v, size_t n, ngsld_rec_std *out_std, ngsld_rec_ext *out_ext) {
  if (n == 0) return NGSLD_OK;
  const bool ign = c->params.ignore_miss_data != 0;
  int T = c->replay_threads > 0 ? c->replay_threads : (int)std::min<unsigned>(32u, usable_threads());
  if ((uint64_t)T > n) T = (int)n;  // (a launch of 1e8 pairs flags a few dozen: sixteen per thread left them to two threads, 2.6 ms)
  if (T < 1) T = 1;
  std::vector<int> rcs((size_t)T, NGSLD_OK), stats((size_t)T, NGSLD_OK);
  std::vector<uint64_t> sites_done((size_t)T, 0);
  auto work = [&](int t) {
    const size_t k0 = n * (size_t)t / (size_t)T, k1 = n * (size_t)(t + 1) / (size_t)T;
    // records come in (s1, s2) order: the row's site is kept, the partners go through a bounded cache
    const size_t cache_cap = std::max<size_t>(64, (256ull << 20) / (32 * c->n_ind + 64));
    std::unordered_map<uint32_t, ReplaySite> cache;
    std::vector<double> tmp;
    ReplaySite row;
    uint32_t row_site = 0xffffffffu;
    try {
      for (size_t k = k0; k < k1; ++k) {
        const uint32_t s1 = s1v[k], s2 = s2v[k];
        if (s1 >= c->n_sites || s2 >= c->n_sites) {
          rcs[(size_t)t] = NGSLD_ERR_INVALID;
          return;
        }
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
        replay_pair(row, hit->second, c->n_ind, ign, &out_std[k], out_ext ? &out_ext[k] : nullptr, &stats[(size_t)t]);
      }
    } catch (...) {
      rcs[(size_t)t] = NGSLD_ERR_NOMEM;
    }
  };
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
  return NGSLD_OK;
}

// Records `recs` (indices into a launch whose record 0 is the plan's record `base`, increasing) are replayed; the new
// records go to h_std / h_ext (host buffers of the batch) or, when those are null, to d_std / d_ext on stream st (synchronised).
int replay_flagged(ngsld_ctx *c, const std::vector<uint64_t> &recs, uint64_t base, ngsld_rec_std *h_std,
                   ngsld_rec_ext *h_ext, ngsld_rec_std *d_std, ngsld_rec_ext *d_ext, hipStream_t st,
                   std::vector<uint32_t> *sites1, std::vector<uint32_t> *sites2) {
  Range range_("ngsld:exact-order replay (host)");
  if (recs.empty()) return NGSLD_OK;
  // which pairs these records are: from the host copy of the plan's items where the run has one anyway (the sink path), from
  // the device's otherwise
  const bool have_items = c->h_items.size() == c->n_items;
  std::vector<uint32_t> loc_s1(recs.size()), loc_s2(recs.size());
  if (!have_items) {
    hipStream_t ls = st != nullptr ? st : replay_stream_of(c);
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
  } else {
    for (size_t k = 0; k < recs.size(); ++k)
      if (!locate_record(c, base + recs[k], &loc_s1[k], &loc_s2[k])) loc_s1[k] = loc_s2[k] = 0xffffffffu;
  }
  for (size_t k = 0; k < recs.size(); ++k)
    if (loc_s1[k] == 0xffffffffu || loc_s2[k] == 0xffffffffu) return fail(c, NGSLD_ERR_INVALID, "exact-order replay failed");
  const bool ext = (h_std != nullptr ? (void *)h_ext : (void *)d_ext) != nullptr;
  std::vector<ngsld_rec_std> out_std(recs.size());
  std::vector<ngsld_rec_ext> out_ext(ext ? recs.size() : 0);
  const int rcp = replay_pairs_on_host(c, loc_s1.data(), loc_s2.data(), recs.size(), out_std.data(), ext ? out_ext.data() : nullptr);
  if (rcp != NGSLD_OK) return rcp;
  if (sites1) *sites1 = loc_s1;  // (the pairs' sites, for callers that format the replayed rows again)
  if (sites2) *sites2 = loc_s2;
  c->replayed_pairs += recs.size();
  c->host_replayed_total += recs.size();
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

// Text batches: which pair each record the batch leaves to the host is, and where its row lies in the batch's text -- written
// into pinned host memory right behind the batch's row lengths and prefix sums, on the pair kernel's stream.  With it the host
// settles such a batch without submitting anything to the device (engine_run.hip): the kernels that located the pairs, patched
// the records, took the rows' lengths again and wrote the text again stood -- in a hardware queue shared with the compute
// streams -- behind the pair kernels of the next two batches, and the loop ran dry behind them (GPU_MAX_HW_QUEUES=1 / 2:
// 0.67-0.69 s for configs[2]'s loop against 0.556, profiles/r05/hw_queues_ab.txt).
__global__ void flag_rows_to_host_kernel(const uint32_t *flags, uint32_t cap, int dev_applied, const uint64_t *offs, const uint64_t *lens,
                                         uint64_t n_records, uint64_t base, const uint64_t *row_off, const uint64_t *item_off, const Item *items,
                                         uint32_t n_sites, FlagRow *rows) {
  // (flags[7]: called genotypes, the launch overflowed its list and the device took all of it -- flagged_records)
  const bool host_list = dev_applied != 0 || flags[7] != 0;
  const uint32_t n = host_list ? flags[1] : flags[0];
  if (n > kFlagRowsCap || n > (host_list ? kFlagHostCap : cap)) return;
  const uint64_t *list = reinterpret_cast<const uint64_t *>(flags + kFlagListAt + (host_list ? 2u * cap : 0u));
  for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
    FlagRow r{};
    r.rec = list[j] & kFlagIndexMask;
    r.s1 = r.s2 = 0xffffffffu;
    if (r.rec < n_records) {
      r.off = offs[r.rec];
      r.len = (uint32_t)lens[r.rec];
      locate_record_on_device(base + r.rec, row_off, item_off, items, n_sites, &r.s1, &r.s2);
    }
    rows[j] = r;
  }
}

int send_flag_rows(ngsld_ctx *c, const uint32_t *d_flags, uint32_t cap, bool dev_applied, const uint64_t *d_offs, const uint64_t *d_lens,
                   uint64_t n_records, uint64_t base, FlagRow *h_rows, hipStream_t st) {
  FlagRow *dev_view = nullptr;
  HIP_TRY(c, hipHostGetDevicePointer((void **)&dev_view, h_rows, 0));
  hipLaunchKernelGGL(flag_rows_to_host_kernel, dim3(1), dim3(64), 0, st, d_flags, cap, dev_applied ? 1 : 0, d_offs, d_lens, n_records, base,
                     c->d_row_off.p, c->d_item_off.p, c->d_items.p, (uint32_t)c->n_sites, dev_view);
  HIP_TRY(c, hipGetLastError());
  return NGSLD_OK;
}

// The head of a launch's flag buffer -- the counters, the listed pairs, the host-only pairs -- written into the batch's pinned
// host buffer by a one-workgroup kernel on the launch's own stream.  NOT hipMemcpyAsync: a copy of this size goes to an SDMA
// queue, where it waits for the kernels before it on its stream -- and everything queued behind it on that engine waits with
// it, whatever stream it came from: the 86 MB D2H copy of the previous batch's text (copy stream) did not start until the pair
// kernel AND the device-side replay of the next batch had finished (rocprofv3 --memory-copy-trace, profiles/r05): 8.7 ms a
// text batch instead of 7.3 on un-called input, 2.9 instead of 2.5 on the headline's.
__global__ void flag_head_to_host_kernel(const uint32_t *flags, uint32_t *h_head, uint32_t cap, int with_list) {
  const uint32_t count = flags[0], host_only = flags[1];
  if (threadIdx.x < kFlagListAt) h_head[threadIdx.x] = flags[threadIdx.x];
  // (with_list == 0: a device-side replay has been through the launch -- the host reads the counters and the host-only list)
  const uint32_t n_list = with_list ? 2u * (count < cap ? count : cap) : 0u, n_host = 2u * (host_only < kFlagHostCap ? host_only : kFlagHostCap);
  for (uint32_t w = threadIdx.x; w < n_list; w += blockDim.x) h_head[kFlagListAt + w] = flags[kFlagListAt + w];
  const uint32_t at = kFlagListAt + 2u * cap;
  for (uint32_t w = threadIdx.x; w < n_host; w += blockDim.x) h_head[at + w] = flags[at + w];
}

int send_flag_head(ngsld_ctx *c, const uint32_t *d_flags, uint32_t *h_head, uint32_t cap, hipStream_t st, bool with_list) {
  uint32_t *dev_view = nullptr;
  HIP_TRY(c, hipHostGetDevicePointer((void **)&dev_view, h_head, 0));
  hipLaunchKernelGGL(flag_head_to_host_kernel, dim3(1), dim3(256), 0, st, d_flags, dev_view, cap, with_list ? 1 : 0);
  HIP_TRY(c, hipGetLastError());
  return NGSLD_OK;
}

// A flag buffer for n records with `cap` list entries, zeroed on `stream`.
int reset_flags(ngsld_ctx *c, DevBuf<uint32_t> &buf, uint64_t n, uint32_t cap, hipStream_t stream) {
  const size_t words = flag_words(n, cap), head = flag_head_words(cap);
  HIP_TRY(c, buf.resize(words));
  HIP_TRY(c, hipMemsetAsync(buf.p, 0, kFlagListAt * sizeof(uint32_t), stream));  // the counters (the list behind them needs no clearing)
  if (words > head)
    HIP_TRY(c, hipMemsetAsync(buf.p + head, 0, (words - head) * sizeof(uint32_t), stream));  // both bitmaps
  return NGSLD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Likelihood matrices: the exact store and the device-side replay (ld_replay_lkl.hip)
// ---------------------------------------------------------------------------------------------------------------

bool lkl_device_eligible(const ngsld_ctx *c) {
  // (beyond 4,096 individuals the wavefront-per-pair kernel has no shape: the lanes take such cohorts -- where their copy of the
  // store cannot be had, start_exact_store gives the matrix up to the host)
  return c->replay_on && c->replay_device && c->exact_mode != 0 && c->have_geno && c->cfg.kernel != kHard;
}

// the planes ARE the store: the caller's own normal-space values (ngsld_set_geno_lkl), or no source to build another from
bool exact_store_is_free(const ngsld_ctx *c) {
  return c->normalised || (c->replay_matrix == nullptr && c->replay_read == nullptr);
}

// Host replay costs ~0.15 us per individual and pair on one thread, the store ~0.25 us per individual and SITE (17 libm calls
// per triple: read_geno's log + post_prob, est_maf's post_prob + exp, main's exp), once: it pays as soon as the host would
// otherwise replay more pairs than half the matrix has sites.
bool exact_store_wanted(const ngsld_ctx *c, uint64_t pending) {
  if (!lkl_device_eligible(c) || c->exact_failed) return false;
  if (exact_store_started(c) || exact_store_is_free(c)) return pending > 0;
  if (c->exact_mode >= 2) return pending > 0;
  return c->host_replayed_total + pending > std::max<uint64_t>(4096, c->n_sites / 2);
}

// The individual-major copy for the lane-per-pair kernel, where the device has room for the matrix once more: its memory
static bool alloc_lane_store(ngsld_ctx *c) {
  c->xT_ready = false;
  const size_t elems = (size_t)c->n_sites * c->n_ind * 3;
  // (4 GB of the device left for what comes later -- text buffers, the lanes' list and sort scratch; under a cap: half a GB of it)
  if (!room_for(elems * sizeof(double), 4ull << 30, 512ull << 20)) return false;
  // The order of the sites in the copy: the RARE ones first (folded frequency below 1/32, or no frequency), the others behind,
  // each class in site order.  The pairs the lanes replay are pairs with a (nearly) monomorphic site, a wavefront's 64 lanes hold
  // a few neighbouring rare sites x 8 partners they share (replay_keys_kernel), and per individual the rare sites' triples are
  // 24 bytes each out of a cache line of their own while they stand among 32 consecutive sites: the launch fetched 1.3 TB
  // through an L2 that hit 18 % of the time (profiles/r06/replay_lane).  Standing together they share their lines: same-box
  // 612.5 / 613.3 -> 601.2 / 602.5 ms a pass at 20 % monomorphic sites (profiles/r06/lane/perm_ab.txt).
  std::vector<uint32_t> perm(c->n_sites);
  {
    uint32_t n_rare = 0;
    auto rare = [&](uint64_t s) {
      const double m = c->h_maf[s], r = m <= 0.5 ? m : 1 - m;
      return !(r >= 0.03125);
    };
    for (uint64_t s = 0; s < c->n_sites; ++s) n_rare += rare(s) ? 1u : 0u;
    uint32_t at_rare = 0, at_rest = n_rare;
    for (uint64_t s = 0; s < c->n_sites; ++s) perm[s] = rare(s) ? at_rare++ : at_rest++;
  }
  if (c->d_xT.resize(elems) != hipSuccess || c->d_xdepth.resize(c->n_sites) != hipSuccess ||
      hipMemset(c->d_xdepth.p, 0, c->n_sites * sizeof(uint32_t)) != hipSuccess ||
      c->d_xperm.resize(c->n_sites) != hipSuccess ||
      hipMemcpy(c->d_xperm.p, perm.data(), c->n_sites * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    c->d_xT.release();
    c->d_xperm.release();
    return false;
  }
  return true;
}

// ... and all of it at once from planes that are a store already (the alias forms), on stream st, synchronised
static hipError_t build_lane_store(ngsld_ctx *c, hipStream_t st) {
  if (!alloc_lane_store(c)) return hipSuccess;
  hipError_t e = launch_transpose_store(c->exact_alias ? c->d_planes.p : c->d_xplanes.p, 3ull * c->np, c->np, (uint32_t)c->n_ind, c->n_sites,
                                        c->d_xT.p, c->d_xdepth.p, c->d_xperm.p, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  c->xT_ready = true;
  return hipSuccess;
}

// a chunk of the store from pinned host memory into place (site_elems is even: np is a multiple of 8)
__global__ __launch_bounds__(256) void store_chunk_kernel(const double *__restrict__ src, double *__restrict__ dst, uint64_t n,
                                                          const double *__restrict__ src_maf, double *__restrict__ dst_maf, uint64_t n_maf) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, T = (uint64_t)gridDim.x * blockDim.x;
  const double2 *s2 = reinterpret_cast<const double2 *>(src);
  double2 *d2 = reinterpret_cast<double2 *>(dst);
  for (uint64_t i = t; i < n / 2; i += T) d2[i] = s2[i];
  if (t == 0 && (n & 1)) dst[n - 1] = src[n - 1];
  for (uint64_t i = t; i < n_maf; i += T) dst_maf[i] = src_maf[i];
}

// The builder (ngsld_ctx::exact_thread): the registered source through the host's libm, chunk by chunk in site order -- two
// pinned staging buffers, a chunk's upload (and its transposition into the lane kernel's copy) beside the next chunk's
// arithmetic --, the frontier published as the uploads land.  Touches nothing of the context a run touches but the store's own buffers and atomics.
static void exact_builder(ngsld_ctx *c) {
  Range range_("ngsld:exact store (host libm -> device)");
  const auto t0 = std::chrono::steady_clock::now();
  int rc = NGSLD_OK;
  std::string msg;
  auto hip_bad = [&](hipError_t e, const char *what) {
    if (e == hipSuccess) return false;
    (void)hipGetLastError();
    rc = NGSLD_ERR_DEVICE;
    msg = std::string(what) + ": " + hipGetErrorString(e);
    return true;
  };
  try {
    const uint64_t n = c->n_sites, ni = c->n_ind, np = c->np, site_elems = 3 * np;
    uint64_t chunk = std::max<uint64_t>(1, (32ull << 20) / (site_elems * sizeof(double)));
    uint64_t slow_us = 0;  // tests: small chunks, a builder the run has to wait for
    if (const char *e = test_knob("EXACT_CHUNK_SITES")) chunk = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));
    if (const char *e = test_knob("EXACT_SLOW_US")) slow_us = std::strtoull(e, nullptr, 10);
    if (chunk > n) chunk = n;
    struct Events {  // (destroyed on every way out)
      hipEvent_t e[2] = {nullptr, nullptr};
      ~Events() {
        for (hipEvent_t x : e)
          if (x) (void)hipEventDestroy(x);
      }
    } up;
    bool ok = !hip_bad(hipSetDevice(c->device), "exact store");
    if (ok && c->exact_stream == nullptr) {
      // a HIGH-PRIORITY stream: the runtime gives it a hardware queue of that priority instead of a share in one of the run's
      // four -- where a chunk's upload stood behind an 80 ms replay kernel of the run, which itself waited for the builder's
      // frontier: one chunk per batch, 120,000 x 2,000 built in 5.1 s instead of 0.4
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
        (void)hipGetLastError();
        least = greatest = 0;
      }
      ok = !hip_bad(hipStreamCreateWithPriority(&c->exact_stream, hipStreamNonBlocking, greatest), "exact store stream");
    }
    for (int k = 0; k < 2 && ok; ++k) {
      ok = !hip_bad(c->h_xstage[k].resize((size_t)chunk * (site_elems + 1)), "exact store staging") &&  // (+ 1: the site's est_maf behind the planes)
           !hip_bad(hipEventCreateWithFlags(&up.e[k], hipEventDisableTiming), "exact store event");
    }
    hipStream_t st = c->exact_stream;
    const int T = c->replay_threads > 0 ? c->replay_threads : (int)std::min<unsigned>(32u, usable_threads());
    std::vector<double> raw_chunk;  // (callback source: the chunk's raw values, fetched with one call)
    uint64_t k = 0, prev_end = 0;
    for (uint64_t s0 = 0; s0 < n && ok && !c->exact_cancel.load(); s0 += chunk, ++k) {
      const int b = (int)(k & 1);
      const uint64_t m = std::min(chunk, n - s0);
      // (buffer b was last read by the upload of chunk k - 2, whose event was waited for when chunk k - 1 was issued)
      const double *raw = nullptr;
      if (c->replay_matrix != nullptr) {
        raw = c->replay_matrix + s0 * 3 * ni;
      } else {
        raw_chunk.resize((size_t)m * 3 * ni);
        std::lock_guard<std::mutex> g(c->replay_mu);
        if (c->replay_read(c->replay_user, s0, m, raw_chunk.data()) != 0) {
          rc = NGSLD_ERR_SINK;
          msg = "the replay source callback failed";
          ok = false;
          break;
        }
        raw = raw_chunk.data();
      }
      if (slow_us) std::this_thread::sleep_for(std::chrono::microseconds(slow_us));
      double *stage = c->h_xstage[b].p, *stage_maf = stage + (size_t)chunk * site_elems;
      const int Tm = (int)std::min<uint64_t>((uint64_t)T, m);
      std::vector<int> good((size_t)Tm, 1);
      c->exact_pool.run(Tm, [&](int t) {
        try {
          for (uint64_t s = m * (uint64_t)t / (uint64_t)Tm; s < m * (uint64_t)(t + 1) / (uint64_t)Tm; ++s)
            replay_site_planes(raw + s * 3 * ni, ni, c->gopts, np, stage + s * site_elems, &stage_maf[s]);
        } catch (...) {
          good[(size_t)t] = 0;
        }
      });
      for (int t = 0; t < Tm; ++t)
        if (!good[(size_t)t]) {
          rc = NGSLD_ERR_NOMEM;
          msg = "out of host memory";
          ok = false;
        }
      if (!ok) break;
      // (a KERNEL reads the pinned chunk over the host link, not hipMemcpyAsync: an SDMA copy can queue behind the run's text
      // copies on that engine, each of which waits for its batch's kernels -- send_flag_head)
      double *stage_dev = nullptr;
      if (hip_bad(hipHostGetDevicePointer((void **)&stage_dev, stage, 0), "exact store upload")) {
        ok = false;
        break;
      }
      hipLaunchKernelGGL(store_chunk_kernel, dim3(512), dim3(256), 0, st, stage_dev, c->d_xplanes.p + s0 * site_elems, (uint64_t)m * site_elems,
                         stage_dev + (size_t)chunk * site_elems, c->d_xmaf.p + s0, (uint64_t)m);
      bool sent = !hip_bad(hipGetLastError(), "exact store upload");
      if (sent && c->xT_ready.load())  // (the lane kernel's copy of these sites, behind their planes on the same stream)
        sent = !hip_bad(launch_transpose_store(c->d_xplanes.p, site_elems, (uint32_t)np, (uint32_t)ni, n, c->d_xT.p, c->d_xdepth.p, c->d_xperm.p, st, s0, s0 + m),
                        "exact store, individual-major copy");
      if (!sent || hip_bad(hipEventRecord(up.e[b], st), "exact store upload")) {
        ok = false;
        break;
      }
      if (k >= 1) {  // the chunk before this one has landed (its upload ran beside this chunk's arithmetic): publish it
        if (hip_bad(hipEventSynchronize(up.e[b ^ 1]), "exact store upload")) {
          ok = false;
          break;
        }
        c->exact_frontier.store(prev_end);
        c->exact_cv.notify_all();
      }
      prev_end = s0 + m;
    }
    if (c->exact_stream != nullptr) {
      const hipError_t e = hipStreamSynchronize(c->exact_stream);
      if (ok) ok = !hip_bad(e, "exact store upload");
    }
    if (ok && !c->exact_cancel.load()) {
      c->exact_frontier.store(n);
      c->exact_cv.notify_all();
    }
    c->h_xstage[0].release();  // (64 MB of pinned host memory, used once per matrix)
    c->h_xstage[1].release();
  } catch (...) {
    rc = NGSLD_ERR_NOMEM;
    msg = "out of host memory";
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  {
    std::lock_guard<std::mutex> g(c->exact_mu);
    c->exact_rc = rc;
    c->exact_msg = msg;
    c->exact_build_s = secs;
    c->exact_state.store(rc == NGSLD_OK && !c->exact_cancel.load() ? 2 : -1);
  }
  c->exact_cv.notify_all();
  if (std::getenv("NGSLD_TRACE") != nullptr)
    std::fprintf(stderr, "[trace] exact store: %llu sites x %llu individuals through the host's libm in %.3f s (a thread of its own beside the run)\n",
                 (unsigned long long)c->n_sites, (unsigned long long)c->n_ind, secs);
}

void stop_exact_store(ngsld_ctx *c) {
  if (c->exact_thread.joinable()) {
    c->exact_cancel.store(true);
    c->exact_thread.join();
  }
  c->exact_cancel.store(false);
  c->exact_state.store(0);
  c->exact_frontier.store(0);
  c->exact_rc = NGSLD_OK;
  c->exact_msg.clear();
  c->exact_ready = false;
  c->xT_ready = false;
  // (a new source or matrix: nothing of the old store is of use, start_exact_store allocates what it needs)
  c->d_xplanes.release();
  c->d_xmaf.release();
  c->d_xT.release();
  c->d_xperm.release();
  c->d_xdepth.release();
}

int start_exact_store(ngsld_ctx *c) {
  if (c->exact_state.load() != 0 || c->exact_failed) return NGSLD_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  if (exact_store_is_free(c)) {  // the planes ARE the store
    c->exact_alias = true;
    c->exact_frontier.store(c->n_sites);
    const hipError_t e = build_lane_store(c, replay_stream_of(c));
    if (e != hipSuccess) return hip_fail(c, e, "exact store, individual-major copy");
    if (!c->xT_ready.load() && replay_lkl_waves((uint32_t)c->n_ind) == 0) {  // (no room for the copy the lanes read, and no other kernel for this cohort)
      c->exact_failed = true;
      return NGSLD_OK;
    }
    c->exact_state.store(2);
    c->exact_ready = true;
    return NGSLD_OK;
  }
  // (a device without room for the matrix once more: no store for this matrix -- its flagged pairs stay with the host's threads)
  const bool pretend = test_knob("EXACT_STORE_NO_ROOM") != nullptr;
  const bool no_room = !room_for((uint64_t)c->n_sites * (3ull * c->np + 1) * sizeof(double), 0, 256ull << 20);  // (ngsld_set_memory_budget)
  if (pretend || no_room || c->d_xplanes.resize((size_t)c->n_sites * 3 * c->np) != hipSuccess || c->d_xmaf.resize(c->n_sites) != hipSuccess) {
    (void)hipGetLastError();
    c->d_xplanes.release();
    c->d_xmaf.release();
    c->exact_failed = true;
    return NGSLD_OK;
  }
  if (c->exact_thread.joinable()) c->exact_thread.join();  // (a builder that ended by itself)
  c->exact_alias = false;
  c->xT_ready = alloc_lane_store(c);  // (filled chunk by chunk behind the planes: usable as far as the frontier, like them)
  if (!c->xT_ready.load() && replay_lkl_waves((uint32_t)c->n_ind) == 0) {  // (see the alias form above)
    c->d_xplanes.release();
    c->d_xmaf.release();
    c->exact_failed = true;
    return NGSLD_OK;
  }
  c->exact_cancel.store(false);
  c->exact_frontier.store(0);
  c->exact_state.store(1);
  try {
    c->exact_thread = std::thread(exact_builder, c);
  } catch (...) {
    c->exact_state.store(0);
    c->exact_failed = true;  // (no thread to be had: host replay)
  }
  return NGSLD_OK;
}

int wait_exact_store(ngsld_ctx *c, uint64_t need_sites, bool *have) {
  *have = false;
  if (c->exact_failed || c->exact_state.load() == 0) return NGSLD_OK;
  const bool all = need_sites >= c->n_sites;
  {
    std::unique_lock<std::mutex> lk(c->exact_mu);
    // (the builder publishes the frontier without the lock: a short timed wait instead of a missed wake-up)
    while (c->exact_state.load() == 1 && (all || c->exact_frontier.load() < need_sites)) c->exact_cv.wait_for(lk, std::chrono::milliseconds(1));
    if (c->exact_state.load() == -1) {
      const int rc = c->exact_rc;
      const std::string msg = c->exact_msg;
      lk.unlock();
      if (c->exact_thread.joinable()) c->exact_thread.join();
      c->exact_state.store(0);
      c->exact_failed = true;  // (not again for this matrix)
      c->d_xplanes.release();
      c->d_xmaf.release();
      c->d_xT.release();
      c->d_xperm.release();
      c->d_xdepth.release();
      c->xT_ready = false;
      return rc != NGSLD_OK ? fail(c, rc, msg.c_str()) : NGSLD_OK;
    }
  }
  if (c->exact_state.load() == 2) {
    if (c->exact_thread.joinable()) c->exact_thread.join();
    c->exact_ready = true;
  }
  *have = true;
  return NGSLD_OK;
}

int ensure_exact_store(ngsld_ctx *c) {
  if (c->exact_ready) return NGSLD_OK;
  const int rc = start_exact_store(c);
  if (rc != NGSLD_OK) return rc;
  bool have = false;
  return wait_exact_store(c, c->n_sites, &have);
}

int try_device_replay_lkl(ngsld_ctx *c, uint32_t *d_flags, uint32_t cap, uint64_t out_base, uint64_t n, ngsld_rec_std *d_std,
                          ngsld_rec_ext *d_ext, hipStream_t st, bool flag_text, int slot, bool *applied, uint64_t need_sites) {
  *applied = false;
  int rc = start_exact_store(c);
  if (rc != NGSLD_OK) return rc;
  bool have = false;
  rc = wait_exact_store(c, need_sites, &have);
  if (rc != NGSLD_OK || !have) return rc;
  rc = device_replay_lkl(c, d_flags, cap, out_base, n, d_std, d_ext, st, flag_text, slot);
  if (rc == NGSLD_OK) *applied = true;
  return rc;
}

// The flagged pairs of a launch replayed on the device (likelihood matrices), on `st` behind the pair kernels that flagged
// them -- before the head of the flag buffer travels to the host, before text rows are formatted.  The store must be ready.
int device_replay_lkl(ngsld_ctx *c, uint32_t *d_flags, uint32_t cap, uint64_t out_base, uint64_t n, ngsld_rec_std *d_std,
                      ngsld_rec_ext *d_ext, hipStream_t st, bool flag_text, int slot) {
  if (!exact_store_started(c) || d_flags == nullptr || n == 0) return NGSLD_OK;
  ReplayLklArgs a{};
  const size_t head = flag_head_words(cap), words = flag_bitmap_words(n);
  a.bits = d_flags + head;
  a.host_bits = d_flags + head + words;
  a.flags = d_flags;
  a.flag_cap = cap;
  a.flag_text = flag_text ? 1u : 0u;
  a.mean_e = c->d_mean.p;
  a.rsx = c->d_rsx.p;
  a.n_records = n;
  a.done = d_flags + 2;
  a.work = d_flags + 3;
  a.row_off = c->d_row_off.p;
  a.item_off = c->d_item_off.p;
  a.items = c->d_items.p;
  a.n_items = c->n_items;
  a.n_sites = (uint32_t)c->n_sites;
  a.rec_base = out_base;
  a.xplanes = c->exact_alias ? c->d_planes.p : c->d_xplanes.p;
  a.site_stride = 3ull * c->np;
  a.np = c->np;
  a.xmaf = c->exact_alias ? c->d_maf.p : c->d_xmaf.p;
  a.n_ind = (uint32_t)c->n_ind;
  a.ignore_miss = c->params.ignore_miss_data;
  a.out_std = d_std;
  a.out_ext = d_ext;
  a.status = c->d_status.p;
  a.xt_sites = c->n_sites;
  a.xdepth = c->d_xdepth.p;
  a.xperm = c->d_xperm.p;
  // (a launch of a few hundred thousand records -- a text batch -- stays with the wavefront-per-pair kernel: a lane takes
  // milliseconds over ONE pair, four wavefronts to a SIMD, and such a launch has no second pair for most lanes: 6 ms a batch
  // against 4, profiles/r05/e2e_uncalled_lanes_on_text_batches.json)
  uint64_t lanes_from = 1ull << 22;
  // ... but cohorts beyond 512 individuals, where SEVERAL wavefronts share a pair and all of them wait for the four lanes'
  // chain (2,000 individuals: 6e6 replayed pairs/s), go to the lanes whatever the launch's size, with a cap on a lane's EM
  // steps so that the launch does not last as long as its slowest pair (what is over the cap is the wavefront kernel's)
  const uint32_t team_waves = replay_lkl_waves((uint32_t)c->n_ind);  // (0: beyond 4,096 individuals -- the lanes or nothing)
  const bool big_cohort = team_waves >= 2 || team_waves == 0;
  if (big_cohort) lanes_from = 0;
  if (team_waves == 0 && !c->xT_ready.load()) return fail(c, NGSLD_ERR_INVALID, "device-side replay without a kernel for this cohort");
  if (c->xT_ready && (n >= lanes_from || team_waves == 0)) {
    a.after_lanes = 1;
    // one lane per pair wherever the individual-major copy is there: the launch's bitmap becomes a list of located pairs (the
    // bits listed are cleared), the lanes work through it; what stays in the bitmap -- ill-conditioned Pearson moments, pairs
    // beyond the list -- is the wavefront-per-pair kernel's, as everything is without the copy
    ngsld_ctx::LaneScratch &ls = slot < 0 ? c->lane_scratch_dev : c->lane_scratch[slot];
    const uint64_t list_cap = std::min<uint64_t>(n, 1ull << 26);
    const size_t temp_bytes = replay_sort_temp_bytes(list_cap, (uint32_t)c->n_sites);
    // (the lanes' scratch -- 40 bytes per record of the launch + the sort's own -- is something the run can do without: where the
    // device has no room for it the launch stays with the wavefront-per-pair kernel, or the host where that has no shape)
    const bool have_scratch = ls.list.resize(list_cap) == hipSuccess && ls.keys_a.resize(list_cap) == hipSuccess &&
                              ls.keys_b.resize(list_cap) == hipSuccess && ls.vals_a.resize(list_cap) == hipSuccess &&
                              ls.vals_b.resize(list_cap) == hipSuccess && ls.temp.resize(temp_bytes ? temp_bytes : 1) == hipSuccess;
    if (!have_scratch) {
      (void)hipGetLastError();
      ls.list.release(); ls.keys_a.release(); ls.keys_b.release(); ls.vals_a.release(); ls.vals_b.release(); ls.temp.release();
      a.after_lanes = 0;
      if (team_waves == 0) {
        HIP_TRY(c, launch_replay_leftover(a, st));
        return NGSLD_OK;
      }
      HIP_TRY(c, launch_replay_lkl(a, c->n_cus, st));
      return NGSLD_OK;
    }
    HIP_TRY(c, launch_replay_expand(a, ls.list.p, list_cap, st));
    HIP_TRY(c, launch_replay_sort(a, ls.list.p, list_cap, ls.keys_a.p, ls.keys_b.p, ls.vals_a.p, ls.vals_b.p, ls.temp.p, temp_bytes, st));
    int lane_waves = n >= (1ull << 22) ? 4 : 1;
    if (n < (1ull << 22) && team_waves != 0) {  // (no wavefront kernel to hand a long pair back to: no cap)
      a.lane_iter_cap = 12;
      if (const char *v = test_knob("LANE_ITER_CAP")) a.lane_iter_cap = (uint32_t)std::strtoul(v, nullptr, 10);  // (tests: hand-backs on every pair)
    }
    HIP_TRY(c, launch_replay_lanes(a, ls.list.p, ls.vals_b.p, c->d_xT.p, c->n_cus, lane_waves, st));
  }
  if (team_waves == 0)
    HIP_TRY(c, launch_replay_leftover(a, st));
  else
    HIP_TRY(c, launch_replay_lkl(a, c->n_cus, st));
  return NGSLD_OK;
}

// Called-genotype matrices: the flagged pairs of a launch replayed on the device (ld_replay.hip), right behind the pair
// kernels on their stream -- before the head of the flag buffer travels to the host, before text rows are formatted.
// out_base: plan index of the launch's record 0; d_std / d_ext: where the launch wrote (device, or pinned host memory).
int device_replay(ngsld_ctx *c, uint32_t *d_flags, uint32_t cap, uint64_t out_base, uint64_t n, ngsld_rec_std *d_std,
                  ngsld_rec_ext *d_ext, hipStream_t st, int slot) {
  if (!c->replay_on || !c->replay_device || c->cfg.kernel != kHard || d_flags == nullptr || n == 0) return NGSLD_OK;
  // A launch that flags more pairs than its list holds (every pair of a monomorphic called site): the bitmap is turned into a
  // list of located pairs and a second kernel works through that -- both leave at once unless the list did overflow
  ngsld_ctx::LaneScratch &ls = slot < 0 ? c->lane_scratch_dev : c->lane_scratch[slot];
  uint64_t list_cap = std::min<uint64_t>(n, 1ull << 26);
  if (const char *v = test_knob("REPLAY_LIST_CAP")) list_cap = std::min<uint64_t>(list_cap, std::max<uint64_t>(1, std::strtoull(v, nullptr, 10)));  // tests: a list that overflows
  HIP_TRY(c, ls.list.resize(list_cap));
  ReplayHardArgs a{};
  a.list = ls.list.p;
  a.host_bits = d_flags + flag_head_words(cap) + flag_bitmap_words(n);
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
  // (text genotype files without --call_geno: the reader's own triple for a missing call, log(1/3) three times and then
  // post_prob -- another two constants; the prep pass has looked at every individual without data, engine.hip)
  a.miss_ok = (c->gopts.call_geno && !c->normalised) || c->missing_canonical ? 1 : 0;
  if (c->missing_canonical)
    replay_missing_constants_text(&a.u_lkl, &a.u_pp);
  else
    replay_missing_constants(&a.u_lkl, &a.u_pp);
  a.out_std = d_std;
  a.out_ext = d_ext;
  a.status = c->d_status.p;
  HIP_TRY(c, launch_replay_hard(a, n, st));
  {
    ReplayLklArgs x{};  // (what launch_replay_expand reads)
    x.bits = d_flags + flag_head_words(cap);
    x.host_bits = a.host_bits;
    x.n_records = n;
    x.flags = d_flags;
    x.flag_cap = cap;
    x.row_off = a.row_off;
    x.item_off = a.item_off;
    x.items = a.items;
    x.n_items = c->n_items;
    x.n_sites = a.n_sites;
    x.rec_base = out_base;
    x.rsx = c->d_rsx.p;
    x.n_ind = a.n_ind;
    x.only_if_overflow = 1;
    HIP_TRY(c, launch_replay_expand(x, ls.list.p, list_cap, st));
    HIP_TRY(c, launch_replay_hard_list(a, c->n_cus, st));
    // what the expansion could not list (a launch that flags more pairs than the list's 2^26 entries: only ngsld_run_device makes
    // launches that large) is still in the bitmap, and flags[7] tells the host that everything but the host-only pairs is done:
    // those pairs join the host-only bitmap and list (a no-op unless the launch overflowed)
    HIP_TRY(c, launch_replay_leftover(x, st));
  }
  return NGSLD_OK;
}

// Everything ngsld_run_device + the device-side replay allocate for a launch of n records, taken now -- all of it grow-only, so
// that a sequence of launches of up to n records (run_grouped: a small first group, then full ones) allocates once, before its
// first kernel, instead of freeing and allocating gigabytes between two groups (125 ms each, measured, with the device idle).
int reserve_device_run(ngsld_ctx *c, uint64_t n) {
  if (!c->replay_on || n == 0) return NGSLD_OK;
  const uint32_t cap = flag_cap_for(n);
  HIP_TRY(c, c->d_flags_dev.resize(flag_words(n, cap)));
  HIP_TRY(c, c->h_flags_dev.resize(flag_head_words(cap)));
  if (lkl_device_eligible(c)) {
    ngsld_ctx::LaneScratch &ls = c->lane_scratch_dev;
    const uint64_t list_cap = std::min<uint64_t>(n, 1ull << 26);
    const size_t temp_bytes = replay_sort_temp_bytes(list_cap, (uint32_t)c->n_sites);
    if (ls.list.resize(list_cap) != hipSuccess || ls.keys_a.resize(list_cap) != hipSuccess || ls.keys_b.resize(list_cap) != hipSuccess ||
        ls.vals_a.resize(list_cap) != hipSuccess || ls.vals_b.resize(list_cap) != hipSuccess || ls.temp.resize(temp_bytes ? temp_bytes : 1) != hipSuccess) {
      (void)hipGetLastError();  // (the launches will find out themselves, and do without the lanes)
      ls.release();
    }
  }
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
  c->flagged_pairs = c->h_flags_dev.p[0];
  if (c->h_flags_dev.p[0] == 0) return NGSLD_OK;
  const uint64_t base = c->h_row_off[c->dev_run.s1_begin], n = c->h_row_off[c->dev_run.s1_end] - base;
  bool applied = c->dev_run.dev_applied;
  if (!applied && exact_store_wanted(c, c->h_flags_dev.p[0] - c->h_flags_dev.p[1])) {
    // a likelihood matrix that flags more pairs than the host should replay: the exact store is built (once per matrix) and
    // the pairs are replayed on the device, behind the kernels on their stream
    int rcx = try_device_replay_lkl(c, c->d_flags_dev.p, c->flag_cap_dev, base, n, c->dev_run.d_std, c->dev_run.d_ext, c->dev_run.st,
                                    c->dev_run_flag_text, -1, &applied, c->n_sites);
    if (rcx != NGSLD_OK) return rcx;
    if (applied) {
      rcx = send_flag_head(c, c->d_flags_dev.p, c->h_flags_dev.p, c->flag_cap_dev, c->dev_run.st, false);
      if (rcx != NGSLD_OK) return rcx;
      HIP_TRY(c, hipStreamSynchronize(c->dev_run.st));
    }
  }
  const double t_dev = ms();
  std::vector<uint64_t> recs;
  const int rcf = flagged_records(c, c->h_flags_dev.p, c->d_flags_dev.p, c->flag_cap_dev, n, recs, applied);
  if (rcf != NGSLD_OK) return rcf;
  const double t_list = ms();
  const int rcr = replay_flagged(c, recs, base, nullptr, nullptr, c->dev_run.d_std, c->dev_run.d_ext, c->dev_run.st);
  if (trace)
    std::fprintf(stderr, "[trace] finish_device: waited for the kernels %.2f ms, exact store + device replay %.2f ms, flag list %.2f ms "
                         "(%u flagged, %u on the device, %zu for the host), host replay + patch %.2f ms\n", t_sync, t_dev - t_sync,
                 t_list - t_dev, c->h_flags_dev.p[0], applied ? c->h_flags_dev.p[2] : 0u, recs.size(), ms() - t_list);
  return rcr;
}
}  // namespace eng
}  // namespace ngsld

extern "C" {

int ngsld_set_replay_source(ngsld_ctx *c, ngsld_read_sites_fn read, void *user) {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  if (!c->have_geno) return fail(c, NGSLD_ERR_INVALID, "set the genotype data before its replay source");
  stop_exact_store(c);  // (the exact store is built from the source)
  c->replay_matrix = nullptr;
  c->replay_read = read;
  c->replay_user = user;
  c->planned = false;  // a --min_maf tie is settled at plan time
  return NGSLD_OK;
}

int ngsld_set_replay_matrix(ngsld_ctx *c, const double *values) {
  if (c == nullptr) return NGSLD_ERR_INVALID;
  if (!c->have_geno) return fail(c, NGSLD_ERR_INVALID, "set the genotype data before its replay source");
  stop_exact_store(c);  // (the exact store is built from the source)
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

int ngsld_set_exact_store(ngsld_ctx *c, int mode) {
  if (c == nullptr || mode < 0 || mode > 2) return NGSLD_ERR_INVALID;
  c->exact_mode = mode;
  return NGSLD_OK;
}

int ngsld_replay_info(ngsld_ctx *c, ngsld_replay_stats_t *out) {
  if (c == nullptr || out == nullptr) return NGSLD_ERR_INVALID;
  out->pairs_flagged = c->flagged_pairs;
  out->pairs_replayed = c->replayed_pairs;
  out->pairs_on_device = c->replayed_on_device;
  out->pairs_on_host = c->replayed_pairs - c->replayed_on_device;
  out->sites_reevaluated = c->replayed_sites;
  out->exact_store = c->exact_state.load() == 2 ? (c->exact_alias ? 1 : 2) : 0;
  out->text_rows_patched = (int32_t)std::min<uint64_t>(c->text_rows_patched, 0x7fffffffull);
  out->exact_store_build_s = c->exact_build_s;
  out->sites_degenerate = c->h_skip_count;
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

}  // extern "C"
