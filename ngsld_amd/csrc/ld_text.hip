// ld_text.hip -- the TSV rows of a batch, formatted on the device (gfx950).
//
// What it replaces: calc_pair_LD's fprintf block (ngsLD.cpp:310-352) -- on the host that is
// ngsld_host_write_batch (host_io.cpp), whose bytes this file reproduces exactly: same "%f" / "%.0f" rule (the
// EXACT binary value rounded half-to-even at the decimal, as glibc prints it), "-nan" / "inf", the float chi2 of
// ngsLD.cpp:328-333, the literal "0.000000" of the loglike column.  At kernel rates the text is the bulk of an
// end-to-end run (9.9e7 extended rows = 15 GB): formatted here it costs a few tens of milliseconds and the host
// only writes bytes.
//
// Two passes over the batch's pairs, one thread per candidate of an item: lengths, then (after an exclusive prefix
// sum, hipCUB) the rows at their final offsets -- rows come out in (s1, s2) order, byte for byte what the host writer
// produces.  A value outside the fast path of the formatter (>= 2^52, or a quotient beyond 63 bits: only absurd D' /
// chi2 of degenerate pairs get there) raises needs_host and the caller falls back to the records for that batch.
#include <hipcub/hipcub.hpp>

#include "ld_text.h"

namespace ngsld {
namespace {

struct Counter {
  uint64_t n = 0;
  __device__ __forceinline__ void put(char) { ++n; }
};
struct Writer {
  char *p;
  __device__ __forceinline__ void put(char c) { *p++ = c; }
};
// A row goes out in aligned 8-byte words: the characters collect in a register and every eighth is one store (a thread's
// row is ~150 bytes of its own, so a byte store is a whole memory request for one byte -- and the 64 lanes of a wave-wide
// store hit 64 different cache lines either way).  The partial words at the two ends of the row, which it shares with the
// neighbouring rows, go out byte by byte.
// (Tried first: rows composed in LDS, the wavefront's contiguous piece copied out 16 bytes per lane.  Its 49 KB of LDS per
// workgroup cannot sit beside the pair kernel's two 72 KB workgroups on a CU, so the write pass, which runs beside the next
// batch's pair kernel, starved: 332 ms against 96 ms for the plain byte stores over configs[2].)
#ifndef NGSLD_TEXT_WORDS
#define NGSLD_TEXT_WORDS 1  // build-time A/B switch
#endif
struct WordWriter {
  char *p;        // aligned address of the word being filled
  uint64_t acc;
  uint32_t cnt;   // bytes of the word filled so far (counting the leading bytes that are not this row's)
  uint32_t head;  // leading bytes of the FIRST word that belong to the previous row
  __device__ __forceinline__ explicit WordWriter(char *dst) {
    const uint32_t mis = (uint32_t)((uintptr_t)dst & 7u);
    p = dst - mis;
    acc = 0;
    cnt = head = mis;
  }
  __device__ __forceinline__ void put(char c) {
    acc |= (uint64_t)(unsigned char)c << (8 * cnt);
    if (++cnt == 8) {
      if (head) {
        for (uint32_t b = head; b < 8; ++b) p[b] = (char)(acc >> (8 * b));
        head = 0;
      } else {
        *reinterpret_cast<uint64_t *>(p) = acc;
      }
      p += 8;
      acc = 0;
      cnt = 0;
    }
  }
  __device__ __forceinline__ void finish() {
    for (uint32_t b = head; b < cnt; ++b) p[b] = (char)(acc >> (8 * b));
  }
};

template <class E>
__device__ __forceinline__ void put_u64(E &e, uint64_t v) {
  char tmp[24];
  int n = 0;
  do {
    tmp[n++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) e.put(tmp[--n]);
}

// host_io.cpp put_fixed<DECIMALS>, minus its snprintf fallback: returns false where that would be taken
template <int DECIMALS, class E>
__device__ __forceinline__ bool put_fixed(E &e, double v) {
  const uint64_t bits = (uint64_t)__double_as_longlong(v);
  const bool neg = bits >> 63;
  const int ebits = (int)((bits >> 52) & 0x7ff);
  uint64_t m = bits & 0xfffffffffffffull;
  if (ebits == 0x7ff) {
    if (m) {
      e.put('-'); e.put('n'); e.put('a'); e.put('n');
      return true;
    }
    if (neg) e.put('-');
    e.put('i'); e.put('n'); e.put('f');
    return true;
  }
  int ex;  // value = m * 2^ex
  if (ebits == 0) {
    ex = -1074;
  } else {
    m |= 1ull << 52;
    ex = ebits - 1075;
  }
  constexpr uint64_t kScale = DECIMALS == 6 ? 1000000ull : 1ull;
  uint64_t q;
  if (ex >= 0) {
    if (ex > 10 || (DECIMALS == 6 && ex > -1)) return false;
    q = (m << ex) * kScale;
  } else {
    const int k = -ex;
    const unsigned __int128 M = (unsigned __int128)m * kScale;
    if (k >= 127) {
      q = 0;
    } else {
      const unsigned __int128 quo = M >> k;
      if (quo >> 63) return false;
      q = (uint64_t)quo;
      const unsigned __int128 rem = M - (quo << k), half = (unsigned __int128)1 << (k - 1);
      if (rem > half || (rem == half && (q & 1))) ++q;
    }
  }
  if (neg) e.put('-');
  if (DECIMALS == 0) {
    put_u64(e, q);
    return true;
  }
  put_u64(e, q / 1000000ull);
  uint32_t f = (uint32_t)(q % 1000000ull);
  char d[6];
  for (int i = 5; i >= 0; --i) {
    d[i] = (char)('0' + f % 10);
    f /= 10;
  }
  e.put('.');
  for (int i = 0; i < 6; ++i) e.put(d[i]);
  return true;
}

template <class E>
__device__ __forceinline__ void put_label(E &e, const TextArgs &A, uint32_t s) {
  if (A.labels == nullptr) {  // glibc prints "(null)" for the reference's NULL labels (ngsLD.cpp:135)
    e.put('('); e.put('n'); e.put('u'); e.put('l'); e.put('l'); e.put(')');
    return;
  }
  const uint64_t b = A.label_off[s], n = A.label_off[s + 1] - b;
  for (uint64_t i = 0; i < n; ++i) e.put(A.labels[b + i]);
}

// host_io.cpp format_row
template <class E>
__device__ __forceinline__ bool format_row(E &e, const TextArgs &A, uint32_t s1, uint32_t s2, uint64_t k) {
  bool ok = true;
  put_label(e, A, s1);
  e.put('\t');
  put_label(e, A, s2);
  e.put('\t');
  // ngsLD.cpp:241: dist is the running sum of pos_dist over (s1, s2]; INFINITY once a chromosome change is passed
  const double dist = A.infc[s2] != A.infc[s1] ? __builtin_inf() : A.cum[s2] - A.cum[s1];
  ok &= put_fixed<0>(e, dist);
  e.put('\t');
  const ngsld_rec_std sr = A.std_rec[k];
  ok &= put_fixed<6>(e, sr.r2_ExpG); e.put('\t');
  ok &= put_fixed<6>(e, sr.D);       e.put('\t');
  ok &= put_fixed<6>(e, sr.Dp);      e.put('\t');
  ok &= put_fixed<6>(e, sr.r2);
  if (A.ext_rec != nullptr) {
    const ngsld_rec_ext er = A.ext_rec[k];
    const double *h = er.hap;
    const double hm0 = 1 - (h[0] + h[1]);  // ngsLD.cpp:297-298
    const double hm1 = 1 - (h[0] + h[2]);
    float chi2 = 0;  // ngsLD.cpp:328-333, float arithmetic as there
    const float freq_A = (float)(h[0] + h[1]);
    const float freq_B = (float)(h[0] + h[2]);
    const float exp_hap[4] = {freq_A * freq_B, freq_A * (1 - freq_B), (1 - freq_A) * freq_B,
                              (1 - freq_A) * (1 - freq_B)};
    for (int i = 0; i < 4; i++) {
      const double d = h[i] - (double)exp_hap[i];
      chi2 = (float)((double)chi2 + d * d / (double)exp_hap[i]);
    }
    e.put('\t');
    put_u64(e, er.n_ind_data);  // ngsLD.cpp:336-349
    e.put('\t');
    ok &= put_fixed<6>(e, A.maf[s1]); e.put('\t');
    ok &= put_fixed<6>(e, A.maf[s2]); e.put('\t');
    for (int i = 0; i < 4; i++) {
      ok &= put_fixed<6>(e, h[i]);
      e.put('\t');
    }
    ok &= put_fixed<6>(e, hm0); e.put('\t');
    ok &= put_fixed<6>(e, hm1); e.put('\t');
    ok &= put_fixed<6>(e, (double)chi2);
    const char lit[] = "\t0.000000\t";  // loglike is the literal 0.0 (ngsLD.cpp:347)
    for (int i = 0; i < 10; ++i) e.put(lit[i]);
    put_u64(e, er.n_iter);
  }
  e.put('\n');
  return ok;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void text_kernel(TextArgs A) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t i = t >> 6;
  const uint32_t c = (uint32_t)(t & 63u);
  if (i >= A.n_items) return;
  const ngsld_item it = A.items[i];
  if (c >= it.count || !((it.mask >> c) & 1ull)) return;  // ngsLD.cpp:270-282: not a computed pair
  const uint64_t k = it.first_record - A.out_base + (uint64_t)__popcll(it.mask & ((1ull << c) - 1ull));
  const uint32_t s1 = it.s1, s2 = it.s2_begin + c;
  if (WRITE && A.text_cap != 0 && A.offs[k] + A.lens[k] > A.text_cap) {
    *A.overflow = 1;
    return;
  }
  if (WRITE && NGSLD_TEXT_WORDS) {
    WordWriter w(A.text + A.offs[k]);
    (void)format_row(w, A, s1, s2, k);
    w.finish();
  } else if (WRITE) {
    Writer w{A.text + A.offs[k]};
    (void)format_row(w, A, s1, s2, k);
  } else {
    Counter n;
    if (!format_row(n, A, s1, s2, k)) atomicExch(A.needs_host, 1);
    A.lens[k] = n.n;
  }
}

// Rows whose records were replaced (exact-order replay): their lengths again.  changed is set when one differs from the
// length pass -- only then do the prefix sums have to be taken again (a replayed record moves a sixth decimal, or turns a
// rounded zero's sign: the row's length changes once in a few thousand replays).
__global__ void text_relength_kernel(TextArgs A, const uint64_t *rec, const uint32_t *s1, const uint32_t *s2, uint64_t n,
                                     uint64_t *changed) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint64_t k = rec[t];
  Counter c;
  if (!format_row(c, A, s1[t], s2[t], k)) atomicExch(A.needs_host, 1);
  if (A.lens[k] != c.n) {
    A.lens[k] = c.n;
    *changed = 1;
  }
}

__global__ void text_total_kernel(const uint64_t *lens, const uint64_t *offs, uint64_t n, uint64_t *total) {
  *total = n ? offs[n - 1] + lens[n - 1] : 0;
}

hipError_t launch(const TextArgs &a, hipStream_t stream, bool write) {
  // one thread per candidate; item lists of any length go out in launches of at most 2^22 workgroups (see
  // launch_pair_kernel: gridDim.x * blockDim.x has to stay below 2^32)
  const uint64_t max_items = (1ull << 22) * 4;
  for (uint64_t off = 0; off < a.n_items; off += max_items) {
    TextArgs b = a;
    b.items = a.items + off;
    b.n_items = a.n_items - off < max_items ? a.n_items - off : max_items;
    const unsigned blocks = (unsigned)((b.n_items * 64 + 255) / 256);
    if (write)
      hipLaunchKernelGGL(text_kernel<true>, dim3(blocks), dim3(256), 0, stream, b);
    else
      hipLaunchKernelGGL(text_kernel<false>, dim3(blocks), dim3(256), 0, stream, b);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace

hipError_t launch_text_lengths(const TextArgs &a, hipStream_t stream) { return launch(a, stream, false); }
hipError_t launch_text_relength(const TextArgs &a, const uint64_t *rec, const uint32_t *s1, const uint32_t *s2, uint64_t n,
                                uint64_t *changed, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(text_relength_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, a, rec, s1, s2, n, changed);
  return hipGetLastError();
}
hipError_t launch_text_write(const TextArgs &a, hipStream_t stream) { return launch(a, stream, true); }

size_t text_scan_temp_bytes(uint64_t n) {
  size_t bytes = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (int)n);
  return bytes;
}

hipError_t text_scan(void *temp, size_t temp_bytes, const uint64_t *lens, uint64_t *offs, uint64_t n, uint64_t *total,
                     hipStream_t stream) {
  if (n > 0x7fffffffull) return hipErrorInvalidValue;
  if (n) {
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, lens, offs, (int)n, stream);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(text_total_kernel, dim3(1), dim3(1), 0, stream, lens, offs, n, total);
  return hipGetLastError();
}

}  // namespace ngsld
