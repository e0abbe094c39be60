// host_io.cpp -- host-side readers and TSV writer of the drop-in (declared in include/ngsld_host.h).
// Pure C++17 + zlib, no device code: this is the part of the reference's L0/L4 layers
// (shared/read_data.cpp, shared/gen_func.cpp read_file, ngsLD.cpp fprintf) the new engine keeps on the host.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ngsld_host.h"

struct ngsld_pos {
  std::vector<double> pos_dist;
  std::vector<std::string> labels;
};

namespace {

std::atomic<int> g_host_threads{1};

int set_err(char *err, size_t errlen, const char *msg) {
  if (err && errlen) std::snprintf(err, errlen, "%s", msg);
  return NGSLD_ERR_INVALID;
}

bool slurp(const char *path, std::string &out) {
  gzFile fh = std::strcmp(path, "-") == 0 ? gzdopen(0, "rb") : gzopen(path, "rb");
  if (fh == nullptr) return false;
  gzbuffer(fh, 1 << 20);
  std::vector<char> buf(1 << 20);
  for (;;) {
    const int n = gzread(fh, buf.data(), (unsigned)buf.size());
    if (n <= 0) break;
    out.append(buf.data(), (size_t)n);
  }
  gzclose(fh);
  return true;
}

// ---- exact "%f" / "%.0f" without printf ------------------------------------------------------------
// glibc prints the EXACT binary value rounded half-to-even at the requested decimal.  |v| = m * 2^e with a
// 53-bit m; m * 10^6 < 2^73 fits an unsigned __int128, so round(m * 10^6 / 2^-e) is computed exactly with
// integer arithmetic.  Values too large for the 64-bit quotient (>= ~9.2e12) fall back to snprintf; NaN is
// "-nan" (see the header), +-inf "inf"/"-inf".  Checked against snprintf on millions of values (tests).
inline char *put_u64(char *p, uint64_t v) {
  char tmp[24];
  int n = 0;
  do {
    tmp[n++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) *p++ = tmp[--n];
  return p;
}

template <int DECIMALS>  // 6 -> "%f", 0 -> "%.0f"
inline char *put_fixed(char *p, double v) {
  uint64_t bits;
  std::memcpy(&bits, &v, 8);
  const bool neg = bits >> 63;
  const int ebits = (int)((bits >> 52) & 0x7ff);
  uint64_t m = bits & 0xfffffffffffffull;
  if (ebits == 0x7ff) {
    if (m) return (char *)std::memcpy(p, "-nan", 4) + 4;
    if (neg) *p++ = '-';
    return (char *)std::memcpy(p, "inf", 3) + 3;
  }
  int e;  // value = m * 2^e
  if (ebits == 0) {
    e = -1074;
  } else {
    m |= 1ull << 52;
    e = ebits - 1075;
  }
  constexpr uint64_t kScale = DECIMALS == 6 ? 1000000ull : 1ull;
  uint64_t q;
  if (e >= 0) {
    if (e > 10 || (DECIMALS == 6 && e > -1)) {  // >= 2^53: beyond the fast path
      return p + std::snprintf(p, 400, DECIMALS == 6 ? "%f" : "%.0f", v);
    }
    q = (m << e) * kScale;
  } else {
    const int k = -e;
    const unsigned __int128 M = (unsigned __int128)m * kScale;
    if (k >= 127) {
      q = 0;  // M < 2^73: far below one half unit
    } else {
      const unsigned __int128 quo = M >> k;
      if (quo >> 63) return p + std::snprintf(p, 400, DECIMALS == 6 ? "%f" : "%.0f", v);
      q = (uint64_t)quo;
      const unsigned __int128 rem = M - (quo << k), half = (unsigned __int128)1 << (k - 1);
      if (rem > half || (rem == half && (q & 1))) ++q;
    }
  }
  if (neg) *p++ = '-';
  if (DECIMALS == 0) return put_u64(p, q);
  p = put_u64(p, q / 1000000ull);
  uint32_t f = (uint32_t)(q % 1000000ull);
  *p++ = '.';
  for (int i = 5; i >= 0; --i) {
    p[i] = (char)('0' + f % 10);
    f /= 10;
  }
  return p + 6;
}

// malloc'd, never zero-filled, grown with the first `keep` bytes preserved
struct RawBuf {
  char *p = nullptr;
  size_t cap = 0;
  RawBuf() = default;
  RawBuf(const RawBuf &) = delete;
  RawBuf &operator=(const RawBuf &) = delete;
  RawBuf(RawBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
  ~RawBuf() { std::free(p); }
  char *data() const { return p; }
  size_t size() const { return cap; }
  bool reserve(size_t n, size_t keep) {
    if (n <= cap) return true;
    char *q = (char *)std::malloc(n);
    if (q == nullptr) return false;
    if (keep) std::memcpy(q, p, keep);
    std::free(p);
    p = q;
    cap = n;
    return true;
  }
};

inline char *put_str(char *p, const char *s) {
  const size_t n = std::strlen(s);
  std::memcpy(p, s, n);
  return p + n;
}

// one TSV row; the caller guarantees room for the two labels + 1024 bytes
inline char *format_row(char *p, const char *label1, const char *label2, double dist, const ngsld_rec_std *sr,
                        const ngsld_rec_ext *er, double maf1, double maf2) {
  // glibc prints "(null)" for the reference's NULL labels when no --pos is given (ngsLD.cpp:135)
  p = put_str(p, label1 ? label1 : "(null)");
  *p++ = '\t';
  p = put_str(p, label2 ? label2 : "(null)");
  *p++ = '\t';
  p = put_fixed<0>(p, dist);  // ngsLD.cpp:314-322
  *p++ = '\t';
  p = put_fixed<6>(p, sr->r2_ExpG); *p++ = '\t';
  p = put_fixed<6>(p, sr->D);       *p++ = '\t';
  p = put_fixed<6>(p, sr->Dp);      *p++ = '\t';
  p = put_fixed<6>(p, sr->r2);
  if (er != nullptr) {
    const double *h = er->hap;
    const double hm0 = 1 - (h[0] + h[1]);  // ngsLD.cpp:297-298
    const double hm1 = 1 - (h[0] + h[2]);
    float chi2 = 0;  // ngsLD.cpp:328-333, float arithmetic as there
    const float freq_A = (float)(h[0] + h[1]);
    const float freq_B = (float)(h[0] + h[2]);
    const float exp_hap[4] = {freq_A * freq_B, freq_A * (1 - freq_B), (1 - freq_A) * freq_B,
                              (1 - freq_A) * (1 - freq_B)};
    for (int i = 0; i < 4; i++) {
      const double d = h[i] - (double)exp_hap[i];
      chi2 = (float)((double)chi2 + d * d / (double)exp_hap[i]);  // pow(d, 2) is d*d exactly
    }
    *p++ = '\t';
    p = put_u64(p, er->n_ind_data);  // ngsLD.cpp:336-349
    *p++ = '\t';
    p = put_fixed<6>(p, maf1); *p++ = '\t';
    p = put_fixed<6>(p, maf2); *p++ = '\t';
    for (int i = 0; i < 4; i++) {
      p = put_fixed<6>(p, h[i]);
      *p++ = '\t';
    }
    p = put_fixed<6>(p, hm0); *p++ = '\t';
    p = put_fixed<6>(p, hm1); *p++ = '\t';
    p = put_fixed<6>(p, (double)chi2);
    p = put_str(p, "\t0.000000\t");  // loglike is the literal 0.0 (ngsLD.cpp:347)
    p = put_u64(p, er->n_iter);
  }
  *p++ = '\n';
  return p;
}

}  // namespace

// No exception crosses the C-ABI: host-side out-of-memory comes back as an error code and text.
#define NGSLD_HOST_CATCH                                                              \
  catch (const std::bad_alloc &) { return set_err(err, errlen, "out of host memory"); } \
  catch (...) { return set_err(err, errlen, "unexpected C++ exception"); }

extern "C" {

int ngsld_host_read_pos(const char *path, int header, uint64_t n_sites, ngsld_pos **out, char *err, size_t errlen) try {
  if (path == nullptr || out == nullptr) return set_err(err, errlen, "invalid argument");
  *out = nullptr;
  std::string text;
  if (!slurp(path, text)) return set_err(err, errlen, "cannot open file!");

  // lines: drop one trailing '\n' or '\r' (chomp, gen_func.cpp:190-197), skip empty and '#' lines
  // (gen_func.cpp:258-261), then skip `header` lines (:263-267).  Unlike the reference, a last line
  // without a newline is kept (there it is lost to the gzeof() test at gen_func.cpp:253).
  std::vector<std::string> lines;
  uint64_t skip = header ? 1 : 0;
  size_t b = 0;
  while (b < text.size()) {
    size_t e = text.find('\n', b);
    const bool last = e == std::string::npos;
    if (last) e = text.size();
    size_t len = e - b;
    if (last && len > 0 && text[b + len - 1] == '\r') --len;  // no '\n': chomp takes a trailing '\r'
    if (len > 0 && text[b] != '#') {
      if (skip > 0)
        --skip;
      else
        lines.emplace_back(text, b, len);
    }
    b = e + 1;
  }
  // In the reference's order: a file without a usable line leaves read_file's array NULL, which read_split reports as a file
  // it could not open (read_data.cpp:134-136); read_split then counts the fields of every line it got (:139-147); only then
  // does read_dist compare the number of lines (:178-181).
  if (lines.empty()) return set_err(err, errlen, "cannot open file!");
  // every line must have the same number of TAB-separated fields, at least 2 (read_data.cpp:139-147,180-181)
  size_t n_fields = 0;
  for (const auto &l : lines) {
    size_t nf = 1;
    for (char ch : l) nf += ch == '\t';
    if (n_fields == 0) n_fields = nf;
    if (nf != n_fields) return set_err(err, errlen, "invalid number of fields in file!");
  }
  if (lines.size() != n_sites) return set_err(err, errlen, "wrong number of lines in POS file!");
  if (n_fields < 2) return set_err(err, errlen, "wrong POS file format!");

  ngsld_pos *p = new ngsld_pos();
  p->pos_dist.resize(n_sites);
  std::string prev_chr;
  unsigned long prev_pos = 0;
  for (uint64_t s = 0; s < n_sites; ++s) {
    const std::string &l = lines[s];
    const size_t t1 = l.find('\t');
    const size_t t2 = l.find('\t', t1 + 1);
    const std::string chr = l.substr(0, t1);
    const std::string f1 = l.substr(t1 + 1, t2 == std::string::npos ? std::string::npos : t2 - t1 - 1);
    const double posd = std::strtod(f1.c_str(), nullptr);
    if (posd == 0) {  // the reference treats this as a header and never advances (read_data.cpp:188-195)
      delete p;
      return set_err(err, errlen, "header line found in POS file; use --posH for files with a header");
    }
    if (prev_chr.empty()) prev_chr = chr;  // read_data.cpp:199-200: "first chromosome" is whenever the stored name is empty -- also after a line with an empty first field
    if (chr == prev_chr) {
      p->pos_dist[s] = posd - (double)prev_pos;  // read_data.cpp:204
      if (p->pos_dist[s] < 1) {
        delete p;
        return set_err(err, errlen, "invalid distance between adjacent sites!");
      }
    } else {
      p->pos_dist[s] = INFINITY;  // read_data.cpp:208
      prev_chr = chr;
    }
    prev_pos = std::strtoul(f1.c_str(), nullptr, 0);  // read_data.cpp:211 (base 0, as there)
  }
  p->labels = std::move(lines);
  for (auto &l : p->labels) {  // ngsLD.cpp:128-132
    const size_t t = l.find('\t');
    if (t != std::string::npos) l[t] = ':';
  }
  *out = p;
  return NGSLD_OK;
} NGSLD_HOST_CATCH

const double *ngsld_host_pos_dist(const ngsld_pos *p) { return p ? p->pos_dist.data() : nullptr; }
const char *ngsld_host_label(const ngsld_pos *p, uint64_t site) {
  return (p && site < p->labels.size()) ? p->labels[site].c_str() : nullptr;
}
void ngsld_host_free_pos(ngsld_pos *p) { delete p; }
ngsld_pos *ngsld_host_pos_slice(const ngsld_pos *p, uint64_t begin, uint64_t end) {
  if (p == nullptr || begin > end || end > p->labels.size()) return nullptr;
  ngsld_pos *q = new ngsld_pos();
  q->pos_dist.assign(p->pos_dist.begin() + (ptrdiff_t)begin, p->pos_dist.begin() + (ptrdiff_t)end);
  q->labels.assign(p->labels.begin() + (ptrdiff_t)begin, p->labels.begin() + (ptrdiff_t)end);
  return q;
}

int ngsld_host_geno_size_ok(uint64_t file_size, uint64_t n_ind, uint64_t n_sites) {
  if (n_ind == 0) return 0;
  return n_sites == file_size / sizeof(double) / n_ind / 3 ? 1 : 0;  // ngsLD.cpp:55
}

int ngsld_host_read_geno_bin(const char *path, uint64_t n_ind, uint64_t n_sites, double *out_raw, char *err,
                             size_t errlen) try {
  if (path == nullptr || out_raw == nullptr) return set_err(err, errlen, "invalid argument");
  const uint64_t total = n_sites * n_ind * 3 * sizeof(double);
  // A plain (not gzip-compressed) regular file is read with pread by the host threads (ngsld_host_set_threads) straight
  // into the caller's buffer -- gzread's transparent mode is one thread and one more copy; same error texts.
  if (std::strcmp(path, "-") != 0) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return set_err(err, errlen, "cannot open GENO file!");
    struct stat st;
    unsigned char magic[2] = {0, 0};
    const bool regular = ::fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
    const bool gz = ::pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (regular && !gz) {
      if ((uint64_t)st.st_size < total) {
        ::close(fd);
        return set_err(err, errlen, "GENO file at premature EOF. Check GENO file and number of sites!");
      }
      if ((uint64_t)st.st_size > total) {
        ::close(fd);
        return set_err(err, errlen, "GENO file not at EOF. Check GENO file and number of sites!");
      }
      const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(1, g_host_threads.load()), total >> 24));
      std::atomic<int> failed{0};
      auto work = [&](int t) {
        uint64_t lo = total * (uint64_t)t / (uint64_t)nt, hi = total * (uint64_t)(t + 1) / (uint64_t)nt;
        char *dst = reinterpret_cast<char *>(out_raw);
        while (lo < hi) {
          const ssize_t n = ::pread(fd, dst + lo, (size_t)std::min<uint64_t>(hi - lo, 1u << 30), (off_t)lo);
          if (n <= 0) {
            failed = 1;
            return;
          }
          lo += (uint64_t)n;
        }
      };
      std::vector<std::thread> th;
      for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
      work(0);
      for (auto &x : th) x.join();
      ::close(fd);
      if (failed) return set_err(err, errlen, "cannot read binary GENO file. Check GENO file and number of sites!");
      return NGSLD_OK;
    }
    ::close(fd);
  }
  gzFile fh = std::strcmp(path, "-") == 0 ? gzdopen(0, "rb") : gzopen(path, "rb");
  if (fh == nullptr) return set_err(err, errlen, "cannot open GENO file!");
  gzbuffer(fh, 1 << 22);
  uint64_t got = 0;
  char *dst = reinterpret_cast<char *>(out_raw);
  while (got < total) {
    const uint64_t want = std::min<uint64_t>(total - got, 1u << 30);
    const int n = gzread(fh, dst + got, (unsigned)want);
    if (n <= 0) break;
    got += (uint64_t)n;
  }
  if (got != total) {
    const bool eof = gzeof(fh);
    gzclose(fh);
    return set_err(err, errlen,
                   eof ? "GENO file at premature EOF. Check GENO file and number of sites!"
                       : "cannot read binary GENO file. Check GENO file and number of sites!");
  }
  char c;
  (void)gzread(fh, &c, 1);  // read_data.cpp:107-109
  const bool at_eof = gzeof(fh);
  gzclose(fh);
  if (!at_eof) return set_err(err, errlen, "GENO file not at EOF. Check GENO file and number of sites!");
  return NGSLD_OK;
} NGSLD_HOST_CATCH

int ngsld_host_read_geno_bin_range(const char *path, uint64_t n_ind, uint64_t site_begin, uint64_t n_sites,
                                   double *out_raw, char *err, size_t errlen) try {
  if (path == nullptr || out_raw == nullptr) return set_err(err, errlen, "invalid argument");
  const uint64_t site_bytes = n_ind * 3 * sizeof(double);
  const uint64_t off = site_begin * site_bytes, total = n_sites * site_bytes;
  char *dst = reinterpret_cast<char *>(out_raw);
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return set_err(err, errlen, "cannot open GENO file!");
  unsigned char magic[2] = {0, 0};
  const bool gz = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
  uint64_t got = 0;
  if (!gz) {  // plain file: positioned reads, nothing before the slab is touched
    while (got < total) {
      const ssize_t n = pread(fd, dst + got, (size_t)std::min<uint64_t>(total - got, 1u << 30), (off_t)(off + got));
      if (n <= 0) break;
      got += (uint64_t)n;
    }
    close(fd);
  } else {  // compressed: zlib has to inflate its way to the slab
    close(fd);
    gzFile fh = gzopen(path, "rb");
    if (fh == nullptr) return set_err(err, errlen, "cannot open GENO file!");
    gzbuffer(fh, 1 << 22);
    if (gzseek(fh, (z_off_t)off, SEEK_SET) == (z_off_t)off)
      while (got < total) {
        const int n = gzread(fh, dst + got, (unsigned)std::min<uint64_t>(total - got, 1u << 30));
        if (n <= 0) break;
        got += (uint64_t)n;
      }
    gzclose(fh);
  }
  if (got != total)
    return set_err(err, errlen, "GENO file at premature EOF. Check GENO file and number of sites!");
  return NGSLD_OK;
} NGSLD_HOST_CATCH

void ngsld_host_set_threads(int n_threads) { g_host_threads.store(n_threads < 1 ? 1 : n_threads); }

int ngsld_host_read_geno_text(const char *path, int in_probs, int log_scale, uint64_t n_ind, uint64_t n_sites,
                              double *out_raw, int *out_log_scale, char *err, size_t errlen) try {
  if (path == nullptr || out_raw == nullptr || out_log_scale == nullptr) return set_err(err, errlen, "invalid argument");
  std::string text;
  if (!slurp(path, text)) return set_err(err, errlen, "cannot open GENO file!");
  *out_log_scale = in_probs ? (log_scale ? 1 : 0) : 1;
  const uint64_t need = n_ind * (in_probs ? 3 : 1);

  // numeric fields of one line (split(char*, sep, double**), gen_func.cpp:381-410: a token counts iff strtod eats
  // all of it); tokens are bounded by blanks / TABs, strtod never runs past one because those stop it
  auto numeric_fields = [&](size_t b, size_t len, std::vector<double> &vals) {
    vals.clear();
    const char *base = text.c_str();  // NUL-terminated: a last token touching the end of the buffer is safe
    size_t p = b;
    const size_t end = b + len;
    while (p < end) {
      size_t q = p;
      while (q < end && base[q] != ' ' && base[q] != '\t') ++q;
      if (q > p) {
        char *stop = nullptr;
        const double v = std::strtod(base + p, &stop);
        if (stop == base + q) vals.push_back(v);
      }
      p = q + 1;
    }
  };
  // one data line -> one site row; returns an error text or nullptr
  auto fill_site = [&](const std::vector<double> &vals, uint64_t s) -> const char * {
    if (vals.size() < need) return "wrong GENO file format. Less fields than expected!";
    const double *ptr = vals.data() + (vals.size() - need);  // last n_ind*n_geno columns (read_data.cpp:80-81)
    double *row = out_raw + s * n_ind * 3;
    if (in_probs) {
      std::memcpy(row, ptr, need * sizeof(double));
      return nullptr;
    }
    for (uint64_t i = 0; i < n_ind; ++i) {
      const int g = (int)ptr[i];
      double *t = row + 3 * i;
      if (g >= 0) {
        if (g > 2) return "wrong GENO file format. Genotypes must be coded as {-1,0,1,2} !";
        t[0] = t[1] = t[2] = -1e15;  // init_ptr(..., -INF), read_data.cpp:21
        t[g] = 0.0;                  // log(1), :92
      } else {
        t[0] = t[1] = t[2] = std::log(1.0 / 3.0);  // :94
      }
    }
    return nullptr;
  };

  // line index: (begin, length) after chomp (gen_func.cpp:190-197); length 0 = an empty line
  std::vector<std::pair<size_t, size_t>> lines;
  for (size_t b = 0; b < text.size();) {
    size_t e = text.find('\n', b);
    if (e == std::string::npos) e = text.size();
    size_t len = e - b;
    if (e == text.size() && len > 0 && text[b + len - 1] == '\r') --len;
    lines.emplace_back(b, len);
    b = e + 1;
  }

  // The reference reads line by line (read_data.cpp:46-104): while no site has been stored yet, a line with fewer numeric fields
  // than a row needs is a header and is skipped (:64); a line without ANY numeric field is skipped wherever it stands; an empty
  // line takes a site's place without filling it (:58-59); a row's own errors come up when the row is read, the two end-of-file
  // checks after all of that.  Well-formed files -- headers at the top, then rows -- go through the parallel pass below with
  // exactly that outcome; anything else (an empty line, a line without numbers among the rows) is walked in the reference's order.
  auto sequential = [&]() -> int {
    std::vector<double> v;
    size_t li = 0;
    bool empty_seen = false;
    for (uint64_t s = 0; s < n_sites;) {
      if (li >= lines.size()) return set_err(err, errlen, "GENO file at premature EOF. Check GENO file and number of sites!");
      const auto ln = lines[li++];
      if (ln.second == 0) {  // (the reference leaves this site's values uninitialised and goes on)
        empty_seen = true;
        ++s;
        continue;
      }
      numeric_fields(ln.first, ln.second, v);
      if (v.empty() || (s == 0 && v.size() < need)) continue;  // "> Header found! Skipping line..."
      if (const char *e = fill_site(v, s)) return set_err(err, errlen, e);
      ++s;
    }
    if (li < lines.size()) return set_err(err, errlen, "GENO file not at EOF. Check GENO file and number of sites!");
    // the one place this reader parts with the reference: a site that an empty line left without values is an error here
    if (empty_seen) return set_err(err, errlen, "empty line in GENO file");
    return NGSLD_OK;
  };

  // header lines at the top (read_data.cpp:64-72, s == 0 throughout)
  size_t first = 0;
  std::vector<double> vals;
  while (first < lines.size() && lines[first].second > 0) {
    numeric_fields(lines[first].first, lines[first].second, vals);
    if (vals.empty() || vals.size() < need) ++first; else break;
  }
  if (first < lines.size() && lines[first].second == 0) return sequential();
  const uint64_t avail = lines.size() - first;
  const uint64_t n_parse = std::min<uint64_t>(n_sites, avail);

  const int nt = (int)std::min<uint64_t>((uint64_t)g_host_threads.load(), n_parse ? n_parse : 1);
  std::vector<const char *> errors(nt, nullptr);
  std::vector<uint64_t> error_at(nt, ~0ull);
  std::atomic<bool> irregular{false};
  auto work = [&](int t) {
    std::vector<double> v;
    for (uint64_t s = (uint64_t)t; s < n_parse && errors[t] == nullptr && !irregular.load(std::memory_order_relaxed); s += (uint64_t)nt) {
      const auto ln = lines[first + s];
      if (ln.second == 0) {
        irregular.store(true);
        break;
      }
      numeric_fields(ln.first, ln.second, v);
      if (v.empty()) {  // a line without numbers among the rows: the reference skips it as a header (:64)
        irregular.store(true);
        break;
      }
      errors[t] = fill_site(v, s);
      if (errors[t]) error_at[t] = s;
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto &x : th) x.join();
  if (irregular.load()) return sequential();
  int worst = -1;  // the reference meets the errors in row order
  for (int t = 0; t < nt; ++t)
    if (errors[t] && (worst < 0 || error_at[t] < error_at[worst])) worst = t;
  if (worst >= 0) return set_err(err, errlen, errors[worst]);
  if (avail < n_sites) return set_err(err, errlen, "GENO file at premature EOF. Check GENO file and number of sites!");
  if (avail > n_sites) return set_err(err, errlen, "GENO file not at EOF. Check GENO file and number of sites!");
  return NGSLD_OK;
} NGSLD_HOST_CATCH

double ngsld_host_missing_call_log(void) { return std::log(1.0 / 3.0); }  // (the expression fill_site stores)

size_t ngsld_host_format_header(char *buf, size_t cap, int extend_out) {
  const int n = std::snprintf(
      buf, cap, "site1\tsite2\tdist\tr2_ExpG\tD\tDp\tr2%s\n",
      extend_out ? "\tsample_size\tmaf1\tmaf2\thap00\thap01\thap10\thap11\thap_maf1\thap_maf2\tchi2\tloglike\tnIter"
                 : "");
  return (n < 0 || (size_t)n >= cap) ? 0 : (size_t)n;
}

size_t ngsld_host_format_pair(char *buf, size_t cap, const char *label1, const char *label2, double dist,
                              const ngsld_rec_std *sr, const ngsld_rec_ext *er, double maf1, double maf2) {
  const size_t need = std::strlen(label1 ? label1 : "(null)") + std::strlen(label2 ? label2 : "(null)") + 1024;
  if (cap < need) return 0;
  return (size_t)(format_row(buf, label1, label2, dist, sr, er, maf1, maf2) - buf);
}

size_t ngsld_host_format_double(char *buf, size_t cap, double v, int decimals) {
  if (cap < 400 || (decimals != 6 && decimals != 0)) return 0;
  return (size_t)((decimals == 6 ? put_fixed<6>(buf, v) : put_fixed<0>(buf, v)) - buf);
}

int ngsld_host_write_batch(const ngsld_batch *b, const ngsld_pos *pos, const double *pos_dist, const double *maf,
                           int n_threads, int fd) try {
  if (b == nullptr || maf == nullptr) return NGSLD_ERR_INVALID;
  if (b->n_items == 0 || b->n_pairs == 0) return NGSLD_OK;
  if (n_threads < 1) n_threads = 1;
  if ((uint64_t)n_threads > b->n_items) n_threads = (int)b->n_items;
  size_t max_label = 6;  // "(null)"
  if (pos)
    for (const auto &l : pos->labels) max_label = std::max(max_label, l.size());
  const size_t row_bytes = 2 * max_label + 1024;
  // contiguous item ranges with equal pair counts; every thread formats into its own buffer
  std::vector<uint64_t> cut(n_threads + 1, b->n_items);
  cut[0] = 0;
  for (int t = 1; t < n_threads; ++t) {
    const uint64_t target = b->n_pairs * (uint64_t)t / (uint64_t)n_threads;
    uint64_t lo = cut[t - 1], hi = b->n_items;
    while (lo < hi) {  // first item whose first_record >= target
      const uint64_t mid = (lo + hi) / 2;
      if (b->items[mid].first_record < target) lo = mid + 1; else hi = mid;
    }
    cut[t] = lo;
  }
  // The threads' text buffers live across calls (a batch is ~1-2 GB of text: allocating, zero-filling and page-faulting
  // that afresh for every batch cost more than the formatting itself); they are plain malloc'd memory, grown on demand.
  static std::mutex cache_mutex;
  static std::vector<RawBuf> cache;
  std::lock_guard<std::mutex> cache_lock(cache_mutex);
  if (cache.size() < (size_t)n_threads) cache.resize((size_t)n_threads);
  std::vector<size_t> out_len((size_t)n_threads, 0);
  std::vector<int> out_err((size_t)n_threads, 0);
  auto work = [&](int t) {
    const uint64_t i0 = cut[t], i1 = cut[t + 1];
    if (i0 >= i1) return;
    const uint64_t rec_end = i1 < b->n_items ? b->items[i1].first_record : b->n_pairs;
    const uint64_t np = rec_end - b->items[i0].first_record;
    RawBuf &buf = cache[(size_t)t];
    if (!buf.reserve(np * (b->ext ? 200 : 96) + np * 2 * max_label + row_bytes, 0)) {
      out_err[(size_t)t] = 1;
      return;
    }
    char *p = buf.data();
    uint64_t cur_s1 = UINT64_MAX, cur_s2 = 0;
    double dist = 0;
    for (uint64_t i = i0; i < i1; ++i) {
      const ngsld_item &it = b->items[i];
      const uint64_t s1 = it.s1;
      if (s1 != cur_s1) {  // a new row: the reference's running sum starts over (ngsLD.cpp:233,241)
        cur_s1 = s1;
        cur_s2 = s1;
        dist = 0;
      }
      const char *l1 = pos ? pos->labels[s1].c_str() : nullptr;
      uint64_t k = it.first_record;
      for (uint32_t cc = 0; cc < it.count; ++cc) {
        const uint64_t s2 = (uint64_t)it.s2_begin + cc;
        while (cur_s2 < s2) {  // dist accumulates over every site passed, kept or not
          ++cur_s2;
          dist += pos_dist ? pos_dist[cur_s2] : INFINITY;
        }
        if (!((it.mask >> cc) & 1ull)) continue;
        if ((size_t)(buf.data() + buf.size() - p) < row_bytes) {  // extreme values printed long: grow
          const size_t used = (size_t)(p - buf.data());
          if (!buf.reserve(buf.size() * 2 + row_bytes, used)) {
            out_err[(size_t)t] = 1;
            return;
          }
          p = buf.data() + used;
        }
        p = format_row(p, l1, pos ? pos->labels[s2].c_str() : nullptr, dist, &b->std[k], b->ext ? &b->ext[k] : nullptr,
                       maf[s1], maf[s2]);
        ++k;
      }
    }
    out_len[(size_t)t] = (size_t)(p - buf.data());
  };
  std::vector<std::thread> th;
  th.reserve((size_t)n_threads);
  for (int t = 1; t < n_threads; ++t) {
    try {
      th.emplace_back(work, t);
    } catch (...) {  // no more threads to be had: this share is formatted by the caller
      work(t);
    }
  }
  work(0);
  for (auto &x : th) x.join();
  for (int t = 0; t < n_threads; ++t)
    if (out_err[(size_t)t]) return NGSLD_ERR_NOMEM;
  for (int t = 0; t < n_threads; ++t) {
    const char *q = cache[(size_t)t].data();
    size_t left = out_len[(size_t)t];
    while (left) {
      const ssize_t w = ::write(fd, q, left);
      if (w <= 0) return NGSLD_ERR_INVALID;
      q += w;
      left -= (size_t)w;
    }
  }
  return NGSLD_OK;
} catch (...) {
  return NGSLD_ERR_NOMEM;
}

}  // extern "C"
