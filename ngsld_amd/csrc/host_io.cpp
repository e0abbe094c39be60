// host_io.cpp -- host-side readers and TSV writer of the drop-in (declared in include/ngsld_host.h).
// Pure C++17 + zlib, no device code: this is the part of the reference's L0/L4 layers
// (shared/read_data.cpp, shared/gen_func.cpp read_file, ngsLD.cpp fprintf) the new engine keeps on the host.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ngsld_host.h"

struct ngsld_pos {
  std::vector<double> pos_dist;
  std::vector<std::string> labels;
};

namespace {

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

// "%f" with the reference's text for the non-finite cases ("-nan", "inf", "-inf")
inline size_t put_f(char *p, size_t cap, double v) {
  if (std::isnan(v)) return (size_t)std::snprintf(p, cap, "-nan");
  return (size_t)std::snprintf(p, cap, "%f", v);
}

}  // namespace

extern "C" {

int ngsld_host_read_pos(const char *path, int header, uint64_t n_sites, ngsld_pos **out, char *err, size_t errlen) {
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
  if (lines.size() != n_sites) return set_err(err, errlen, "wrong number of lines in POS file!");

  // every line must have the same number of TAB-separated fields, at least 2 (read_data.cpp:139-147,180-181)
  size_t n_fields = 0;
  for (const auto &l : lines) {
    size_t nf = 1;
    for (char ch : l) nf += ch == '\t';
    if (n_fields == 0) n_fields = nf;
    if (nf != n_fields) return set_err(err, errlen, "invalid number of fields in file!");
  }
  if (n_fields < 2) return set_err(err, errlen, "wrong POS file format!");

  ngsld_pos *p = new ngsld_pos();
  p->pos_dist.resize(n_sites);
  std::string prev_chr;
  bool have_chr = false;
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
    if (!have_chr) {
      prev_chr = chr;
      have_chr = true;
    }
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
}

const double *ngsld_host_pos_dist(const ngsld_pos *p) { return p ? p->pos_dist.data() : nullptr; }
const char *ngsld_host_label(const ngsld_pos *p, uint64_t site) {
  return (p && site < p->labels.size()) ? p->labels[site].c_str() : nullptr;
}
void ngsld_host_free_pos(ngsld_pos *p) { delete p; }

int ngsld_host_geno_size_ok(uint64_t file_size, uint64_t n_ind, uint64_t n_sites) {
  if (n_ind == 0) return 0;
  return n_sites == file_size / sizeof(double) / n_ind / 3 ? 1 : 0;  // ngsLD.cpp:55
}

int ngsld_host_read_geno_bin(const char *path, uint64_t n_ind, uint64_t n_sites, double *out_raw, char *err,
                             size_t errlen) {
  if (path == nullptr || out_raw == nullptr) return set_err(err, errlen, "invalid argument");
  gzFile fh = std::strcmp(path, "-") == 0 ? gzdopen(0, "rb") : gzopen(path, "rb");
  if (fh == nullptr) return set_err(err, errlen, "cannot open GENO file!");
  gzbuffer(fh, 1 << 22);
  const uint64_t total = n_sites * n_ind * 3 * sizeof(double);
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
}

size_t ngsld_host_format_header(char *buf, size_t cap, int extend_out) {
  const int n = std::snprintf(
      buf, cap, "site1\tsite2\tdist\tr2_ExpG\tD\tDp\tr2%s\n",
      extend_out ? "\tsample_size\tmaf1\tmaf2\thap00\thap01\thap10\thap11\thap_maf1\thap_maf2\tchi2\tloglike\tnIter"
                 : "");
  return (n < 0 || (size_t)n >= cap) ? 0 : (size_t)n;
}

size_t ngsld_host_format_pair(char *buf, size_t cap, const char *label1, const char *label2, double dist,
                              const ngsld_rec_std *sr, const ngsld_rec_ext *er, double maf1, double maf2) {
  // glibc prints "(null)" for the reference's NULL labels when no --pos is given (ngsLD.cpp:135)
  if (label1 == nullptr) label1 = "(null)";
  if (label2 == nullptr) label2 = "(null)";
  const size_t need = std::strlen(label1) + std::strlen(label2) + 1024;
  if (cap < need) return 0;
  char *p = buf;
  p += std::snprintf(p, cap, "%s\t%s\t%.0f\t", label1, label2, dist);  // ngsLD.cpp:314-322
  p += put_f(p, 400, sr->r2_ExpG); *p++ = '\t';
  p += put_f(p, 400, sr->D);       *p++ = '\t';
  p += put_f(p, 400, sr->Dp);      *p++ = '\t';
  p += put_f(p, 400, sr->r2);
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
      chi2 = (float)((double)chi2 + std::pow(d, 2) / (double)exp_hap[i]);
    }
    p += std::snprintf(p, 64, "\t%lu\t", (unsigned long)er->n_ind_data);  // ngsLD.cpp:336-349
    p += put_f(p, 400, maf1); *p++ = '\t';
    p += put_f(p, 400, maf2); *p++ = '\t';
    for (int i = 0; i < 4; i++) {
      p += put_f(p, 400, h[i]);
      *p++ = '\t';
    }
    p += put_f(p, 400, hm0); *p++ = '\t';
    p += put_f(p, 400, hm1); *p++ = '\t';
    p += put_f(p, 400, (double)chi2);
    p += std::snprintf(p, 64, "\t%f\t%lu", 0.0, (unsigned long)er->n_iter);
  }
  *p++ = '\n';
  return (size_t)(p - buf);
}

}  // extern "C"
