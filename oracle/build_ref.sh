#!/usr/bin/env bash
# Build oracle/_ref/libngsld_ref.so from the reference's OWN sources, read where they lie under
# /root/reference (nothing is copied into the repo; the only output is the .so under oracle/_ref/).
#
# What can and cannot be built here
# ---------------------------------
# The reference binary needs GSL (Makefile:7,10 there; gsl_rng in gen_func.hpp:12, gsl_statistics in
# ngsLD.hpp:3).  GSL is not installed in this image and, per the rules of this build, NO stand-in
# header or library is written for it: ngsLD.cpp / parse_args.cpp AS FILES (main, pth_struct's gsl_rng,
# draw_rnd, pearson_r) are therefore UNBUILDABLE here.  What IS built of ngsLD.cpp: the GSL-free line
# ranges of calc_pair_LD, cut out by anchor and compiled verbatim (second half of this script) -- only
# pearson_r (gsl_stats_correlation) and the --rnd_sample draws stay unpinned (see ngsld_oracle.h).
#
# shared/gen_func.cpp and shared/read_data.cpp touch GSL in exactly three places, none on the
# hot path: the include (gen_func.hpp:12), the draw_rnd prototype (gen_func.hpp:47) and the
# 3-line draw_rnd definition (gen_func.cpp:117-119).  This script streams those four files through a
# filter that drops those lines -- and nothing else -- plus the two quoted self-includes, and compiles
# the stream together with oracle/ref_shim.cpp (extern "C" doors).  Every remaining function
# (haplo_freq, pair_freq_iter, est_maf, post_prob, logsum, conv_space, miss_data, read_geno,
# read_dist, read_file, transp_matrix, init_ptr/free_ptr ...) is the reference's text, verbatim.
set -euo pipefail
REF=${NGSLD_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
if [ ! -f "$REF/shared/gen_func.cpp" ]; then
  echo "build_ref.sh: $REF not present; skipping (prebuilt oracle/_ref is used if it exists)" >&2
  exit 0
fi
mkdir -p "$HERE/_ref"

drop_gsl_hpp() {   # header: no GSL include, no draw_rnd prototype, no '#pragma once' (main-file warning)
  sed -e '/#include <gsl\//d' -e '/draw_rnd(gsl_rng/d' -e '/#pragma once/d' -e '/#include "gen_func.hpp"/d' "$1"
}
drop_gsl_cpp() {   # source: no self-include, no draw_rnd definition (signature line .. first "^}")
  awk '
    /^double draw_rnd\(gsl_rng/ {skip=1}
    skip { if ($0 ~ /^}/) skip=0; next }
    /#include "gen_func.hpp"/ {next}
    /#include "read_data.hpp"/ {next}
    {print}
  ' "$1"
}

# ---- ngsLD.cpp: the GSL-free lines of calc_pair_LD, compiled from where they lie ------------------------------
# ngsLD.cpp as a whole cannot be built (gsl_rng in pth_struct, draw_rnd, gsl_stats_correlation).  Its in-tree
# ARITHMETIC can: the lines below are cut out of the file by ANCHOR (a regular expression that must match exactly one
# line -- if the reference's text ever changes this script fails instead of compiling something else) and wrapped,
# verbatim, into extern "C" functions whose locals carry the names those lines use (p->pars->..., s1, s2, dist, D, Dp,
# r2, hap_freq, n_ind_data, n_iter, r2pear).  `params` / `pth_struct` come from ngsLD.hpp itself, streamed with its GSL
# include and the gsl_rng member dropped (the same filter technique as gen_func.hpp above).
#   ref_walk_row      ngsLD.cpp:240-275  the s2 walk: running dist, max_kb / max_snp `break`s, maf[s1] `break`, maf[s2] skip
#   ref_pair_stats    ngsLD.cpp:296-306  hap-derived maf, D, D', r2      + :328-333  chi2 in float
#   ref_format_row    ngsLD.cpp:296-306 + :311-351  the two fprintf formats (standard + extended columns) and the newline
#   ref_print_header  ngsLD.cpp:77       the header line
# parse_args.cpp (init_pars, parse_cmd_args: the option table, the defaults, the argument echo, the validation messages) uses
# nothing of GSL but the header it includes: it is compiled WHOLE, with `version` taken from its one line of ngsLD.cpp
# (ref_parse_args in ref_shim.cpp calls it; tests/test_cli_args_vs_ref.py holds the drop-in binary's own parser to it).
CPP="$REF/ngsLD.cpp"
anchor() {  # anchor <regex>: the number of the ONE line of ngsLD.cpp it matches
  local hits
  hits=$(grep -nE -- "$1" "$CPP" | cut -d: -f1)
  if [ "$(echo "$hits" | wc -w)" != 1 ]; then
    echo "build_ref.sh: anchor /$1/ matches $(echo "$hits" | wc -w) lines of $CPP (expected exactly 1): the reference's text changed" >&2
    exit 1
  fi
  echo "$hits"
}
cut_lines() { echo "#line $1 \"reference:ngsLD.cpp\""; sed -n "$1,$2p" "$CPP"; }
W0=$(anchor '^  while \(s2 < p->pars->n_sites\)\{$')
W1=$(( $(anchor '^    // Random sampling$') - 1 ))
S0=$(anchor '^    double maf\[2\];$')
S1=$(anchor '^    r2 = pow\(D / sqrt\(maf\[0\] \* maf\[1\] \* \(1-maf\[0\]\) \* \(1-maf\[1\]\)\), 2\);$')
F0=$(( $(anchor '^    pthread_mutex_lock\(&printf_mutex\);$') + 1 ))
F1=$(( $(anchor '^    pthread_mutex_unlock\(&printf_mutex\);$') - 1 ))
C0=$(anchor '^      float chi2 = 0;$')
C1=$(anchor '^	    chi2 \+= pow\(hap_freq\[i\]-exp_hap_freq\[i\],2\)/exp_hap_freq\[i\];$')
H0=$(anchor '^  fprintf\(pars->out_fh, "site1\\tsite2\\tdist\\tr2_ExpG')
V0=$(anchor '^char const\* version = "[^"]*";$')
# the walk's last line must be the closing brace of the maf[s2] skip, the print block must end with the newline fprintf
sed -n "${W1}p" "$CPP" | grep -qE '^    \}$' || { echo "build_ref.sh: ngsLD.cpp:$W1 is not the end of the maf[s2] block" >&2; exit 1; }
sed -n "${F1}p" "$CPP" | grep -qE '^    fprintf\(p->pars->out_fh, "\\n"\);$' || { echo "build_ref.sh: ngsLD.cpp:$F1 is not the newline fprintf" >&2; exit 1; }
[ "$W0" -lt "$W1" ] && [ "$W1" -lt "$S0" ] && [ "$S0" -lt "$S1" ] && [ "$S1" -lt "$F0" ] && [ "$F0" -lt "$C0" ] && [ "$C0" -lt "$C1" ] && [ "$C1" -lt "$F1" ] || {
  echo "build_ref.sh: anchors of ngsLD.cpp out of order ($W0 $W1 $S0 $S1 $F0 $C0 $C1 $F1)" >&2; exit 1; }

# ---- main() and calc_pair_LD WHOLE, minus the statements that need GSL -----------------------------------------------------
# ngsLD.cpp from `int main` to the end of calc_pair_LD, compiled as it stands but for: the six statements that allocate, seed
# or free a gsl_rng (main: master stream, per-row streams; calc_pair_LD's free) -- dropped; the --rnd_sample block of
# calc_pair_LD (`// Random sampling` ... its closing brace: the draw is GSL's) -- dropped, so the door refuses rnd_sample < 1;
# the ONE call of pearson_r (gsl_stats_correlation) -- replaced by a lookup of the value the caller supplies for that pair
# (the oracle's r2_ExpG: this column stays unpinned, everything else of the run is the reference's text); `main` renamed
# ref_main.  shared/threadpool.c is compiled with it (as C++, like the reference's Makefile does).  The counts of what the
# filter touched are asserted.  ref_main(argc, argv) therefore IS the reference's program flow -- argument parsing, file
# checks, reader, call_geno loop, est_maf loop, exp / expected genotypes, positions and labels, thread pool, per-row walk, EM,
# statistics, fprintf -- on real files, writing the reference's TSV.
M0=$(anchor '^int main \(int argc, char\*\* argv\) \{$')
M1=$(( $(anchor '^double pearson_r \(double \*s1, double \*s2, uint64_t n_ind\)\{$') - 1 ))
main_and_calc() {
  sed -n "${M0},${M1}p" "$CPP" | awk '
    /^int main \(int argc, char\*\* argv\) \{$/ { print "extern \"C\" int ref_main (int argc, char** argv) {"; renamed++; next }
    /gsl_rng/ { rng++; next }
    /^    \/\/ Random sampling$/ { skip=1; blocks++ }
    skip { if ($0 ~ /^    \}$/) skip=0; next }
    /^    r2pear = pearson_r\(p->pars->expected_geno\[s1\], p->pars->expected_geno\[s2\], p->pars->n_ind\);$/ {
      print "    r2pear = ref_r2pear_lookup(s1, s2);"; pear++; next }
    { print }
    END { if (renamed != 1 || rng != 6 || blocks != 1 || pear != 1) { print "#error build_ref.sh: ngsLD.cpp changed (main " renamed ", gsl_rng lines " rng ", sampling blocks " blocks ", pearson_r calls " pear ")" } }
  '
}

drop_gsl_ngsld_hpp() {  # ngsLD.hpp: no GSL include, no gsl_rng member, no quoted includes (their text is already in the stream)
  sed -e '/#include <gsl\//d' -e '/gsl_rng\* rnd_gen;/d' -e '/#pragma once/d' -e '/#include "read_data.hpp"/d' \
      -e '/#include "threadpool.h"/d' "$1"
}

{
  echo '#line 1 "reference:shared/gen_func.hpp (GSL lines dropped)"'
  drop_gsl_hpp "$REF/shared/gen_func.hpp"
  echo '#line 1 "reference:shared/read_data.hpp"'
  drop_gsl_hpp "$REF/shared/read_data.hpp"
  echo '#line 1 "reference:shared/gen_func.cpp (draw_rnd dropped)"'
  drop_gsl_cpp "$REF/shared/gen_func.cpp"
  echo '#line 1 "reference:shared/read_data.cpp"'
  drop_gsl_cpp "$REF/shared/read_data.cpp"
  echo '#line 1 "reference:shared/threadpool.h"'
  sed -e '/#pragma once/d' "$REF/shared/threadpool.h"
  echo '#line 1 "reference:ngsLD.hpp (GSL include and gsl_rng member dropped)"'
  drop_gsl_ngsld_hpp "$REF/ngsLD.hpp"

  # -- the s2 walk of calc_pair_LD: lines W0..W1 verbatim, closed by the emit this door adds in place of the lines that
  #    follow them in the reference (random sampling, pearson_r, haplo_freq, print) and the reference's own `s2++; }`
  cat <<'CXX'
extern "C" uint64_t ref_walk_row(params *pars, uint64_t site, uint64_t *out_s2, double *out_dist, uint64_t cap) {
  pth_struct pth_local; pth_local.pars = pars; pth_local.site = site;
  pth_struct *p = &pth_local;
  uint64_t s1 = p->site;
  uint64_t s2 = s1 + 1;
  double dist = 0;
  uint64_t n_out = 0;
CXX
  cut_lines "$W0" "$W1"
  cat <<'CXX'
    if (n_out < cap) { out_s2[n_out] = s2; out_dist[n_out] = dist; }
    n_out++;
    s2++;
  }
  return n_out;
}
CXX
  # -- D / D' / r2 (+ the hap-derived maf) and the float chi2
  cat <<'CXX'
extern "C" void ref_pair_stats(const double *hap_in, double *out_D, double *out_Dp, double *out_r2, double *out_hap_maf, float *out_chi2) {
  double hap_freq[4] = {hap_in[0], hap_in[1], hap_in[2], hap_in[3]};
  double D, Dp, r2;
CXX
  cut_lines "$S0" "$S1"
  echo '  {'
  cut_lines "$C0" "$C1"
  cat <<'CXX'
    *out_chi2 = chi2;
  }
  *out_D = D; *out_Dp = Dp; *out_r2 = r2; out_hap_maf[0] = maf[0]; out_hap_maf[1] = maf[1];
}
CXX
  # -- one TSV row: the statistics again, then the reference's whole print block into a memory stream
  cat <<'CXX'
extern "C" long ref_format_row(char *buf, size_t cap, const char *label1, const char *label2, double dist_in, double r2pear,
                               const double *hap_in, uint64_t n_ind_data, double maf1, double maf2, uint64_t n_iter,
                               int extend_out) {
  params pars_local; memset(&pars_local, 0, sizeof(pars_local));
  char *labels_local[2] = {const_cast<char *>(label1), const_cast<char *>(label2)};
  double maf_local[2] = {maf1, maf2};
  pars_local.labels = labels_local; pars_local.maf = maf_local; pars_local.extend_out = extend_out != 0;
  pars_local.out_fh = fmemopen(buf, cap, "w");
  if (pars_local.out_fh == NULL) return -1;
  pth_struct pth_local; pth_local.pars = &pars_local; pth_local.site = 0;
  pth_struct *p = &pth_local;
  uint64_t s1 = 0, s2 = 1;
  double dist = dist_in;
  double hap_freq[4] = {hap_in[0], hap_in[1], hap_in[2], hap_in[3]};
  double D, Dp, r2;
CXX
  cut_lines "$S0" "$S1"
  cut_lines "$F0" "$F1"
  cat <<'CXX'
  long n = ftell(pars_local.out_fh);
  fclose(pars_local.out_fh);
  return n;
}
extern "C" long ref_print_header(char *buf, size_t cap, int extend_out) {
  params pars_local; memset(&pars_local, 0, sizeof(pars_local));
  params *pars = &pars_local;
  pars->extend_out = extend_out != 0;
  pars->out_fh = fmemopen(buf, cap, "w");
  if (pars->out_fh == NULL) return -1;
CXX
  cut_lines "$H0" "$H0"
  cat <<'CXX'
  long n = ftell(pars->out_fh);
  fclose(pars->out_fh);
  return n;
}
CXX
  # -- main() and calc_pair_LD whole (GSL statements removed, see above) + the thread pool
  echo 'extern "C" double ref_r2pear_lookup(uint64_t s1, uint64_t s2);'
  echo '#line 1 "reference:shared/threadpool.c"'
  sed -e '/#include "threadpool.h"/d' "$REF/shared/threadpool.c"
  echo "#line $M0 \"reference:ngsLD.cpp (main + calc_pair_LD, GSL statements removed)\""
  main_and_calc
  # -- the command line: parse_args.cpp whole (its include of ngsLD.hpp is already in the stream), `version` from ngsLD.cpp
  cut_lines "$V0" "$V0"
  echo '#line 1 "reference:parse_args.cpp"'
  sed -e '/#include "ngsLD.hpp"/d' "$REF/parse_args.cpp"
  echo '#line 1 "oracle/ref_shim.cpp"'
  cat "$HERE/ref_shim.cpp"
} | g++ -x c++ -O3 -w -fPIC -shared -ffp-contract=off -o "$HERE/_ref/libngsld_ref.so" - -lz -lpthread

echo "built $HERE/_ref/libngsld_ref.so (ngsLD.cpp lines $W0-$W1, $S0-$S1, $F0-$F1, $H0, $V0 and parse_args.cpp compiled in)"

# ---- the reference's main() with the library plugged in (the drop-in boundary, in situ) ------------------------------------
# INTEGRATION.md section 2 for real: ngsLD.cpp's main() from where it lies, its thread-pool section (`// Create thread pool`
# .. "cannot free thread pool!", ngsLD.cpp:153-198) replaced by the ONE line a maintainer would write --
# ngsld_compute_all(pars), integration/ngsld_binding.h -- and calc_pair_LD's print block (F0..F1, with the two hap-derived
# frequencies it prints, S0..S0+2) moved as it stands into print_pair(), which the binding's record sink calls.  Dropped besides:
# the three gsl_rng statements of main outside that section and the free of the pth array the section allocated.  Everything
# else -- parse_cmd_args, read_geno, call_geno, est_maf, the exp() loop, read_dist, labels, output file + header, the frees --
# is the reference's text, compiled and run.  Linked against the product's libngsld.so: built only where that exists.
# -> oracle/_ref/libngsld_ref_hip.so, entry ref_main_hip(argc, argv); tests/test_gpu_ref_main_patched.py.
LIBDIR=$(cd "$HERE/../ngsld_amd" && pwd)
if [ ! -f "$LIBDIR/libngsld.so" ]; then
  echo "build_ref.sh: $LIBDIR/libngsld.so not built yet; skipping the patched main (libngsld_ref_hip.so)" >&2
  exit 0
fi
sed -n "$((S0 + 1))p" "$CPP" | grep -qE '^    maf\[0\] = 1 - \(hap_freq\[0\] \+ hap_freq\[1\]\);$' || { echo "build_ref.sh: ngsLD.cpp:$((S0 + 1)) is not maf[0]" >&2; exit 1; }
sed -n "$((S0 + 2))p" "$CPP" | grep -qE '^    maf\[1\] = 1 - \(hap_freq\[0\] \+ hap_freq\[2\]\);$' || { echo "build_ref.sh: ngsLD.cpp:$((S0 + 2)) is not maf[1]" >&2; exit 1; }
main_patched() {
  sed -n "${M0},${M1}p" "$CPP" | awk '
    done_main { next }
    /^int main \(int argc, char\*\* argv\) \{$/ { print "extern \"C\" int ref_main_hip (int argc, char** argv) {"; renamed++; next }
    /^  \/\/ Create thread pool$/ { skip=1; pool++; print "  ngsld_compute_all(pars);   // integration/ngsld_binding.h, in place of ngsLD.cpp:153-198" }
    skip { if ($0 ~ /^    error\(__FUNCTION__, "cannot free thread pool!"\);$/) { skip=0; ended++ }; next }
    /gsl_rng/ { rng++; next }
    /^  free_ptr\(\(void\*\*\) pth\);/ { pth++; next }
    { print }
    /^}$/ { done_main=1 }
    END { if (renamed != 1 || pool != 1 || ended != 1 || rng != 3 || pth != 1 || !done_main) { print "#error build_ref.sh: ngsLD.cpp main changed (main " renamed ", pool " pool "/" ended ", gsl_rng lines " rng ", pth frees " pth ")" } }
  '
}
{
  echo '#line 1 "reference:shared/gen_func.hpp (GSL lines dropped)"'
  drop_gsl_hpp "$REF/shared/gen_func.hpp"
  echo '#line 1 "reference:shared/read_data.hpp"'
  drop_gsl_hpp "$REF/shared/read_data.hpp"
  echo '#line 1 "reference:shared/gen_func.cpp (draw_rnd dropped)"'
  drop_gsl_cpp "$REF/shared/gen_func.cpp"
  echo '#line 1 "reference:shared/read_data.cpp"'
  drop_gsl_cpp "$REF/shared/read_data.cpp"
  echo '#line 1 "reference:shared/threadpool.h"'
  sed -e '/#pragma once/d' "$REF/shared/threadpool.h"
  echo '#line 1 "reference:ngsLD.hpp (GSL include and gsl_rng member dropped)"'
  drop_gsl_ngsld_hpp "$REF/ngsLD.hpp"
  echo '#line 1 "integration/ngsld_binding.h"'
  sed -e '/#pragma once/d' "$HERE/../integration/ngsld_binding.h"
  cat <<'CXX'
void print_pair(params *pars_in, uint64_t s1, uint64_t s2, double dist, double r2pear, double D, double Dp, double r2,
                double *hap_freq, uint64_t n_ind_data, uint64_t n_iter) {
  pth_struct pth_local; pth_local.pars = pars_in; pth_local.site = s1;
  pth_struct *p = &pth_local;
CXX
  cut_lines "$S0" "$((S0 + 2))"
  cut_lines "$F0" "$F1"
  echo '}'
  echo "#line $M0 \"reference:ngsLD.cpp (main, thread-pool section replaced by ngsld_compute_all)\""
  main_patched
  cut_lines "$V0" "$V0"
  echo '#line 1 "reference:parse_args.cpp"'
  sed -e '/#include "ngsLD.hpp"/d' "$REF/parse_args.cpp"
} | g++ -x c++ -O2 -w -fPIC -shared -ffp-contract=off -I"$HERE/../include" -o "$HERE/_ref/libngsld_ref_hip.so" - \
      -L"$LIBDIR" -lngsld -Wl,-rpath,"$LIBDIR" -Wl,-rpath,/opt/rocm/lib -lz -lpthread
echo "built $HERE/_ref/libngsld_ref_hip.so (the reference's main with ngsLD.cpp:153-198 replaced by integration/ngsld_binding.h)"

