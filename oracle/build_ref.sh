#!/usr/bin/env bash
# Build oracle/_ref/libngsld_ref.so from the reference's OWN sources, read where they lie under
# /root/reference (nothing is copied into the repo; the only output is the .so under oracle/_ref/).
#
# What can and cannot be built here
# ---------------------------------
# The reference binary needs GSL (Makefile:7,10 there; gsl_rng in gen_func.hpp:12, gsl_statistics in
# ngsLD.hpp:3).  GSL is not installed in this image and, per the rules of this build, NO stand-in
# header or library is written for it: ngsLD.cpp / parse_args.cpp (main, calc_pair_LD, pearson_r)
# are therefore UNBUILDABLE here and stay unpinned (see ngsld_oracle.h).
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

{
  echo '#line 1 "reference:shared/gen_func.hpp (GSL lines dropped)"'
  drop_gsl_hpp "$REF/shared/gen_func.hpp"
  echo '#line 1 "reference:shared/read_data.hpp"'
  drop_gsl_hpp "$REF/shared/read_data.hpp"
  echo '#line 1 "reference:shared/gen_func.cpp (draw_rnd dropped)"'
  drop_gsl_cpp "$REF/shared/gen_func.cpp"
  echo '#line 1 "reference:shared/read_data.cpp"'
  drop_gsl_cpp "$REF/shared/read_data.cpp"
  echo '#line 1 "oracle/ref_shim.cpp"'
  cat "$HERE/ref_shim.cpp"
} | g++ -x c++ -O3 -w -fPIC -shared -ffp-contract=off -o "$HERE/_ref/libngsld_ref.so" - -lz -lpthread

echo "built $HERE/_ref/libngsld_ref.so"
