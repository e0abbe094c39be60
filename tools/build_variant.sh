#!/usr/bin/env bash
# Same-source A/B builds of the C-ABI library: tools/build_variant.sh <name> <-D flags...>  ->  ngsld_amd/ab/libngsld_<name>.so
# (every HIP translation unit is recompiled with the flags; the host objects are taken from csrc/build).
# Run with NGSLD_LIB=$PWD/ngsld_amd/ab/libngsld_<name>.so (tools/ab.sh).
set -euo pipefail
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/ngsld_amd/csrc
B=$C/build_$name
mkdir -p $B $R/ngsld_amd/ab
make -C $C -s all
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result"
pids=()
UNITS="ld_pair_w1 ld_pair_wn ld_pair_ab ld_pair_stream ld_pair_hard ld_prep ld_text ld_replay ld_replay_lkl engine engine_plan engine_run engine_replay multi"
for u in $UNITS; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $C/$u.hip -o $B/$u.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for u in $UNITS; do OBJS="$OBJS $B/$u.o"; done
for u in host_io stream replay gz_out; do OBJS="$OBJS $C/build/$u.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/ngsld_amd/ab/libngsld_$name.so $OBJS -lz -lpthread -ldl
rm -rf $B
echo "built ngsld_amd/ab/libngsld_$name.so ($*)"
