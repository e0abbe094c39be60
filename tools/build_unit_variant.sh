#!/usr/bin/env bash
# A/B builds that differ in ONE HIP translation unit: tools/build_unit_variant.sh <name> <unit> <-D flags...>
#   -> ngsld_amd/ab/libngsld_<name>.so = csrc/build's objects with <unit>.o recompiled under the flags.  Use with NGSLD_LIB (tools/ab_libs.sh).
set -euo pipefail
name=$1; unit=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/ngsld_amd/csrc
mkdir -p $R/ngsld_amd/ab $C/build_ab
make -C $C -s all
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result "$@" -c $C/$unit.hip -o $C/build_ab/${unit}_$name.o
OBJS=""
for o in $C/build/*.o; do
  b=$(basename $o .o)
  if [ $b = cli_main ]; then continue; fi
  if [ $b = $unit ]; then OBJS="$OBJS $C/build_ab/${unit}_$name.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/ngsld_amd/ab/libngsld_$name.so $OBJS -lz -lpthread -ldl
echo "built ngsld_amd/ab/libngsld_$name.so ($unit: $*)"
