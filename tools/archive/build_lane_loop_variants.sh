#!/bin/bash
# Builds the library once per closed candidate of the lane-per-pair replay's individual loop (tools/experiments/lane_loop_*.patch,
# patches against ngsld_amd/csrc/ld_replay_lkl.hip) into ngsld_amd/ab/ -- what tools/r05_lane_loop_ab.sh compares on one box:
#   libngsld_rcp.so  the tree as it stands (the baseline's name in that script)
#   libngsld_v1.so   two register sets used in turn            (lane_loop_two_sets.patch)
#   libngsld_v2.so   one set, refetched between the two phases  (lane_loop_refetch_in_place.patch)
#   libngsld_v3.so   no per-individual test-and-branch          (lane_loop_branch_free.patch)
# Runs here (no GPU needed: hipcc cross-compiles); the source file is put back as it was, the tree's own library rebuilt last.
# Then:  gpurun -- 'CANDIDATES="v1 v2 v3" bash tools/r05_lane_loop_ab.sh'
set -e
cd "$(dirname "$0")/.."
SRC=ngsld_amd/csrc/ld_replay_lkl.hip
mkdir -p ngsld_amd/ab
cp $SRC /tmp/ld_replay_lkl.hip.keep
trap 'cp /tmp/ld_replay_lkl.hip.keep '"$SRC"'; make -s -C ngsld_amd/csrc -j8' EXIT
make -s -C ngsld_amd/csrc -j8 && cp ngsld_amd/libngsld.so ngsld_amd/ab/libngsld_rcp.so
for v in v1:two_sets v2:refetch_in_place v3:branch_free; do
  cp /tmp/ld_replay_lkl.hip.keep $SRC
  patch -s $SRC < tools/experiments/lane_loop_${v#*:}.patch
  make -s -C ngsld_amd/csrc -j8
  cp ngsld_amd/libngsld.so ngsld_amd/ab/libngsld_${v%%:*}.so
done
sha256sum ngsld_amd/ab/libngsld_*.so | cut -c1-16,64-
