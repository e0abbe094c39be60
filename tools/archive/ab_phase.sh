#!/usr/bin/env bash
# round 5 (review item 5): configs[4]'s shape, one rank's share -- the two workgroups of a CU started out of phase (NGSLD_PHASE_DELAY x 64 cycles
# for the workgroup in the odd wavefront slot), same box, interleaved.  Build the variants first: tools/build_variant.sh phase<N> -DNGSLD_PHASE_DELAY=<N>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$PWD/ngsld_amd/ab
ROUNDS=3 BENCH_ARGS="--config c4 --sites 125000 --steps 1 --warmup 1 --no-cpu --no-traffic --no-sink --no-e2e" bash tools/ab.sh \
  "base=NGSLD_X=0" "delay10=NGSLD_LIB=$A/libngsld_phase10.so" "delay20=NGSLD_LIB=$A/libngsld_phase20.so" "delay40=NGSLD_LIB=$A/libngsld_phase40.so"
