#!/usr/bin/env bash
# Same-box A/B: the third value of the per-iteration reduction joined by a row broadcast (default) against folded with itself.
A=$PWD/ngsld_amd/ab
for shape in ${SHAPES:-"--config_c2" "--config_c3_--sites_12000" "--config_c4_--sites_60000"}; do
  shape=${shape//_/ }
  echo "== $shape"
  BENCH_ARGS="--no-cpu --no-sink --no-e2e $shape --steps 2 --warmup 1" ROUNDS=3 tools/ab.sh "bcast=X=1" "fold=NGSLD_LIB=$A/libngsld_foldt3.so"
done
