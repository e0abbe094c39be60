#!/usr/bin/env bash
# Items per run (one workgroup each) on small problems: does a finer cut shorten the tail?  NGSLD_RUN_LEN = 16 (default), 8, 4, 2.
for shape in ${SHAPES:-"--config_c1" "--config_c1_--sites_12000" "--config_c2_--sites_20000"}; do
  shape=${shape//_/ }
  echo "== $shape"
  BENCH_ARGS="--no-cpu --no-sink --no-e2e $shape --steps 5 --warmup 2" ROUNDS=2 tools/ab.sh "len16=X=1" "len8=NGSLD_RUN_LEN=8" "len4=NGSLD_RUN_LEN=4" "len2=NGSLD_RUN_LEN=2"
done
