#!/usr/bin/env bash
# Same-box A/B of the multi-wavefront kernels: LDS exchange as assembly + parked Pearson sums (default) against the exchange
# alone (noparked) and against the compiler's own LDS accesses (noasm).
A=$PWD/ngsld_amd/ab
for shape in ${SHAPES:-"--config_c3_--sites_12000" "--config_c4_--sites_60000" "--config_c3_--sites_12000_--ind_700"}; do
  shape=${shape//_/ }
  echo "== $shape"
  BENCH_ARGS="--no-cpu --no-sink --no-e2e $shape --steps 2 --warmup 1" ROUNDS=2 tools/ab.sh "default=X=1" "noparked=NGSLD_LIB=$A/libngsld_noparked.so" "noasm=NGSLD_LIB=$A/libngsld_noasm.so"
done
