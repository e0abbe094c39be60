#!/usr/bin/env bash
# Same-box A/B: tiled workgroup order of the multi-wavefront kernel on long rows (default) against the plain item order.
for shape in ${SHAPES:-"--config_c3" "--config_c3_--sites_12000" "--config_c3_--sites_30000_--ind_2000"}; do
  shape=${shape//_/ }
  echo "== $shape"
  BENCH_ARGS="--no-cpu --no-sink --no-e2e $shape --steps 1 --warmup 0" ROUNDS=2 tools/ab.sh "tiles=X=1" "plain=NGSLD_TEST_TILES=0"
done
