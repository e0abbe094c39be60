#!/usr/bin/env bash
# Same-box sweep: rows per tile of the multi-wavefront kernel's tiled order (NGSLD_TILE_ROWS), configs[3] at full size.
for shape in ${SHAPES:-"--config_c3"}; do
  shape=${shape//_/ }
  echo "== $shape"
  BENCH_ARGS="--no-cpu --no-sink --no-e2e $shape --steps 1 --warmup 0" ROUNDS=1 tools/ab.sh "rows64=X=1" "rows16=NGSLD_TILE_ROWS=16" "rows32=NGSLD_TILE_ROWS=32" "rows128=NGSLD_TILE_ROWS=128" "rows256=NGSLD_TILE_ROWS=256"
done
