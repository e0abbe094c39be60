#!/usr/bin/env bash
# Same-box sweep: candidates per work item of the multi-wavefront kernel (one workgroup per item).
for shape in "--config c3 --sites 12000" "--config c4 --sites 60000"; do
  for ppi in 16 32 64 16 32 64; do
    out=$(python bench.py --no-cpu --no-sink --no-e2e $shape --steps 2 --warmup 1 --pairs-per-item $ppi 2>&1 | tail -1)
    ms=$(echo "$out" | grep -o '"kernel_ms_per_launch": [0-9.]*' | grep -o '[0-9.]*$')
    echo "$shape  pairs_per_item=$ppi  kernel_ms=$ms"
  done
done
