#!/usr/bin/env bash
# Kernel rate over cohort sizes (100 kb window over ~100 bp gaps), with and without --ignore_miss_data: pairs/s and
# pairs/s x n_ind, to spot a shape that falls off the trend.   tools/sweep_nind.sh > gpurun_out/sweep_nind.txt
for n in ${NINDS:-8 16 24 32 48 64 96 100 128 160 192 224 256 300 384 448 512 513 600 700 800 1000 1024 1025 1500 2000 2048 2049 3000 4096 4097}; do
  sites=$(python -c "print(int(max(4000, min(100000, 4e7 / $n))))")
  for m in "" "--ignore-miss"; do
    python bench.py --no-cpu --no-sink --no-e2e --no-traffic --config c2 --sites $sites --ind $n $m --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['value']
print('%5d %-13s sites %6d  %-28s %10.4g pairs/s  %8.3g ind-pairs/s  iters %.2f' % ($n, '$m' or 'all-individuals', $sites, d['roofline']['kernel'], v, v*$n, d['config']['mean_executed_em_iterations']))"
  done
done
