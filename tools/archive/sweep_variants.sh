#!/usr/bin/env bash
# Kernel rate over cohort sizes for several kernel selections (env assignments), same box, interleaved:
#   NINDS="513 576 640" tools/sweep_variants.sh "default=" "ab=NGSLD_PAIR_KERNEL=ab" "multi=NGSLD_PAIR_KERNEL=multi"
# prints one line per (n_ind, mask, variant): pairs/s and ind-pairs/s.
for n in ${NINDS:-513 576 640 704 768 832 896 960 1024}; do
  sites=$(python -c "print(int(max(4000, min(100000, 4e7 / $n))))")
  for m in "" "--ignore-miss"; do
    for v in "$@"; do
      label=${v%%=*}; envs=${v#*=}
      env $envs python bench.py --no-cpu --no-sink --no-e2e --no-traffic --config c2 --sites $sites --ind $n $m --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); v=d['value']
    print('%5d %-15s %-8s %-34s %10.4g pairs/s  %8.3g ind-pairs/s  iters %.2f  chk %016x' % ($n, '$m' or 'all-individuals', '$label', d['roofline']['kernel'], v, v*$n, d['config']['mean_executed_em_iterations'], d['config']['rank_records'][0]['records_checksum_u64']))
except Exception as e:
    print('%5d %-15s %-8s FAILED %r' % ($n, '$m' or 'all-individuals', '$label', e))"
    done
  done
done
