#!/usr/bin/env bash
# Same-box A/B of library builds: tools/ab.sh "<label>=<env assignments>" ...   (each label run ROUNDS times, interleaved)
ROUNDS=${ROUNDS:-2}
ARGS=${BENCH_ARGS:---no-cpu --no-traffic}
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    label=${v%%=*}; envs=${v#*=}
    out=$(env $envs python bench.py $ARGS 2>&1 | tail -1)
    ms=$(echo "$out" | grep -o '"kernel_ms_per_launch": [0-9.]*' | grep -o '[0-9.]*$')
    val=$(echo "$out" | grep -o '"value": [0-9.]*' | grep -o '[0-9.]*$')
    echo "round $r  $label  kernel_ms=$ms  pairs/s=$val"
  done
done
