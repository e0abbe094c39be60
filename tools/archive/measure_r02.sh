#!/usr/bin/env bash
# Round 2 measurement batch (GPU box): the BASELINE configurations through bench.py on one MI355X, the clocks / power
# the device reports while each kernel family runs, and the rocprofv3 evidence of the default bench.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
: > $O/configs_r02.jsonl; : > $O/clocks_r02.txt
for C in "c1" "c2 --no-cpu --no-e2e" "c3 --no-cpu" "c4 --no-cpu"; do
  ( python bench.py --config $C 2> $O/bench_err.log | tail -1 >> $O/configs_r02.jsonl ) &
  BP=$!
  echo "=== bench.py --config $C" >> $O/clocks_r02.txt
  sleep 12
  for k in 1 2 3 4 5 6; do
    kill -0 $BP 2>/dev/null || break
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr -s ' ' | head -4 >> $O/clocks_r02.txt
    echo "--" >> $O/clocks_r02.txt
    sleep 4
  done
  wait $BP
done
python - $O/configs_r02.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["config"]["workload"], "|", d["config"]["pairs_per_step"], "pairs |", f'{d["value"]:.4g} pairs/s |', f'{d["ms_per_step"]:.1f} ms |',
          "host-resident", f'{(d.get("value_host_resident") or 0):.4g}', "| iters", d["config"]["mean_executed_em_iterations"], "| frac",
          round(d["roofline"]["frac"], 3), "fp64", round(d["roofline"]["fp64_valu"]["frac"], 3), d["roofline"]["kernel"])
PY
bash profiles/collect_pmc.sh r02 --steps 2 --warmup 1 2>&1 | tail -15
