#!/usr/bin/env bash
# Round 2, second session -- measurement batch on the final kernels (GPU box): the BASELINE configurations through bench.py
# on one MI355X with the clocks / power the device reports meanwhile, the rocprofv3 evidence of the default bench
# (kernel stats + counter passes), and one counter comparison across the kernel families.   Output: gpurun_out/r02b/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02b; mkdir -p $O
cd $R
: > $O/configs_r02b.jsonl; : > $O/clocks_r02b.txt
for C in "c1" "c2" "c3 --no-cpu" "c4 --no-cpu"; do
  ( python bench.py --config $C 2> $O/bench_err.log | tail -1 >> $O/configs_r02b.jsonl ) &
  BP=$!
  echo "=== bench.py --config $C" >> $O/clocks_r02b.txt
  sleep 12
  for k in 1 2 3 4 5 6; do
    kill -0 $BP 2>/dev/null || break
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr -s ' ' | head -4 >> $O/clocks_r02b.txt
    echo "--" >> $O/clocks_r02b.txt
    sleep 4
  done
  wait $BP
done
sed -n 2p $O/configs_r02b.jsonl > $O/bench_r02b_final.json
python - $O/configs_r02b.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["config"]["workload"], "|", d["config"]["pairs_per_step"], "pairs |", f'{d["value"]:.4g} pairs/s |', f'{d["ms_per_step"]:.1f} ms |',
          "host-resident", f'{(d.get("value_host_resident") or 0):.4g}', "| iters", d["config"]["mean_executed_em_iterations"], "| frac",
          round(d["roofline"]["frac"], 3), "fp64", round(d["roofline"]["fp64_valu"]["frac"], 3), d["roofline"]["kernel"])
PY
bash profiles/collect_pmc.sh r02b --steps 2 --warmup 1 2>&1 | tail -15
PMC_OUT=r02b/pmc_families_r02b.txt bash tools/pmc_compare.sh " -- --config c1" " -- --config c3 --sites 25000 --steps 1 --warmup 0" " -- --config c4 --sites 60000" > /dev/null 2>&1
tail -60 $O/pmc_families_r02b.txt
