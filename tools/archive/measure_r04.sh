#!/usr/bin/env bash
# Round 4 -- measurement batch on the final tree (GPU box).  Output: gpurun_out/r04/final/
#   0. the GPU test suite, a fuzz soak over all kernel families
#   1. the driver's own bench line (python bench.py --gpus 1 --steps 20 --warmup 5: cpu_baseline, e2e, traffic measured in the run)
#   2. the other BASELINE configurations through bench.py on one MI355X (c3 and c4 at FULL size), the called-genotype pass
#   3. rocprofv3 evidence of the default bench: kernel stats + four counter passes + summary (profiles/collect_pmc.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04/final; mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_final.txt 2>&1
grep -E "passed|failed|parity:|c5 full size|multi-ranks" $O/pytest_gpu_final.txt | tail -6
timeout 900 python tools/fuzz_soak.py 400 4400 > $O/fuzz_soak_r04.txt 2>&1; tail -1 $O/fuzz_soak_r04.txt
timeout 600 python tools/fuzz_soak.py 10060 11060 > $O/fuzz_soak_r04_shapes.txt 2>&1; tail -1 $O/fuzz_soak_r04_shapes.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r04_final.json 2> $O/bench_r04_final.err
tail -c 400 $O/bench_r04_final.json
: > $O/configs_r04.jsonl
for C in "c1" "c3 --no-cpu --no-traffic" "c4 --no-cpu --no-traffic" "c2 --hard-calls --no-cpu --no-traffic --no-e2e"; do
  timeout 1500 python bench.py --config $C 2>> $O/bench_err.log | tail -1 >> $O/configs_r04.jsonl
done
python - $O/bench_r04_final.json $O/configs_r04.jsonl <<'PY'
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        if not l.startswith("{"): continue
        d = json.loads(l)
        print(d["config"]["workload"], "|", d["config"]["pairs_per_step"], "pairs |", f'{d["value"]:.4g} pairs/s |', f'{d["ms_per_step"]:.1f} ms |',
              "kernel", f'{d["roofline"]["kernel_ms_per_launch"]:.1f}', "| host-resident", f'{(d.get("value_host_resident") or 0):.4g}', "| iters", d["config"]["mean_executed_em_iterations"], "| frac",
              round(d["roofline"]["frac"], 3), "fp64", round(d["roofline"]["fp64_valu"]["frac"], 3), d["roofline"]["kernel"],
              "| traffic/pair", d["roofline"].get("traffic_per_pair"), "| e2e", (d.get("e2e_file_to_tsv_s") or {}).get("seconds"))
PY
timeout 900 bash profiles/collect_pmc.sh r04 --steps 2 --warmup 1 2>&1 | tail -25
mkdir -p $O/prof && cp gpurun_out/prof_r04/* $O/prof/ 2>/dev/null
