#!/usr/bin/env bash
# Round 3 -- measurement batch on the final kernels (GPU box).  Output: gpurun_out/r03/final/
#   1. the driver's own bench line (python bench.py --gpus 1 --steps 20 --warmup 5: cpu_baseline, e2e, traffic measured in the run)
#   2. the BASELINE configurations through bench.py on one MI355X (c3 and c4 at FULL size)
#   3. rocprofv3 evidence of the default bench: kernel stats + four counter passes + summary (profiles/collect_pmc.sh)
#   4. counters of the other kernel families (group: c1, two wavefronts: c3, four wavefronts: c4)
#   5. cohort sizes 8..5,200 with and without --ignore_miss_data
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03/final; mkdir -p $O
cd $R
#   0. the GPU test suite and two fuzz soaks (the cohort sizes whose kernel changed this round; the round-1 generator)
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_final.txt 2>&1
grep -E "passed|failed|parity:|c5 full size|multi-ranks" $O/pytest_gpu_final.txt | tail -6
timeout 900 python tools/fuzz_soak.py 10060 12060 > $O/fuzz_soak_r03_newshapes.txt 2>&1; tail -1 $O/fuzz_soak_r03_newshapes.txt
timeout 900 python tools/fuzz_soak.py 400 8400 > $O/fuzz_soak_r03.txt 2>&1; tail -1 $O/fuzz_soak_r03.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r03_final.json 2> $O/bench_r03_final.err
tail -c 400 $O/bench_r03_final.json
: > $O/configs_r03.jsonl
for C in "c1" "c3 --no-cpu --no-traffic" "c4 --no-cpu --no-traffic"; do
  timeout 1500 python bench.py --config $C 2>> $O/bench_err.log | tail -1 >> $O/configs_r03.jsonl
done
python - $O/bench_r03_final.json $O/configs_r03.jsonl <<'PY'
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        if not l.startswith("{"): continue
        d = json.loads(l)
        print(d["config"]["workload"], "|", d["config"]["pairs_per_step"], "pairs |", f'{d["value"]:.4g} pairs/s |', f'{d["ms_per_step"]:.1f} ms |',
              "host-resident", f'{(d.get("value_host_resident") or 0):.4g}', "| iters", d["config"]["mean_executed_em_iterations"], "| frac",
              round(d["roofline"]["frac"], 3), "fp64", round(d["roofline"]["fp64_valu"]["frac"], 3), d["roofline"]["kernel"],
              "| traffic/pair", d["roofline"].get("traffic_per_pair"))
PY
timeout 900 bash profiles/collect_pmc.sh r03 --steps 2 --warmup 1 2>&1 | tail -25
mkdir -p $O/prof && cp gpurun_out/prof_r03/* $O/prof/ 2>/dev/null
PMC_OUT=r03/final/pmc_families_r03.txt timeout 1200 bash tools/pmc_compare.sh " -- --config c1" " -- --config c3 --sites 25000 --steps 1 --warmup 0" " -- --config c4 --sites 60000" " -- --ind 640 --sites 60000" " -- --ind 768 --sites 50000" > /dev/null 2>&1
tail -80 $O/pmc_families_r03.txt
NINDS="8 16 24 32 48 64 96 100 128 160 192 224 256 300 384 448 512 513 576 577 640 641 700 768 832 833 896 1000 1024 1025 1100 1152 1153 1280 1281 1500 2000 2048 2049 2304 2305 2560 2561 3000 4096 4097 4608 4609 5120 5121 6000" timeout 1800 bash tools/sweep_nind.sh > $O/sweep_nind_r03.txt 2>&1
cat $O/sweep_nind_r03.txt
