#!/usr/bin/env bash
# Round 3, after the multi-wavefront kernel got its row slice in registers and streamed --call_geno runs their kernel:
# the part of tools/measure_r03.sh those changes touch, on the final tree.  Output: gpurun_out/r03/late/
#   0. the GPU test suite, a fuzz soak over the cohort sizes whose kernel changed
#   1. the driver's own bench line (unchanged kernel: a second box for the spread)
#   2. configs[3] / configs[4] at FULL size through bench.py
#   3. counters of the two- and four-wavefront shapes (VALU busy after the change)
#   4. the cohort sizes the multi-wavefront kernel serves, with and without --ignore_miss_data
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03/late; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_final.txt 2>&1
grep -E "passed|failed|parity:|c5 full size|multi-ranks" $O/pytest_gpu_final.txt | tail -6
timeout 600 python tools/fuzz_soak.py 10060 11560 > $O/fuzz_soak_late.txt 2>&1; tail -1 $O/fuzz_soak_late.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r03_late.json 2> $O/bench_r03_late.err
tail -c 300 $O/bench_r03_late.json
: > $O/configs_late.jsonl
for C in "c3 --no-cpu --no-traffic" "c4 --no-cpu --no-traffic"; do
  timeout 900 python bench.py --config $C 2>> $O/bench_err.log | tail -1 >> $O/configs_late.jsonl
done
python - $O/configs_late.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if not l.startswith("{"): continue
    d = json.loads(l)
    print(d["config"]["workload"], "|", d["config"]["pairs_per_step"], "pairs |", f'{d["value"]:.4g} pairs/s |', f'{d["ms_per_step"]:.1f} ms |',
          "frac", round(d["roofline"]["frac"], 3), "fp64", round(d["roofline"]["fp64_valu"]["frac"], 3), d["roofline"]["kernel"])
PY
PMC_OUT=r03/late/pmc_multi_late.txt timeout 900 bash tools/pmc_compare.sh " -- --config c3 --sites 25000 --steps 1 --warmup 0" " -- --config c4 --sites 60000" " -- --ind 1500 --sites 30000" > /dev/null 2>&1
tail -60 $O/pmc_multi_late.txt
NINDS="961 1000 1024 1281 1400 1536 1537 1700 2000 2048 2561 2800 3072 3073 3300 3600 4096" timeout 1200 bash tools/sweep_nind.sh > $O/sweep_nind_late.txt 2>&1
cat $O/sweep_nind_late.txt
