#!/usr/bin/env bash
# One MI355X, the other BASELINE.json configs through bench.py (kernel rate, inputs resident): writes one JSON line each.
O=${1:-gpurun_out/configs.jsonl}; : > $O
python bench.py --no-cpu --no-traffic --no-e2e --ind 100 --sites 5000 --max-kb 0 --steps 5 --warmup 2 | tail -1 >> $O                 # configs[1]
python bench.py --no-cpu --no-traffic --no-e2e --steps 3 --warmup 1 | tail -1 >> $O                                                       # configs[2]
python bench.py --no-cpu --no-traffic --no-e2e --ind 1000 --sites 50000 --max-kb 0 --steps 1 --warmup 0 | tail -1 >> $O                 # configs[3], all 8 shards on one GPU
python bench.py --no-cpu --no-traffic --no-e2e --ind 2000 --sites 125000 --max-kb 500 --max-gap 2000 --steps 1 --warmup 1 | tail -1 >> $O # configs[4], one rank's share
python - $O <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["config"]["workload"], "|", d["config"]["pairs_per_step"], "pairs |", f'{d["value"]:.4g} pairs/s |', f'{d["ms_per_step"]:.1f} ms |',
          "iters", d["config"]["mean_executed_em_iterations"], "| frac", round(d["roofline"]["frac"], 3), d["roofline"]["kernel"])
PY
