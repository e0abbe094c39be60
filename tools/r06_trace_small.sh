#!/usr/bin/env bash
# kernel trace of the driver's unfiltered_input shape (10,000 x 500, 100 kb, 20 % monomorphic sites): where a 68 ms pass goes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_small; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --sites 10000 --mono-frac 0.2 --steps 5 --warmup 2 --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered --no-other-configs > $O/run.json 2> $O/run.err
python - <<PY
import csv,glob,json
f=glob.glob("$O/trace/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    if "at::" in r["Name"]: continue
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms")
d=json.loads(open("$O/run.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"], d["config"]["replay_off"])
PY
