#!/usr/bin/env bash
# One rocprofv3 --pmc pass over the bench's pair kernel: tools/pmc_once.sh "<counters>" [bench args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SET=$1; shift
rm -rf /tmp/pmc_once
rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_once -o run -- python $R/bench.py --no-cpu --no-traffic --no-e2e --steps 1 --warmup 0 "$@" > /tmp/pmc_once.log 2>&1
python - <<'PY'
import csv, glob
acc = {}
for f in glob.glob("/tmp/pmc_once/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pair_ld" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v)/len(v):.6g}")
PY
