#!/usr/bin/env bash
# L2-miss traffic (FETCH_SIZE, KiB per launch) of the pair kernel for two library builds, same box.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  label=${v%%=*}; envs=${v#*=}
  rm -rf /tmp/pmc_$label
  env $envs rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_$label -o run -- python $R/bench.py --no-cpu --steps 1 --warmup 0 > /tmp/pmc_$label.log 2>&1
  python - /tmp/pmc_$label $label <<'PY'
import csv, glob, sys
tot = 0.0; n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pair_ld" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print(sys.argv[2], "FETCH_SIZE KiB per launch:", tot / max(n, 1), "launches", n)
PY
done
