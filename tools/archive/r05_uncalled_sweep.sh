#!/bin/bash
# input that is NOT SNP-called beyond the two cases of the review: depth 2 (the survey's "stress" variant), half the sites monomorphic,
# --ignore_miss_data, the large cohorts -- is there a cliff left anywhere?  One line per case: pass rate, pairs flagged / on device / on host.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="--steps 2 --warmup 1 --no-cpu --no-e2e --no-traffic --no-sink --no-unfiltered --no-other-configs"
run() { echo "== $*"; python bench.py $B "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=c.get('replay_rank0_last_step',{})
print(f\"{d['value']:.4g} pairs/s, {d['ms_per_step']:.1f} ms/step, first pass {c.get('first_pass_s')}, flagged {r.get('pairs_flagged')} device {r.get('pairs_on_device')} host {r.get('pairs_on_host')} store {r.get('exact_store')} built in {r.get('exact_store_build_s')}\")"; }
run --mono-frac 0.2 --depth 2
run --sfs --depth 2
run --mono-frac 0.5
run --mono-frac 0.2 --ignore-miss
run --mono-frac 0.2 --depth 1 --ignore-miss
run --mono-frac 0.2 --ind 100 --sites 200000
run --mono-frac 0.2 --ind 1000 --sites 50000
run --mono-frac 0.2 --ind 2000 --sites 50000 --max-kb 200
run --mono-frac 0.2 --ind 64 --sites 200000
run --mono-frac 0.2 --ind 37 --sites 200000
