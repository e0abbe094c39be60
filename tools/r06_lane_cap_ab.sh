#!/usr/bin/env bash
# Round 6: a step cap on the lanes of LONG launches too (the pairs beyond it go to the wavefront-per-pair kernel behind) -- does
# the lane kernel's tail (one 38-step pair alone on a SIMD) cost what the cap saves?  Same box, NGSLD_TEST_LANE_ITER_CAP.
# (needs an experiment build: in the tree the knob acts on launches below 2^22 pairs only -- engine_replay.hip, device_replay_lkl)
#   tools/r06_lane_cap_ab.sh [rounds]   -> gpurun_out/r06_lane_cap/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_lane_cap; mkdir -p $O
cd $R
B="--steps 5 --warmup 2 --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered --no-other-configs"
line() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ri=d.get("replay_rank0_last_step") or d.get("config",{}).get("replay_rank0_last_step") or {}
rr=d["config"]["rank_records"][0] if "rank_records" in d.get("config",{}) else {}
print(f"{d['ms_per_step']:.2f} ms  {d['value']:.4e} pairs/s  checksum {rr.get('records_checksum_u64')}  dev {ri.get('pairs_on_device')} host {ri.get('pairs_on_host')}")
PY
}
for round in $(seq 1 ${1:-2}); do
  for cap in 0 12 16 24; do
    export NGSLD_TEST_LANE_ITER_CAP=$cap
    python bench.py --sites 10000 --mono-frac 0.2 $B > $O/s_$cap.json 2>$O/err.txt; echo "round $round 10000 mono cap=$cap $(line $O/s_$cap.json)" | tee -a $O/ab.txt
    python bench.py --sites 10000 --sfs $B > $O/ss_$cap.json 2>$O/err.txt; echo "round $round 10000 sfs  cap=$cap $(line $O/ss_$cap.json)" | tee -a $O/ab.txt
    python bench.py --mono-frac 0.2 $B > $O/b_$cap.json 2>$O/err.txt; echo "round $round 100000 mono cap=$cap $(line $O/b_$cap.json)" | tee -a $O/ab.txt
  done
done
unset NGSLD_TEST_LANE_ITER_CAP
