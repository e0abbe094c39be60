#!/usr/bin/env bash
# rocprofv3 counter passes over the replay kernels of an un-called bench pass (bench.py --mono-frac 0.2, one step after the store's build):
#   tools/pmc_replay.sh [bench args]   ->  counters per launch of replay_lane_kernel / replay_lkl_kernel / pair_ld_run_kernel
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rm -rf /tmp/pmc_replay
  timeout 240 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_replay -o run -- python $R/bench.py --mono-frac 0.2 --no-cpu --no-sink --no-traffic --no-e2e --steps 1 --warmup 1 "$@" > /tmp/pmc_replay.log 2>&1
  python - <<'PY'
import csv, glob
acc = {}
for f in glob.glob("/tmp/pmc_replay/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "lane" if "replay_lane" in n else "wave" if "replay_lkl" in n else "pair" if "pair_ld" in n else None
        if k:
            acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:5s} {c:32s} per launch {sum(v)/len(v):.6g}  ({len(v)} launches)")
PY
done
