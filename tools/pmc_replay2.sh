#!/usr/bin/env bash
# rocprofv3 counter passes, memory side, over the replay kernels of an un-called bench pass (as tools/pmc_replay.sh):
#   tools/pmc_replay2.sh [bench args]  -> per launch of replay_lane_kernel / pair_ld_run_kernel: L1 (TCP), TA and L2 (TCC) counters
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum"; do
  rm -rf /tmp/pmc_replay
  timeout 240 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_replay -o run -- python $R/bench.py --mono-frac 0.2 --no-cpu --no-sink --no-traffic --no-e2e --no-unfiltered --no-other-configs --steps 1 --warmup 1 "$@" > /tmp/pmc_replay.log 2>&1 || { echo "set [$SET] failed: $(tail -2 /tmp/pmc_replay.log | head -1 | cut -c1-200)"; continue; }
  python - <<'PY'
import csv, glob
acc = {}
for f in glob.glob("/tmp/pmc_replay/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "lane" if "replay_lane" in n else "pair" if "pair_ld" in n else None
        if k:
            acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:5s} {c:36s} per launch {sum(v)/len(v):.6g}  ({len(v)} launches)")
PY
done
