#!/usr/bin/env bash
# Round 6: the skip of degenerate sites' pairs on the other BASELINE shapes (un-called twins, 20 % monomorphic sites), same box.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_skip; mkdir -p $O
cd $R
B="--steps 3 --warmup 1 --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered --no-other-configs --mono-frac 0.2"
line() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ri=d["config"].get("replay_rank0_last_step") or {}
rr=d["config"]["rank_records"][0]
print(f"{d['ms_per_step']:.2f} ms  {d['value']:.4e} pairs/s  kernel {d['roofline']['kernel_ms_per_launch']:.2f} ms  checksum {rr.get('records_checksum_u64')}  flagged {ri.get('pairs_flagged')} host {ri.get('pairs_on_host')} degenerate sites {ri.get('sites_degenerate')}")
PY
}
for cfg in "--config c1" "--sites 12000 --ind 1000 --max-kb 0" "--sites 125000 --ind 2000 --max-kb 500"; do
  for v in on off; do
    if [ $v = off ]; then export NGSLD_REPLAY_SKIP=0; else unset NGSLD_REPLAY_SKIP; fi
    python bench.py $cfg $B > $O/cfg.json 2>$O/err.txt; echo "[$cfg] skip=$v $(line $O/cfg.json)" | tee -a $O/ab_configs.txt
  done
done
