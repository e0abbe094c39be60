#!/usr/bin/env bash
# Round 6: the pairs of degenerate sites skip their EM in the pair kernel (NGSLD_REPLAY_SKIP=0: the tree without it), same box.
#   tools/r06_skip_ab.sh [rounds]   -> gpurun_out/r06_skip/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_skip; mkdir -p $O
cd $R
if [ -z "$NO_TESTS" ]; then
  python -m pytest tests/test_gpu_replay_lkl.py -x -q -k "degenerate or snp_called" 2>&1 | tail -5 | tee $O/pytest_skip.txt
fi
B="--steps 4 --warmup 2 --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered --no-other-configs"
line() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get("config",{})
ri=d.get("replay_rank0_last_step") or d.get("config",{}).get("replay_rank0_last_step") or {}
rr=d["config"]["rank_records"][0] if "rank_records" in d.get("config",{}) else {}
print(f"{d['ms_per_step']:.2f} ms  {d['value']:.4e} pairs/s  kernel {d['roofline']['kernel_ms_per_launch']:.2f} ms  checksum {rr.get('records_checksum_u64')}  {ri}")
PY
}
for round in $(seq 1 ${1:-2}); do
  for v in on off; do
    if [ $v = off ]; then export NGSLD_REPLAY_SKIP=0; else unset NGSLD_REPLAY_SKIP; fi
    python bench.py --mono-frac 0.2 $B > $O/mono_$v.json 2>$O/err.txt; echo "round $round mono skip=$v $(line $O/mono_$v.json)" | tee -a $O/ab.txt
    python bench.py --sfs $B > $O/sfs_$v.json 2>$O/err.txt; echo "round $round sfs  skip=$v $(line $O/sfs_$v.json)" | tee -a $O/ab.txt
  done
done
unset NGSLD_REPLAY_SKIP
python bench.py $B > $O/head.json 2>$O/err.txt; echo "headline $(line $O/head.json)" | tee -a $O/ab.txt
