#!/usr/bin/env bash
# Round 6: the lane store with its rare sites first (NGSLD_REPLAY_PERM=0: in site order), and the sort key's tiling on top of it; same box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_lane; mkdir -p $O
LIB=${LIB:-u2}
one() {
  NGSLD_LIB=$PWD/ngsld_amd/ab/libngsld_$LIB.so python bench.py $ARGS --steps 3 --warmup 1 --no-cpu --no-e2e --no-traffic --no-sink --no-unfiltered --no-other-configs 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['config'].get('replay_rank0_last_step',{}); print('%.2f %.5g kernel %.2f dev %s host %s %s' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_launch'], r.get('pairs_on_device'), r.get('pairs_on_host'), d['config']['rank_records'][0]['records_checksum_u64']))"
}
for r in 1 2; do
  for ARGS in "--mono-frac 0.2" "--sfs"; do
    export ARGS
    NGSLD_REPLAY_PERM=0 bash -c "true"; export NGSLD_REPLAY_PERM=0; unset NGSLD_REPLAY_TILE; echo "round $r [$ARGS] perm=0 tile=5,3 $(one)" | tee -a $O/perm_ab.txt
    unset NGSLD_REPLAY_PERM
    for t in 5,3 3,3 4,3 2,4 3,4; do export NGSLD_REPLAY_TILE=$t; echo "round $r [$ARGS] perm=1 tile=$t $(one)" | tee -a $O/perm_ab.txt; done
    unset NGSLD_REPLAY_TILE
  done
done
