#!/usr/bin/env bash
# Same-box A/B of library builds on one bench.py workload (on the GPU box):
#   ROUNDS=2 tools/ab_libs.sh "<bench.py args>" <out.txt> <lib name>...    (names under ngsld_amd/ab/libngsld_<name>.so; "tree" = ngsld_amd/libngsld.so)
# per line: round, library, ms a step, pairs/s, kernel ms, pairs replayed on the device / host, records checksum
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ARGS=$1; OUT=$2; shift 2
mkdir -p $(dirname $OUT)
one() {
  local lib=$PWD/ngsld_amd/ab/libngsld_$1.so; [ $1 = tree ] && lib=$PWD/ngsld_amd/libngsld.so
  NGSLD_LIB=$lib python bench.py $ARGS --no-cpu --no-e2e --no-traffic --no-sink --no-unfiltered --no-other-configs 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['config'].get('replay_rank0_last_step',{}); print('%.2f %.5g kernel %.2f dev %s host %s %s' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_launch'], r.get('pairs_on_device'), r.get('pairs_on_host'), d['config']['rank_records'][0]['records_checksum_u64']))"
}
for r in $(seq 1 ${ROUNDS:-2}); do for v in "$@"; do echo "round $r $v $(one $v)" | tee -a $OUT; done; done
