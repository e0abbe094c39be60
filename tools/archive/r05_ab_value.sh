#!/bin/bash
# same-box A/B of two library builds (value and ms per step of bench.py; tools/ab.sh prints the pair kernel's time only):
#   tools/r05_ab_value.sh "<bench args>" ...     old = ngsld_amd/ab/libngsld_<commit>.so (OLD_LIB), built from a worktree of that commit
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OLD=${OLD_LIB:-$PWD/ngsld_amd/ab/libngsld_b19a26f.so}
[ $# -eq 0 ] && set -- "--mono-frac 0.2"
for args in "$@"; do
  echo "== $args"
  for r in 1 2; do for v in "now=NGSLD_X=0" "old=NGSLD_LIB=$OLD"; do
    label=${v%%=*}; envs=${v#*=}
    env $envs python bench.py $args --steps 3 --warmup 1 --no-cpu --no-e2e --no-traffic --no-sink --no-unfiltered --no-other-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['config'].get('replay_rank0_last_step',{}); print('round $r $label', '%.4g' % d['value'], '%.1f ms' % d['ms_per_step'], r.get('pairs_on_device'), r.get('pairs_on_host'))"
  done; done
done
