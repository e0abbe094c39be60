#!/bin/bash
# same-box A/B of two library builds on the un-called pass: value and ms per step (tools/ab.sh prints the pair kernel's time only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in 1 2 3; do for v in "now=NGSLD_X=0" "old=NGSLD_LIB=$PWD/ngsld_amd/ab/libngsld_b19a26f.so"; do
  label=${v%%=*}; envs=${v#*=}
  env $envs python bench.py ${BENCH_ARGS:---mono-frac 0.2} --steps 3 --warmup 1 --no-cpu --no-e2e --no-traffic --no-sink --no-unfiltered 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('round $r $label', d['value'], d['ms_per_step'], d['config']['replay_rank0_last_step']['pairs_on_device'])"
done; done
