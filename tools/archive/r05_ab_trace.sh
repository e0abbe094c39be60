#!/bin/bash
# kernel trace of the un-called pass for two library builds (same box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$PWD/gpurun_out/ab_trace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "now=NGSLD_X=0" "old=NGSLD_LIB=$R/ngsld_amd/ab/libngsld_b19a26f.so"; do
  label=${v%%=*}; envs=${v#*=}
  env $envs rocprofv3 --kernel-trace --stats -d $O/$label -o t -- python $R/bench.py --mono-frac 0.2 --steps 2 --warmup 1 --no-cpu --no-e2e --no-traffic --no-sink --no-unfiltered --no-other-configs > $O/$label.log 2>&1
  DB=$(find $O/$label -name "*.db" | head -1)
  echo "== $label"; python $R/tools/rocpd_summary.py $DB | head -9
done
