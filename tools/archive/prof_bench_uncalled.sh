#!/bin/bash
# rocprofv3 kernel trace of bench.py --mono-frac 0.2 (records on the device): which kernels the replay's share of a pass goes to
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r05_prof_bench; rm -rf $O; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o bench -- python $R/bench.py --mono-frac 0.2 --steps 2 --warmup 1 --no-cpu --no-sink --no-e2e --no-traffic > $O/run.log 2>&1
cd $R && python tools/rocpd_summary.py $(find $O -name "*.db" | head -1) > $O/summary.txt; head -24 $O/summary.txt
