#!/usr/bin/env bash
# Round 6, first call: the shipped tree's un-called pass (bench.py --mono-frac 0.2 / --sfs at 100,000 x 500):
# rates, kernel trace, counters of replay_lane_kernel.  Output under gpurun_out/r06_base/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_base; rm -rf $O; mkdir -p $O
cd $R
python bench.py --mono-frac 0.2 --steps 5 --warmup 2 --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered --no-other-configs > $O/bench_mono.json 2> $O/bench_mono.err
python bench.py --sfs --steps 5 --warmup 2 --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered --no-other-configs > $O/bench_sfs.json 2> $O/bench_sfs.err
python bench.py --steps 5 --warmup 2 --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered --no-other-configs > $O/bench_head.json 2> $O/bench_head.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --mono-frac 0.2 --steps 2 --warmup 1 --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered --no-other-configs > $O/trace.log 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
bash $R/tools/pmc_replay.sh --no-unfiltered --no-other-configs > $O/pmc_replay.txt 2>&1
rm -rf $O/trace/*/*.db
tail -3 $O/bench_mono.json $O/bench_sfs.json | cut -c1-600; head -12 $O/kernel_stats.csv; cat $O/pmc_replay.txt
