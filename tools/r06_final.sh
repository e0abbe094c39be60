#!/usr/bin/env bash
# Round 6, last measurement batch, one box: the round-5 library (ngsld_amd/ab/libngsld_r05.so, built from the round-5 sources) against
# the tree on the un-called passes and the headline; the binary end to end; kernel trace + counters of the tree; the driver's line.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_final; rm -rf $O; mkdir -p $O
cd $R
ROUNDS=2 bash tools/ab_libs.sh "--mono-frac 0.2 --steps 3 --warmup 1" $O/ab_mono.txt r05 tree
ROUNDS=2 bash tools/ab_libs.sh "--sfs --steps 3 --warmup 1" $O/ab_sfs.txt r05 tree
ROUNDS=1 bash tools/ab_libs.sh "--steps 3 --warmup 1" $O/ab_headline.txt r05 tree
for v in grouped plain; do
  if [ $v = plain ]; then export NGSLD_TEST_TEXT_GROUPS=0; else unset NGSLD_TEST_TEXT_GROUPS; fi
  E2E_ONLY=default,mono20,sfs timeout 600 python tools/e2e_uncalled.py > $O/e2e_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/e2e_$v.json')); print('$v', {k:v['seconds'] for k,v in d['runs'].items()})" | tee -a $O/e2e.txt
done
unset NGSLD_TEST_TEXT_GROUPS
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_head -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered --no-other-configs > $O/trace_head.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_mono -o t -- python $R/bench.py --mono-frac 0.2 --steps 3 --warmup 1 --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered --no-other-configs > $O/trace_mono.json 2>/dev/null
cp $(find $O/trace_head -name "*kernel_stats.csv" | head -1) $O/kernel_stats_headline.csv
cp $(find $O/trace_mono -name "*kernel_stats.csv" | head -1) $O/kernel_stats_mono.csv
rm -rf $O/trace_head $O/trace_mono
bash $R/tools/pmc_replay.sh --no-unfiltered --no-other-configs > $O/pmc_replay.txt 2>&1
cd $R
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
head -4 $O/kernel_stats_mono.csv | cut -c1-160; cat $O/ab_mono.txt $O/ab_sfs.txt $O/ab_headline.txt $O/e2e.txt; grep -v "^wave" $O/pmc_replay.txt | head -30; cat $O/bench_default.time
