#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of the pair kernel for the BASELINE configurations other than the bench default
# (one MI355X): appends "<config>,<kernel stats csv line>" to gpurun_out/configs_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/configs_kernel_stats.csv
echo '"config","Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"' > $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; label=$2; shift 2
  rm -rf /tmp/pc_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$tag -o run -- python $R/bench.py --no-cpu "$@" > /tmp/pc_$tag.log 2>&1
  grep -E "pair_ld" /tmp/pc_$tag/run_kernel_stats.csv | sed "s/^/\"$label\",/" >> $OUT
  grep -m1 '^{' /tmp/pc_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', '%.4g pairs/s' % d['value'], 'HIP-event ms per launch %.3f' % d['roofline']['kernel_ms_per_launch'])"
}
run c1 "configs[1] 5000x100 all pairs" --ind 100 --sites 5000 --max-kb 0 --steps 5 --warmup 2
run c3 "configs[3] 50000x1000 all pairs (all eight shards on one GPU)" --ind 1000 --sites 50000 --max-kb 0 --steps 1 --warmup 0
run c4 "configs[4] share 125000x2000 500kb" --ind 2000 --sites 125000 --max-kb 500 --max-gap 2000 --steps 1 --warmup 1
run hard "called genotypes 100000x500 100kb" --hard-calls --steps 3 --warmup 1
cat $OUT
