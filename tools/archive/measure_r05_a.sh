#!/bin/bash
# round 5: the device-side replay of likelihood matrices on input that is not SNP-called
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_a; mkdir -p $O
COMMON="--no-cpu --no-sink --no-e2e --no-traffic"
NGSLD_TRACE=1 timeout 900 python bench.py --mono-frac 0.2 --steps 3 --warmup 1 $COMMON > $O/bench_mono20_full.json 2> $O/bench_mono20_full.err
NGSLD_TRACE=1 timeout 900 python bench.py --sfs --steps 3 --warmup 1 $COMMON > $O/bench_sfs_full.json 2> $O/bench_sfs_full.err
python bench.py --steps 3 --warmup 1 $COMMON > $O/bench_default.json 2> $O/bench_default.err
grep -h "trace\] finish\|trace\] exact" $O/*.err | head -40
