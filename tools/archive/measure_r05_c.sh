#!/bin/bash
# round 5, third measurement: deep un-called data (ill-conditioned Pearson moments), the other BASELINE shapes un-called, the driver's own line
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_c; mkdir -p $O
COMMON="--no-cpu --no-sink --no-e2e --no-traffic"
timeout 900 python bench.py --mono-frac 0.2 --depth 30 --steps 2 --warmup 1 $COMMON > $O/bench_mono20_depth30.json 2> $O/err1.txt
timeout 900 python bench.py --config c1 --mono-frac 0.2 --steps 3 --warmup 1 $COMMON > $O/bench_c1_mono20.json 2> $O/err2.txt
timeout 900 python bench.py --config c3 --sites 12000 --mono-frac 0.2 --steps 1 --warmup 1 $COMMON > $O/bench_c3_12000_mono20.json 2> $O/err3.txt
timeout 900 python bench.py --config c4 --sites 125000 --mono-frac 0.2 --steps 1 --warmup 1 $COMMON > $O/bench_c4_125000_mono20.json 2> $O/err4.txt
timeout 1200 python bench.py > $O/bench_driver_line.json 2> $O/err5.txt
tail -n 3 $O/err*.txt
