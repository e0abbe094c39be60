#!/bin/bash
# round 5, "before": what the tree of round 4 does on matrices that are not SNP-called (README.md:73)
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_before; mkdir -p $O
COMMON="--no-cpu --no-sink --no-e2e --no-traffic"
python bench.py --steps 3 --warmup 1 $COMMON > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --mono-frac 0.2 --sites 20000 --steps 1 --warmup 0 $COMMON > $O/bench_mono20_20k.json 2> $O/bench_mono20_20k.err
NGSLD_REPLAY=0 timeout 600 python bench.py --mono-frac 0.2 --sites 20000 --steps 1 --warmup 0 $COMMON > $O/bench_mono20_20k_noreplay.json 2> $O/bench_mono20_20k_noreplay.err
timeout 600 python bench.py --sfs --sites 20000 --steps 1 --warmup 0 $COMMON > $O/bench_sfs_20k.json 2> $O/bench_sfs_20k.err
timeout 1200 python bench.py --mono-frac 0.2 --steps 1 --warmup 0 $COMMON > $O/bench_mono20_full.json 2> $O/bench_mono20_full.err
nproc > $O/nproc.txt
tail -c 600 $O/*.err
