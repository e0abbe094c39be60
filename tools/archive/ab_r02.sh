#!/usr/bin/env bash
# Round 2, same-box A/B at n_ind 1000 (configs[3] shape, 12,000 sites all pairs = 7.2e7 pairs): the a/b-form one-wavefront
# kernel (TREE 8 / 4) against the two-wavefront kernel; then parity of the new kernel on its cohort sizes.
export BENCH_ARGS="--no-cpu --no-sink --no-e2e --config c3 --sites ${SITES:-12000} --steps 2 --warmup 1"
tools/ab.sh "multi=NGSLD_PAIR_KERNEL=multi" "ab_tree8=X=1" "ab_tree4=NGSLD_LIB=$PWD/ngsld_amd/ab/libngsld_t4.so"
