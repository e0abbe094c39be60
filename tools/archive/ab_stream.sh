#!/usr/bin/env bash
# Same-box: the streaming kernel (n_ind > 5,120) of this build against round 2's library.  tools/ab_stream.sh > gpurun_out/sweep_stream.txt
# (round 2's library: git worktree add /tmp/wt2 8f1eda4 && make -C /tmp/wt2/ngsld_amd/csrc && cp /tmp/wt2/ngsld_amd/libngsld.so ngsld_amd/ab/libngsld_r02.so)
A=$PWD/ngsld_amd/ab
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streaming" 2>&1 | tail -3
NINDS="${NINDS:-5121 6000 8000 10000 16000}" timeout 900 tools/sweep_variants.sh "r02=NGSLD_LIB=$A/libngsld_r02.so" "now="
