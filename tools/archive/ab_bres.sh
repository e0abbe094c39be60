#!/usr/bin/env bash
# Same-box A/B: the streaming kernel with the candidate's vector resident (default; beyond 10,240 individuals part of it) against the plain
# streaming kernel (NGSLD_PAIR_KERNEL=stream), with record checksums.   tools/ab_bres.sh > gpurun_out/r03/sweep_bres.txt
NINDS="${NINDS:-5121 5632 6000 7000 8000 9000 10000 10240 10241 12000 16000 20000}" bash tools/sweep_variants.sh "plain=NGSLD_PAIR_KERNEL=stream" "resident=NGSLD_PAIR_KERNEL=bres"
