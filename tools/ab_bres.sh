#!/usr/bin/env bash
# Same-box A/B: the streaming kernel with the candidate's vector resident (default up to 10,240 individuals) against the plain
# streaming kernel (NGSLD_STREAM_RESIDENT=0), with record checksums.   tools/ab_bres.sh > gpurun_out/r03/sweep_bres.txt
NINDS="${NINDS:-5121 5632 6000 7000 8000 9000 10000 10240 10241}" bash tools/sweep_variants.sh "plain=NGSLD_STREAM_RESIDENT=0" "resident="
