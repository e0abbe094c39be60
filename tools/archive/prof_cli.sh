#!/usr/bin/env bash
# rocprofv3 kernel stats of the drop-in binary on configs[2] (file -> TSV to /dev/null): which kernels the end-to-end time goes to.
R=${GRAFT_REPO_ROOT:-/root/repo}; D=/dev/shm/prof_cli; mkdir -p $D $R/gpurun_out/r02
cd $R
python - <<PY
import os, sys
sys.path.insert(0, "$R")
import torch
from ngsld_amd import synth
n_sites, n_ind = 100000, 500
raw = synth.make_gl_torch(n_sites, n_ind, 3, torch.device("cuda", 0)).cpu().numpy()
raw.tofile("$D/in.glf")
chrs, pos = synth.make_positions(n_sites, 3)
synth.write_pos("$D/in.pos", chrs, pos)
PY
cd /tmp && export TMPDIR=/tmp
CMD="$R/ngsld_amd/bin/ngsLD --geno $D/in.glf --n_ind 500 --n_sites 100000 --pos $D/in.pos --max_kb_dist 100 --extend_out --n_threads 16 --verbose 0 --out /dev/null"
NGSLD_TIMING=1 $CMD 2>&1 | tail -12
rm -rf /tmp/prof_cli_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cli_out -o run -- $CMD > /dev/null 2>&1
cut -d, -f1-6 /tmp/prof_cli_out/run_kernel_stats.csv | head -12 | tee $R/gpurun_out/r02/cli_kernel_stats.csv
rm -rf $D
