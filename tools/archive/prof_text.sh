#!/usr/bin/env bash
# rocprofv3 kernel stats of the drop-in binary on configs[2], text write pass in 8-byte words against byte stores.
R=${GRAFT_REPO_ROOT:-/root/repo}; D=/dev/shm/prof_text; mkdir -p $D
cd $R
python - <<PY
import os, sys
sys.path.insert(0, "$R")
import torch
from ngsld_amd import synth
n_sites, n_ind = 100000, 500
synth.make_gl_torch(n_sites, n_ind, 3, torch.device("cuda", 0)).cpu().numpy().tofile("$D/in.glf")
chrs, pos = synth.make_positions(n_sites, 3)
synth.write_pos("$D/in.pos", chrs, pos)
PY
cd /tmp && export TMPDIR=/tmp
CMD="$R/ngsld_amd/bin/ngsLD --geno $D/in.glf --n_ind 500 --n_sites 100000 --pos $D/in.pos --max_kb_dist 100 --extend_out --n_threads 16 --verbose 0 --out /dev/null"
for v in words bytes; do
  rm -rf /tmp/prof_text_out
  if [ $v = bytes ]; then export LD_PRELOAD=$R/ngsld_amd/ab/libngsld_bytes.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_text_out -o run -- $CMD > /dev/null 2>&1
  unset LD_PRELOAD
  echo "== $v"; cut -d, -f1-6 /tmp/prof_text_out/run_kernel_stats.csv | head -6
done
rm -rf $D
