#!/usr/bin/env bash
# Dev: where the drop-in binary's wall time goes on configs[2] (NGSLD_TIMING=1), streamed-in-slabs vs resident path.
set -e
D=/dev/shm/e2e_$$; mkdir -p $D
python - $D <<'PY'
import sys, os
sys.path.insert(0, ".")
import torch
from ngsld_amd import synth
d = sys.argv[1]
synth.make_gl_torch(100000, 500, 3, torch.device("cuda", 0)).cpu().numpy().tofile(os.path.join(d, "in.glf"))
chrs, pos = synth.make_positions(100000, 3)
synth.write_pos(os.path.join(d, "in.pos"), chrs, pos)
PY
for t in 8 256; do for pipe in 0; do
  echo "--n_threads $t NGSLD_PIPELINE=$pipe"
  NGSLD_TIMING=1 NGSLD_PIPELINE=$pipe ngsld_amd/bin/ngsLD --geno $D/in.glf --n_ind 500 --n_sites 100000 --pos $D/in.pos --max_kb_dist 100 --extend_out --n_threads $t --verbose 0 --out /dev/null 2>&1 | tail -12
done; done
rm -rf $D
