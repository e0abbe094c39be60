#!/usr/bin/env bash
# Dev: the drop-in binary on configs[2], items per run 16 (default) / 8 / 4 / 2: shorter workgroups = shorter tail per text batch.
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
CMD="ngsld_amd/bin/ngsLD --geno $D/in.glf --n_ind 500 --n_sites 100000 --pos $D/in.pos --max_kb_dist 100 --extend_out --n_threads 16 --verbose 0 --out /dev/null"
for r in 1 2; do for e in 16 8 4 2; do
  echo "== round $r NGSLD_RUN_LEN=$e"
  NGSLD_TIMING=1 NGSLD_RUN_LEN=$e $CMD 2>&1 | grep -E "pair kernels|total"
done; done
rm -rf $D
