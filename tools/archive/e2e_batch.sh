#!/usr/bin/env bash
# Dev: configs[2] end to end through the drop-in binary for several NGSLD_TEST_BATCH_PAIRS (device-side TSV).
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
for bp in 8388608 4194304 2097152 1048576 524288; do
  echo "NGSLD_TEST_BATCH_PAIRS=$bp"
  NGSLD_TIMING=1 NGSLD_TEST_BATCH_PAIRS=$bp ngsld_amd/bin/ngsLD --geno $D/in.glf --n_ind 500 --n_sites 100000 --pos $D/in.pos --max_kb_dist 100 --extend_out --n_threads 8 --verbose 0 --out /dev/null 2>&1 | tail -1
done
rm -rf $D
