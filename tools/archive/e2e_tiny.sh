set -e
D=/dev/shm/e2e_tiny; mkdir -p $D
python - $D <<'PY'
import sys, os
sys.path.insert(0, ".")
import torch
from ngsld_amd import synth
d = sys.argv[1]
synth.make_gl_torch(100000, 8, 3, torch.device("cuda", 0)).cpu().numpy().tofile(os.path.join(d, "in.glf"))
chrs, pos = synth.make_positions(100000, 3)
synth.write_pos(os.path.join(d, "in.pos"), chrs, pos)
PY
CMD="ngsld_amd/bin/ngsLD --geno $D/in.glf --n_ind 8 --n_sites 100000 --pos $D/in.pos --max_kb_dist 100 --extend_out --n_threads 16 --verbose 2 --out /dev/null"
for t in 8 16 32; do echo "== NGSLD_REPLAY_THREADS=$t"; NGSLD_TIMING=1 NGSLD_REPLAY_THREADS=$t $CMD 2>&1 | grep -E "pair kernels|total|replayed"; done
echo "== replay off"; NGSLD_TIMING=1 NGSLD_REPLAY=0 $CMD 2>&1 | grep -E "pair kernels|total"
rm -rf $D
