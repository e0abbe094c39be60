#!/usr/bin/env bash
# Dev: the drop-in binary on configs[2], rows written in aligned 8-byte words (default) against byte stores.
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
CMD="ngsld_amd/bin/ngsLD --geno $D/in.glf --n_ind 500 --n_sites 100000 --pos $D/in.pos --max_kb_dist 100 --extend_out --n_threads 16 --verbose 0"
$CMD --out $D/a.ld; LD_PRELOAD=$PWD/ngsld_amd/ab/libngsld_bytes.so $CMD --out $D/b.ld
cmp $D/a.ld $D/b.ld && echo "outputs byte-identical: $(stat -c %s $D/a.ld) bytes, md5 $(md5sum < $D/a.ld)"
rm -f $D/a.ld $D/b.ld
for r in 1 2; do
  echo "== round $r words"; NGSLD_TIMING=1 $CMD --out /dev/null 2>&1 | grep -E "pair kernels|total"
  echo "== round $r bytes"; NGSLD_TIMING=1 LD_PRELOAD=$PWD/ngsld_amd/ab/libngsld_bytes.so $CMD --out /dev/null 2>&1 | grep -E "pair kernels|total"
done
rm -rf $D
