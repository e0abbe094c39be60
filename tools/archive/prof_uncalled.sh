#!/bin/bash
# rocprofv3 kernel trace of the drop-in binary on configs[2]'s shape with 20 % monomorphic sites (text path, device-side replay)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r05_prof; mkdir -p $O
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from ngsld_amd import synth
n=100000
raw = synth.make_gl_torch(n, 500, 3, torch.device("cuda", 0), mono_frac=float(os.environ.get("MONO","0.2")), sfs=os.environ.get("SFS")=="1")
with open("/dev/shm/in.glf","wb") as fh:
    for lo in range(0,n,20000): fh.write(raw[lo:lo+20000].cpu().numpy().tobytes())
chrs,pos = synth.make_positions(n,3)
synth.write_pos("/dev/shm/in.pos",chrs,pos)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O -o cli -- $OLDPWD/ngsld_amd/bin/ngsLD --geno /dev/shm/in.glf --n_ind 500 --n_sites 100000 --pos /dev/shm/in.pos --max_kb_dist 100 --extend_out --n_threads 16 --verbose 0 --out /dev/null > $O/run.log 2>&1
ls $O
find $O -name "*kernel_stats.csv" | head -1 | xargs head -20
rm -f /dev/shm/in.glf /dev/shm/in.pos
