#!/bin/bash
# rocprofv3 kernel + copy trace of the drop-in binary on configs[2] with ONE hardware queue (GPU_MAX_HW_QUEUES=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r05_prof_q1; rm -rf $O; mkdir -p $O
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from ngsld_amd import synth
n=100000
raw = synth.make_gl_torch(n, 500, 3, torch.device("cuda", 0))
with open("/dev/shm/in.glf","wb") as fh:
    for lo in range(0,n,20000): fh.write(raw[lo:lo+20000].cpu().numpy().tobytes())
chrs,pos = synth.make_positions(n,3)
synth.write_pos("/dev/shm/in.pos",chrs,pos)
PY
R=$PWD
cd /tmp && export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=${Q:-1} rocprofv3 --kernel-trace --memory-copy-trace -d $O -o cli -- $R/ngsld_amd/bin/ngsLD --geno /dev/shm/in.glf --n_ind 500 --n_sites 100000 --pos /dev/shm/in.pos --max_kb_dist 100 --extend_out --n_threads 16 --verbose 0 --out /dev/null > $O/run.log 2>&1
rm -f /dev/shm/in.glf /dev/shm/in.pos
