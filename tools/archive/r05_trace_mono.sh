cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from ngsld_amd import synth
n=100000
raw = synth.make_gl_torch(n, 500, 3, torch.device("cuda", 0), mono_frac=0.2)
with open("/dev/shm/in.glf","wb") as fh:
    for lo in range(0,n,20000): fh.write(raw[lo:lo+20000].cpu().numpy().tobytes())
chrs,pos = synth.make_positions(n,3)
synth.write_pos("/dev/shm/in.pos",chrs,pos)
PY
NGSLD_TRACE=1 ngsld_amd/bin/ngsLD --geno /dev/shm/in.glf --n_ind 500 --n_sites 100000 --pos /dev/shm/in.pos --max_kb_dist 100 --extend_out --n_threads 16 --verbose 0 --out /dev/null 2> gpurun_out/trace_mono.txt
rm -f /dev/shm/in.glf /dev/shm/in.pos
