#!/bin/bash
# where the time of the binary goes on un-called input of a large cohort: [timing] lines and the per-batch host timeline (NGSLD_TRACE)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from ngsld_amd import synth
n, ni = int(os.environ.get("SITES", "120000")), int(os.environ.get("IND", "2000"))
raw = synth.make_gl_torch(n, ni, 5, torch.device("cuda", 0), mono_frac=float(os.environ.get("MONO", "0.2")))
with open("/dev/shm/in.glf", "wb") as fh:
    for lo in range(0, n, 8192): fh.write(raw[lo:lo + 8192].cpu().numpy().tobytes())
chrs, pos = synth.make_positions(n, 5, max_gap=2000)
synth.write_pos("/dev/shm/in.pos", chrs, pos)
PY
for env in "NGSLD_X=product" "NGSLD_REPLAY_LANES_FROM=4194304" "NGSLD_REPLAY=0"; do
  echo "== $env"
  env $env NGSLD_PIPELINE=0 NGSLD_TRACE=1 ngsld_amd/bin/ngsLD --geno /dev/shm/in.glf --n_ind ${IND:-2000} --n_sites ${SITES:-120000} --pos /dev/shm/in.pos --max_kb_dist 500 --n_threads 16 --verbose 2 --out /dev/null 2> /tmp/err.txt
  grep "timing\|exact store\|replayed" /tmp/err.txt | tail -14
  grep "trace\] batch" /tmp/err.txt | sed -n '40,44p'
done
rm -f /dev/shm/in.glf /dev/shm/in.pos
