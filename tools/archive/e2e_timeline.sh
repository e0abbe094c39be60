#!/usr/bin/env bash
# Dev: kernel + copy timeline of the drop-in binary on configs[2] (rocprofv3), a window of ~60 ms from the middle of the run.
set -e
mkdir -p gpurun_out/r04
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
R=$PWD
CMD="$R/ngsld_amd/bin/ngsLD --geno $D/in.glf --n_ind 500 --n_sites 100000 --pos $D/in.pos --max_kb_dist 100 --extend_out --n_threads 16 --verbose 0 --out /dev/null"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/e2e_trace -o t -- env ${E2E_ENV:-} $CMD > /dev/null 2>&1 || true
cd $R
python - > gpurun_out/r04/e2e_timeline.txt <<'PY'
import csv, glob
k = glob.glob('/tmp/e2e_trace/**/*kernel_trace.csv', recursive=True)[0]
m = glob.glob('/tmp/e2e_trace/**/*memory_copy_trace.csv', recursive=True)
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'].split('(')[0][-48:], r.get('Queue_Id', '')))
if m:
    for r in csv.DictReader(open(m[0])):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r.get('Direction', '') + ' ' + r.get('Size', r.get('Bytes', '')), ''))
ev.sort()
pk = [e for e in ev if 'pair_ld' in e[2]]
t0 = pk[20][0]
for s, e, n, q in ev:
    if t0 - 1e6 <= s <= t0 + 45e6:
        print(f"{(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f} ms  ({(e - s) / 1e6:7.3f})  q{q:4s} {n}")
PY
rm -rf /tmp/e2e_trace $D
cat gpurun_out/r04/e2e_timeline.txt
