"""Dev probe (GPU box): the text path's loop on configs[2]'s shape WITHOUT torch in the process -- libngsld.so then runs on the
system's HIP runtime (/opt/rocm), as the drop-in binary does, not on the copy torch ships.  python tools/r04_text_loop_probe2.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ngsld_amd import capi, shard, synth

n_sites, n_ind = 100_000, 500
raw_h = np.concatenate([synth.make_gl_numpy(25_000, n_ind, 3 + k) for k in range(4)])
chrs, pos = synth.make_positions(n_sites, 3)
pd = shard.pos_dist_from_positions(chrs, pos)
labels = [f"{c}:{p}" for c, p in zip(chrs, pos)]
eng = capi.Engine(0)
eng.set_geno_raw(raw_h)
eng.set_replay_source(raw_h)
eng.set_pos_dist(pd)
eng.plan(max_kb_dist=100, extend_out=True)
eng.set_text_output(labels)
got = [0]


def sink(_u, bp):
    got[0] += bp.contents.text_len
    return 0


cb = capi.SINK_FN(sink)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    eng._check(eng._L.ngsld_run(eng._h, 0, n_sites, cb, None))
    ts.append(time.perf_counter() - t0)
print("no torch in the process, matrix from host memory: text loop, three passes: " + " / ".join(f"{t:.3f}" for t in ts) + " s")
import subprocess
print(subprocess.run("grep -E 'libamdhip64|libhsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid(), shell=True, capture_output=True, text=True).stdout)
eng.close()
