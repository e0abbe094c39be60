#!/usr/bin/env python
"""Dev measurement (GPU box): the host-resident leg (ngsld_run with a counting sink) on configs[2], repeated, with the
exact-order replay on and off.  python tools/sink_probe.py [n_sites] [n_ind]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ngsld_amd import capi, shard, synth  # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
n_ind = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda", 0)
raw = synth.make_gl_torch(n_sites, n_ind, 3, dev)
chrs, pos = synth.make_positions(n_sites, 3)
pd = shard.pos_dist_from_positions(chrs, pos)
eng = capi.Engine(0)
eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
eng.set_pos_dist(pd)
for replay in (True, False, True):
    eng.set_replay(replay)
    n = eng.plan(max_kb_dist=100, extend_out=True)
    for k in range(3):
        t0 = time.perf_counter()
        got = eng.run_discard(0, n_sites)
        dt = time.perf_counter() - t0
        ms, nl, _ = eng.last_kernel_time()
        print(f"replay={replay} pass {k}: {got} pairs in {dt * 1e3:.1f} ms ({got / dt / 1e6:.1f} M pairs/s), kernels {ms:.1f} ms in "
              f"{nl} launches, replayed {eng.replay_stats()[0]}", flush=True)
