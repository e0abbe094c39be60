#!/usr/bin/env python
"""GPU box: CALLED genotypes with missing calls (-1 in a text genotype file: three equal likelihoods, read_data.cpp:90-96) on a
matrix that is not SNP-called (20 % monomorphic sites): where are the flagged pairs replayed, and at what rate?
python tools/r05_called_missing.py [n_sites] [n_ind]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ngsld_amd import capi, shard, synth

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
n_ind = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda", 0)
gl = synth.make_gl_torch(n_sites, n_ind, 3, dev, mono_frac=0.2)
g = torch.argmax(gl, dim=2)
raw = torch.full_like(gl, -1e15)                       # what read_geno stores for a text genotype file (log scale): log(1) over -INF ...
raw.scatter_(2, g[..., None], 0.0)
for miss in (0.0, 0.05):
    r = raw.clone()
    if miss:
        gen = torch.Generator(device=dev); gen.manual_seed(1)
        m = torch.rand((n_sites, n_ind), generator=gen, device=dev) < miss
        r[m] = capi.missing_call_log()                 # ... and log(1/3) three times for a missing call
    host = r.cpu().numpy()
    chrs, pos = synth.make_positions(n_sites, 3)
    pd = shard.pos_dist_from_positions(chrs, pos)
    for text, call in ((True, None), (True, (0.0, 0.0))):
        eng = capi.Engine(0)
        try:
            eng.set_geno_raw(host, log_scale=True, text=text, call_geno=call)
            eng.set_pos_dist(pd)
            n = eng.plan(max_kb_dist=100, extend_out=True)
            t0 = time.perf_counter(); eng.run_discard(); t1 = time.perf_counter() - t0
            t0 = time.perf_counter(); eng.run_discard(); t2 = time.perf_counter() - t0
            print(f"missing {miss}: text={text} call_geno={call}: kernel {eng.pair_kernel()}, {n} pairs, second pass {n / t2:.4g} pairs/s ({t2:.3f} s; first {t1:.3f} s), {eng.replay_info()}", flush=True)
        finally:
            eng.close()
