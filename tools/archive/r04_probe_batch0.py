#!/usr/bin/env python
"""Dev probe (GPU box): why is the first batch of an ngsld_run pass 2.5 ms slower than the others?  Kernel times (HIP
events) of run_device over row ranges of equal pair counts, in several orders."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ngsld_amd import capi, shard, synth  # noqa: E402

n_sites, n_ind = 100_000, 500
dev = torch.device("cuda", 0)
raw = synth.make_gl_torch(n_sites, n_ind, 3, dev)
chrs, pos = synth.make_positions(n_sites, 3)
pd = shard.pos_dist_from_positions(chrs, pos)
eng = capi.Engine(0)
eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
eng.set_pos_dist(pd)
n = eng.plan(max_kb_dist=100, extend_out=True)
row_off, _ = eng.plan_rows()
d_std = torch.empty(n * 32, dtype=torch.uint8, device=dev)
d_ext = torch.empty(n * 40, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
ext_i32 = d_ext.view(torch.int32).view(-1, 10)
cuts = [0]
while cuts[-1] < n_sites:
    r = int(np.searchsorted(row_off, row_off[cuts[-1]] + (1 << 23), side="right")) - 1
    cuts.append(min(max(r, cuts[-1] + 1), n_sites))
print("cuts", cuts)


def one(a, b):
    eng.run_device(a, b, d_std.data_ptr(), d_ext.data_ptr(), stream)
    eng.finish_device()
    ms, nl, npairs = eng.last_kernel_time()
    it = ext_i32[:npairs, 9].to(torch.int64)
    return ms, npairs, float(torch.clamp(it + 1, max=100).sum()) / npairs


one(0, n_sites)
for rep in range(2):
    for k in (0, 1, 2, 5, 0, 11, 0, 1):
        a, b = cuts[k], cuts[k + 1]
        ms, npairs, iters = one(a, b)
        print(f"rows [{a}, {b}): {npairs} pairs, kernel {ms:.2f} ms, {ms / npairs * 1e6:.4f} us/Mpair-ish, mean executed iterations {iters:.3f}", flush=True)
eng.close()
