"""Dev probe: executed EM iterations by site distance on the bench generator (all pairs of a few thousand sites)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from ngsld_amd import capi, synth

dev = torch.device("cuda", 0)
for n_ind in (500, 1000):
    n_sites = 4000
    raw = synth.make_gl_torch(n_sites, n_ind, 3, dev, depth=10.0)
    eng = capi.Engine(0)
    eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
    eng.set_pos_dist(None)
    n = eng.plan(extend_out=True)
    s1, s2, std, ext = eng.run()
    eng.close()
    d = (s2 - s1).astype(np.int64)
    it = np.minimum(ext["n_iter"].astype(np.int64) + 1, 100)
    print("n_ind", n_ind, "mean", it.mean())
    for lo, hi in ((1, 10), (10, 30), (30, 100), (100, 300), (300, 1000), (1000, 4000)):
        m = (d >= lo) & (d < hi)
        print(f"  distance [{lo},{hi}): pairs {m.sum():9d}  mean executed iterations {it[m].mean():6.2f}  median {np.median(it[m]):4.0f}  mean r2 {np.nanmean(std['r2'][m]):.4f}")
