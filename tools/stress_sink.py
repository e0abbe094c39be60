"""Dev stress: large all-pairs jobs through ngsld_run (batched kernel || D2H || sink), counting records and checking a
checksum of what the sink sees against the device-resident run of a row subset."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from ngsld_amd import capi, synth
import ctypes as C

dev = torch.device("cuda", 0)
for n_sites, n_ind in ((30000, 8), (20000, 100), (12000, 500)):
    raw = synth.make_gl_torch(n_sites, n_ind, 5, dev, depth=6.0)
    eng = capi.Engine(0)
    eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
    eng.set_pos_dist(None)
    n = eng.plan(extend_out=True)
    acc = {"n": 0, "iters": 0, "bad": 0}
    def sink(_u, bp):
        b = bp.contents
        ext = np.ctypeslib.as_array(C.cast(b.ext, C.POINTER(C.c_uint32)), shape=(b.n_pairs, 10))
        acc["n"] += b.n_pairs
        acc["iters"] += int(ext[:, 9].astype(np.int64).sum())
        acc["bad"] += int((ext[:, 8] != n_ind).sum())
        return 0
    t0 = time.perf_counter()
    eng._check(eng._L.ngsld_run(eng._h, 0, n_sites, capi.SINK_FN(sink), None))
    dt = time.perf_counter() - t0
    # device-resident run of the first rows for a cross-check of the iteration total
    rows = 200
    ro, _ = eng.plan_rows()
    k = int(ro[rows])
    d_std = torch.empty(k * 32, dtype=torch.uint8, device=dev); d_ext = torch.empty(k * 40, dtype=torch.uint8, device=dev)
    eng.run_device(0, rows, d_std.data_ptr(), d_ext.data_ptr(), None)
    print(f"{n_sites} x {n_ind}: plan {n} pairs, sink saw {acc['n']} ({'OK' if acc['n']==n==n_sites*(n_sites-1)//2 else 'MISMATCH'}), "
          f"sample_size wrong on {acc['bad']}, mean n_iter {acc['iters']/max(acc['n'],1):.3f}, {dt:.2f} s = {n/dt:.3g} pairs/s through the sink")
    eng.close()
    del raw
