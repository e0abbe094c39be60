#!/usr/bin/env python
"""EXPERIMENT (round 6, verdict item 4): a lane per pair for configs[1]'s shape, timed beside the library's lockstep kernel on one box.
    python tools/experiments/c1_lanes/run.py [n_sites]      (needs libc1lanes.so: tools/experiments/c1_lanes/build.sh)
Checks hap[4] / n_iter against the library's own records, then times both (HIP events)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..", "..")))
from ngsld_amd import capi, synth  # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n_ind = 100
dev = torch.device("cuda:0")
L = C.CDLL(os.path.join(HERE, "libc1lanes.so"))
L.c1_lanes_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_void_p]
KMAX = L.c1_lanes_max_seg()

raw = synth.make_gl_torch(n_sites, n_ind, 2, dev, depth=10.0)          # [site][ind][3], configs[1]'s seed
eng = capi.Engine(0)
eng.set_geno_raw(raw.data_ptr(), n_sites, n_ind)
n_pairs = eng.plan(max_kb_dist=0, extend_out=True)
assert n_pairs == n_sites * (n_sites - 1) // 2
maf = torch.from_numpy(eng.maf()).to(dev)
d_std = torch.empty(n_pairs * capi.REC_STD.itemsize, dtype=torch.uint8, device=dev)
d_ext = torch.empty(n_pairs * capi.REC_EXT.itemsize, dtype=torch.uint8, device=dev)


def lib_pass():
    eng.run_device(0, n_sites, d_std.data_ptr(), d_ext.data_ptr())
    eng.finish_device()


lib_pass()
torch.cuda.synchronize()
ext = np.frombuffer(d_ext.cpu().numpy().tobytes(), dtype=capi.REC_EXT)
std = np.frombuffer(d_std.cpu().numpy().tobytes(), dtype=capi.REC_STD)

gl = (raw / raw.sum(dim=2, keepdim=True)).permute(0, 2, 1).contiguous()  # [site][3][ind]

# the stream: row blocks of 32, each against the sites behind its first row; 256 equal shares of the partners
ROWS = 32
n_wg = 256
units = []  # (row0, first partner, partners)
for row0 in range(0, n_sites - 1, ROWS):
    units.append((row0, row0 + 1, n_sites - 1 - row0))
total = sum(u[2] for u in units)
share = -(-total // n_wg)
segs = np.zeros((n_wg, KMAX, 4), dtype=np.uint32)
n_segs = np.zeros(n_wg, dtype=np.uint32)
wg, room = 0, share
for row0, k0, nk in units:
    while nk > 0:
        take = min(nk, room)
        j = n_segs[wg]
        assert j < KMAX
        segs[wg, j] = (row0, k0, k0 + take, 0)
        n_segs[wg] += 1
        k0 += take
        nk -= take
        room -= take
        if room == 0:
            wg, room = wg + 1, share
d_segs = torch.from_numpy(segs).to(dev)
d_nsegs = torch.from_numpy(n_segs).to(dev)
out_f = torch.full((n_pairs, 4), -1.0, dtype=torch.float64, device=dev)
out_ld = torch.full((n_pairs, 3), -1.0, dtype=torch.float64, device=dev)
out_it = torch.full((n_pairs,), 0xFFFFFFFF, dtype=torch.int64, device=dev).to(torch.int32)  # (u32 view)
out_it = torch.full((n_pairs,), -1, dtype=torch.int32, device=dev)


dbg = torch.zeros((n_wg, 4, 4), dtype=torch.int64, device=dev)
use_dbg = [False]


def lane_pass():
    rc = L.c1_lanes_run(gl.data_ptr(), maf.data_ptr(), n_sites, n_ind, d_segs.data_ptr(), d_nsegs.data_ptr(), n_wg, out_f.data_ptr(),
                        out_ld.data_ptr(), out_it.data_ptr(), torch.cuda.current_stream().cuda_stream, dbg.data_ptr() if use_dbg[0] else None)
    assert rc == 0, rc


lane_pass()
torch.cuda.synchronize()
f = out_f.cpu().numpy()
it = out_it.cpu().numpy().astype(np.int64)
ld = out_ld.cpu().numpy()
missing = int((it < 0).sum())
same_iter = int((it == ext["n_iter"].astype(np.int64)).sum())
dh = np.abs(f - ext["hap"])
dh = np.where(np.isnan(f) & np.isnan(ext["hap"]), 0.0, dh)
dr = np.abs(ld[:, 2] - std["r2"])
dr = np.where(np.isnan(ld[:, 2]) & np.isnan(std["r2"]), 0.0, dr)
print(f"{n_pairs} pairs: records not written {missing}; n_iter equal {same_iter} ({same_iter / n_pairs:.6f}); largest |hap difference| "
      f"{np.nanmax(dh):.3e}; pairs with |r2 difference| > 1e-9: {int((dr > 1e-9).sum())} (NaN differences {int(np.isnan(dr).sum())}); "
      f"executed EM steps {int((ext['n_iter'].astype(np.int64) + (ext['n_iter'] < 100)).sum())}")


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


use_dbg[0] = True
lane_pass()
torch.cuda.synchronize()
use_dbg[0] = False
d = dbg.cpu().numpy().astype(np.float64)
print(f"per wavefront (s_memtime ticks): individuals' loop {d[:, :, 0].mean():.3e} of {d[:, :, 1].mean():.3e} in all (slowest wavefront {d[:, :, 1].max():.3e}, fastest "
      f"{d[:, :, 1].min():.3e}); steps {d[:, :, 2].mean():.0f}; ticks per step in the loop {d[:, :, 0].sum() / d[:, :, 2].sum():.0f}; lanes active per step "
      f"{d[:, :, 3].sum() / d[:, :, 2].sum():.1f}")
order = np.argsort(-d[:, :, 1].max(axis=1))
for wgi in list(order[:6]) + list(order[-3:]):
    print(f"  workgroup {wgi}: segments {n_segs[wgi]}, ticks {d[wgi, :, 1].max():.3e}, in the loop {d[wgi, :, 0].mean():.3e}, steps {d[wgi, :, 2].mean():.0f}, "
          f"first segment {tuple(segs[wgi, 0, :3])}")
for rnd in range(3):
    t_lane = timed(lane_pass)
    t_lib = timed(lib_pass)
    k_ms = eng.last_kernel_time()[0]
    print(f"round {rnd}: lane per pair {t_lane:.2f} ms a pass; library (lockstep `{eng.pair_kernel()}` kernel + its replay) {t_lib:.2f} ms a pass, "
          f"pair kernel {k_ms:.2f} ms")
