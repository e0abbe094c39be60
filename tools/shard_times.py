#!/usr/bin/env python
"""Per-shard kernel times of the 8-GPU configurations, measured on ONE MI355X: the eight row shards a world-size-8 run deals
out (bench.py / shard.py: contiguous row ranges), each run by itself on the whole device, under both policies --
`pairs` (equal pair counts, the default) and `work` (equal ESTIMATED work: pairs x executed EM iterations, sampled on every
~100th row, bench.py --balance work).  An 8-GPU pass takes as long as its slowest shard, so

    predicted efficiency of the split = mean(shard time) / max(shard time)

is what row sharding itself costs (no communication happens on the data path; the one broadcast is outside the timed region).
It is a PREDICTOR: no 8-GPU node has run this code (SCALE_r0x.json are skipped records).

    python tools/shard_times.py [c3] [c4] [--c4-sites N]   ->  one JSON object on stdout
"""
from __future__ import annotations

import json
import os
import sys
import time
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def shard_times(name, n_sites, n_ind, max_kb, max_gap, seed, world=8, repeats=2):
    import torch
    import bench
    from ngsld_amd import capi, shard, synth
    dev = torch.device("cuda", 0)
    chrs, pos = synth.make_positions(n_sites, seed, max_gap=max_gap)
    pd = shard.pos_dist_from_positions(chrs, pos)
    row_end = shard.row_ends(pd, max_kb, 0)
    counts = row_end - (np.arange(n_sites, dtype=np.int64) + 1)
    raw = synth.make_gl_torch(n_sites, n_ind, seed, dev, depth=10.0)
    args = types.SimpleNamespace(ignore_miss=False, max_kb=max_kb, rnd_sample=1.0)
    work, info = bench.estimate_row_work(raw, pd, args, 0, n_sites, n_ind)
    out = {"config": name, "n_sites": n_sites, "n_ind": n_ind, "max_kb_dist": max_kb, "world": world,
           "pairs_total": int(counts.sum()), "work_estimate": info, "policies": {}}
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
        del raw
        torch.cuda.empty_cache()
        eng.set_replay(False)             # (kernel time only: what is compared is the split)
        eng.set_pos_dist(pd)
        eng.plan(max_kb_dist=max_kb, extend_out=True)
        row_off, _ = eng.plan_rows()
        row_off = row_off.astype(np.int64)
        for policy, bounds in (("pairs", shard.split_rows(counts, world)), ("work", shard.split_rows_weighted(work, world))):
            ms, pairs, iters = [], [], []
            for lo, hi in bounds:
                m = int(row_off[hi] - row_off[lo])
                d_std = torch.empty(max(m, 1) * 32, dtype=torch.uint8, device=dev)
                d_ext = torch.empty(max(m, 1) * 40, dtype=torch.uint8, device=dev)
                best = None
                for _ in range(repeats):
                    eng.run_device(int(lo), int(hi), d_std.data_ptr(), d_ext.data_ptr(), None)
                    t, _, _ = eng.last_kernel_time()
                    best = t if best is None else min(best, t)
                it = d_ext.view(torch.int32).view(-1, 10)[:m, 9].to(torch.int64)
                ms.append(best)
                pairs.append(m)
                iters.append(float(torch.clamp(it + 1, max=100).sum()) / max(m, 1))
                del d_std, d_ext
            t = np.asarray(ms)
            out["policies"][policy] = {
                "rows": [[int(a), int(b)] for a, b in bounds], "pairs": pairs, "kernel_ms": [round(x, 2) for x in ms],
                "mean_executed_iterations": [round(x, 3) for x in iters],
                "max_over_mean": float(t.max() / t.mean()), "predicted_efficiency": float(t.mean() / t.max()),
                "sum_ms": float(t.sum())}
    finally:
        eng.close()
    return out


def main():
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c3", "c4"]
    c4_sites = 1_000_000
    if "--c4-sites" in sys.argv:
        c4_sites = int(sys.argv[sys.argv.index("--c4-sites") + 1])
    res = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "device": None, "configs": []}
    import torch
    res["device"] = torch.cuda.get_device_name(0)
    t0 = time.time()
    if "c3" in which:
        res["configs"].append(shard_times("BASELINE configs[3]: 50,000 x 1,000 all pairs", 50_000, 1000, 0, 200, 4))
    if "c4" in which:
        res["configs"].append(shard_times(f"BASELINE configs[4]: {c4_sites:,} x 2,000, 500 kb window, ~1 kb gaps", c4_sites, 2000, 500, 2000, 5))
    res["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
