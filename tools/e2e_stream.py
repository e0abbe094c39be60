#!/usr/bin/env python
"""Dev measurement (GPU box): the drop-in binary on a configs[4]-shaped shard (n_ind 2000, 500 kb window over
~1 kb gaps), resident against streamed under a --max_gpu_mem cap: same bytes out, wall time of both.
python tools/e2e_stream.py [n_sites] [n_ind] [max_gpu_mem_GB] [threads]"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ngsld_amd import capi, shard, synth  # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
n_ind = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
cap_gb = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
threads = int(sys.argv[4]) if len(sys.argv) > 4 else os.cpu_count()


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest(), os.path.getsize(path)


with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    g, p = os.path.join(d, "in.glf"), os.path.join(d, "in.pos")
    raw = synth.make_gl_torch(n_sites, n_ind, 5, torch.device("cuda", 0), mono_frac=float(os.environ.get("E2E_MONO", "0")))   # E2E_MONO=0.2: not SNP-called
    with open(g, "wb") as fh:
        for s in range(0, n_sites, 8192):
            fh.write(raw[s:s + 8192].cpu().numpy().tobytes())
    del raw
    torch.cuda.empty_cache()
    chrs, pos = synth.make_positions(n_sites, 5, max_gap=2000)
    synth.write_pos(p, chrs, pos)
    pd = shard.pos_dist_from_positions(chrs, pos)
    n_pairs = int(shard.row_pair_counts(pd, 500, 0).sum())
    slab = capi.slab_sites_for_budget(n_ind, int(cap_gb * 1e9))
    print(f"{n_sites} x {n_ind}: file {os.path.getsize(g) / 1e9:.2f} GB, {n_pairs} pairs; cap {cap_gb} GB -> slabs of "
          f"{slab} sites, {len(capi.plan_slabs(pd, n_sites, slab, max_kb_dist=500))} slabs")
    res = {}
    for mode, extra in (("resident", []), ("pipelined", []), ("streamed", ["--max_gpu_mem", str(cap_gb)])):
        o = os.path.join(d, mode + ".ld")
        t0 = time.perf_counter()
        r = subprocess.run([capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p,
                            "--max_kb_dist", "500", "--n_threads", str(threads), "--verbose", "1", "--out", o] + extra,
                           capture_output=True, text=True,
                           env=dict(os.environ, NGSLD_PIPELINE="0" if mode == "resident" else "1"))
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
        assert ("Streaming the genotype matrix" in r.stderr) == (mode != "resident"), r.stderr
        res[mode] = md5(o)
        os.unlink(o)
        print(f"  {mode}: {dt:.2f} s wall, {n_pairs / dt / 1e6:.2f} M pairs/s file -> TSV, md5 {res[mode][0]} "
              f"({res[mode][1] / 1e9:.2f} GB)")
    assert res["resident"] == res["streamed"] == res["pipelined"]
    print("  IDENTICAL output")
