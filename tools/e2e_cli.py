#!/usr/bin/env python
"""Dev measurement (GPU box): end-to-end wall time of the ngsLD drop-in binary, file -> TSV (to /dev/null),
on a synthetic binary GL file.  python tools/e2e_cli.py [n_sites] [n_ind] [threads]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngsld_amd import capi, shard, synth  # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_ind = int(sys.argv[2]) if len(sys.argv) > 2 else 500
threads = int(sys.argv[3]) if len(sys.argv) > 3 else os.cpu_count()
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    import torch
    raw = synth.make_gl_torch(n_sites, n_ind, 3, torch.device("cuda", 0)).cpu().numpy()
    g, p = os.path.join(d, "in.glf"), os.path.join(d, "in.pos")
    raw.tofile(g)
    chrs, pos = synth.make_positions(n_sites, 3)
    synth.write_pos(p, chrs, pos)
    n_pairs = int(shard.row_pair_counts(shard.pos_dist_from_positions(chrs, pos), 100, 0).sum())
    del raw
    if os.environ.get("E2E_MD5") == "1":      # content check: thread count must not change a single byte
        import hashlib
        sums = []
        for t in (1, threads):
            o = os.path.join(d, f"out{t}.ld")
            r = subprocess.run([capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p,
                                "--max_kb_dist", "100", "--extend_out", "--n_threads", str(t), "--verbose", "0",
                                "--out", o], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            h = hashlib.md5()
            with open(o, "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""):
                    h.update(blk)
            sums.append((h.hexdigest(), os.path.getsize(o)))
            os.unlink(o)
        print("md5/size per thread count:", sums, "IDENTICAL" if sums[0] == sums[1] else "DIFFERENT")
        assert sums[0] == sums[1]
    for t in (1, threads):
        t0 = time.perf_counter()
        r = subprocess.run([capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p,
                            "--max_kb_dist", "100", "--extend_out", "--n_threads", str(t), "--verbose", "0",
                            "--out", "/dev/null"], capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
        print(f"ngsLD end to end: {n_sites} x {n_ind}, {n_pairs} pairs, --n_threads {t}: {dt:.2f} s "
              f"({n_pairs / dt / 1e6:.2f} M pairs/s incl. file read, H2D, D2H, text)")
