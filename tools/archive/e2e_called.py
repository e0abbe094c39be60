#!/usr/bin/env python
"""Dev measurement (GPU box): end-to-end wall time of the ngsLD drop-in binary on a CALLED-genotype text file
({-1,0,1,2}, gzip-compressed) -> TSV (to /dev/null), with the per-phase report.  python tools/e2e_called.py [n_sites] [n_ind]"""
import gzip
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
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    import torch
    g_called = synth.make_gl_torch(n_sites, n_ind, 3, torch.device("cuda", 0)).argmax(dim=2).cpu().numpy().astype(np.int8)
    g_called[np.random.default_rng(3).random(g_called.shape) < 0.05] = -1
    g, p = os.path.join(d, "in.geno.gz"), os.path.join(d, "in.pos")
    with gzip.open(g, "wt", compresslevel=1) as fh:
        for row in g_called:
            fh.write("\t".join(map(str, row.tolist())))
            fh.write("\n")
    chrs, pos = synth.make_positions(n_sites, 3)
    synth.write_pos(p, chrs, pos)
    n_pairs = int(shard.row_pair_counts(shard.pos_dist_from_positions(chrs, pos), 100, 0).sum())
    for extra in ([], ["--extend_out"]):
        t0 = time.perf_counter()
        r = subprocess.run([capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p,
                            "--max_kb_dist", "100", "--n_threads", str(os.cpu_count()), "--verbose", "0", "--out", "/dev/null"] + extra,
                           capture_output=True, text=True, env=dict(os.environ, NGSLD_TIMING="1"))
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
        print(f"{n_sites} x {n_ind} called text.gz ({os.path.getsize(g) / 1e6:.1f} MB), {n_pairs} pairs{' --extend_out' if extra else ''}: "
              f"{dt:.2f} s end to end = {n_pairs / dt:.3g} rows/s")
        print(r.stderr.strip()[-600:])
