#!/usr/bin/env python
"""Dev measurement (GPU box): end-to-end wall time of the ngsLD drop-in binary vs --n_threads (file in /dev/shm -> /dev/null)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngsld_amd import capi, shard, synth
import torch
n_sites, n_ind = int(sys.argv[1]), int(sys.argv[2])
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    raw = synth.make_gl_torch(n_sites, n_ind, 3, torch.device("cuda", 0)).cpu().numpy()
    g, p = os.path.join(d, "in.glf"), os.path.join(d, "in.pos")
    raw.tofile(g); del raw
    chrs, pos = synth.make_positions(n_sites, 3)
    synth.write_pos(p, chrs, pos)
    base = [capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p, "--max_kb_dist", "100", "--verbose", "0"]
    for extra, t in ((["--extend_out"], 8), (["--extend_out"], 32), (["--extend_out"], 128), (["--extend_out"], 256), ([], 256)):
        t0 = time.perf_counter()
        r = subprocess.run(base + extra + ["--n_threads", str(t), "--out", "/dev/null"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        print(f"{n_sites} x {n_ind} {' '.join(extra) or 'standard columns'} --n_threads {t}: {time.perf_counter() - t0:.2f} s")
    t0 = time.perf_counter()
    r = subprocess.run(base + ["--max_snp_dist", "1", "--n_threads", "8", "--out", "/dev/null"], capture_output=True, text=True)
    print(f"same input, one pair per row (fixed costs: start-up, file read, upload, prep): {time.perf_counter() - t0:.2f} s")
