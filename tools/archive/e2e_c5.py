#!/usr/bin/env python
"""GPU box: BASELINE configs[4] at FULL size on one MI355X -- 1,000,000 sites x 2,000 individuals (a 48 GB binary GL file),
--max_kb_dist 500 with ~1 kb gaps (~5e8 pairs, ~75 GB of extended TSV) -- through the drop-in binary, twice:
  A  resident   the whole matrix on the device (NGSLD_PIPELINE=0: no slab pipeline)
  B  streamed   under --max_gpu_mem 24 (GB): slabs of rows + halo, two contexts alternating
and the two outputs compared byte for byte (cmp).  Then every pair of the first rows against the oracle through the API.
    python tools/e2e_c5.py [n_sites] [n_ind] [parity_rows] > gpurun_out/r02/e2e_c5.json
Scratch: /dev/shm (the input file and both outputs live there for the duration)."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
from ngsld_amd import capi, shard, synth  # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_ind = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000
parity_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000
max_kb, max_gap, seed = 500, 2000, 5
out = {"workload": f"{n_sites} x {n_ind}, --max_kb_dist {max_kb}, gaps ~ U[1,{max_gap}], --extend_out"}
dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    g, p = os.path.join(d, "in.glf"), os.path.join(d, "in.pos")
    t0 = time.perf_counter()
    chrs, pos = synth.make_positions(n_sites, seed, max_gap=max_gap)
    synth.write_pos(p, chrs, pos)
    pd = shard.pos_dist_from_positions(chrs, pos)
    n_pairs = int(shard.row_pair_counts(pd, max_kb, 0).sum())
    out["pairs"] = n_pairs
    # the matrix is generated on the device in one pass (the copying chain runs along the sites) and written out in chunks
    raw = synth.make_gl_torch(n_sites, n_ind, seed, dev)
    torch.cuda.synchronize()
    out["generate_s"] = round(time.perf_counter() - t0, 1)
    t0 = time.perf_counter()
    step = max(1, (1 << 30) // (n_ind * 24))
    head = None
    with open(g, "wb") as fh:
        for lo in range(0, n_sites, step):
            blk = raw[lo:lo + step].cpu().numpy()
            if lo == 0:
                head = blk
            fh.write(blk.tobytes())
    halo = int(shard.row_ends(pd, max_kb, 0)[:parity_rows].max())
    head = raw[:halo].cpu().numpy()
    del raw
    torch.cuda.empty_cache()
    out["write_file_s"] = round(time.perf_counter() - t0, 1)
    out["file_bytes"] = os.path.getsize(g)
    threads = min(64, len(os.sched_getaffinity(0)))
    base = [capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p, "--max_kb_dist",
            str(max_kb), "--extend_out", "--n_threads", str(threads), "--verbose", "1"]
    runs = {}
    for tag, extra, env in (("resident", [], {"NGSLD_PIPELINE": "0"}), ("streamed_24GB", ["--max_gpu_mem", "24"], {})):
        o = os.path.join(d, tag + ".ld")
        t0 = time.perf_counter()
        r = subprocess.run(base + extra + ["--out", o], capture_output=True, text=True,
                           env=dict(os.environ, NGSLD_TIMING="1", **env))
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr[-2000:]
        slabs = [l for l in r.stderr.splitlines() if "slabs" in l or "Streaming" in l]
        runs[tag] = {"seconds": round(dt, 2), "output_bytes": os.path.getsize(o), "pairs_per_s_file_to_tsv": n_pairs / dt,
                     "mode_lines": slabs[:2], "timing": [l for l in r.stderr.splitlines() if l.startswith("[timing]")][-12:]}
    t0 = time.perf_counter()
    same = subprocess.run(["cmp", os.path.join(d, "resident.ld"), os.path.join(d, "streamed_24GB.ld")]).returncode == 0
    out["cmp_s"] = round(time.perf_counter() - t0, 1)
    out["outputs_identical"] = same
    wc = subprocess.run(["wc", "-l", os.path.join(d, "resident.ld")], capture_output=True, text=True).stdout.split()[0]
    out["output_lines"] = int(wc)
    out["lines_equal_pairs_plus_header"] = int(wc) == n_pairs + 1
    md5 = subprocess.run(f"head -c {64 << 20} {os.path.join(d, 'resident.ld')} | md5sum", shell=True, capture_output=True,
                         text=True).stdout.split()[0]
    out["md5_first_64MiB"] = md5
    out["runs"] = runs
    assert same and out["lines_equal_pairs_plus_header"]

# ---- every pair of the first rows against the oracle (API, same matrix head) ----
from oracle import orc  # noqa: E402
from util import check_records  # noqa: E402
t0 = time.perf_counter()
o = orc.Oracle(head, pd[:len(head)], max_kb_dist=max_kb, n_threads=len(os.sched_getaffinity(0)))
want = o.run(0, parity_rows)
out["oracle_s"] = round(time.perf_counter() - t0, 1)
eng = capi.Engine(0)
eng.set_geno_raw(head)
eng.set_pos_dist(pd[:len(head)])
eng.plan(max_kb_dist=max_kb, extend_out=True)
s1, s2, std, ext = eng.run(0, parity_rows)
assert np.array_equal(s1, want["s1"]) and np.array_equal(s2, want["s2"])
n_checked = check_records(std, ext, want)
out["parity"] = {"rows": parity_rows, "pairs": int(n_checked), "nIter_and_sample_size_equal": True,
                 "max_abs_diff": {k: float(np.nanmax(np.abs(np.asarray(a, dtype=float) - np.asarray(b, dtype=float))))
                                  for k, a, b in (("hap", ext["hap"], want["hap"]), ("D", std["D"], want["D"]),
                                                  ("Dp", std["Dp"], want["Dp"]), ("r2", std["r2"], want["r2"]),
                                                  ("r2_ExpG", std["r2_ExpG"], want["r2pear"]))}}
eng.close()
print(json.dumps(out))
