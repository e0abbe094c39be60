#!/usr/bin/env python
"""GPU box: what the exact store costs in device memory, priced (round 6, verdict item 5).  The drop-in binary on configs[2]'s
un-called twin (100,000 x 500, 100 kb, 20 % monomorphic sites) under --max_gpu_mem budgets between "everything fits" and "the
planes fit, planes + store + its individual-major copy do not": seconds file -> TSV (/dev/null), and where the flagged pairs
were replayed (the binary's --verbose 2 line).      python tools/r06_store_budget.py [n_sites]  ->  JSON on stdout"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ngsld_amd import capi, shard, synth  # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
n_ind = 500
chrs, pos = synth.make_positions(n_sites, 3)
n_pairs = int(shard.row_pair_counts(shard.pos_dist_from_positions(chrs, pos), 100, 0).sum())
planes_gb = n_sites * 3 * 512 * 8 / 1e9
out = {"n_sites": n_sites, "n_ind": n_ind, "pairs": n_pairs, "planes_gb": round(planes_gb, 3),
       "budget_helper": {"fixed_part_gb_two_contexts": 5.16,
                         "resident_with_store_from_gb": round(2.58 + 3 * planes_gb, 2), "resident_bare_from_gb": round(2.58 + planes_gb, 2)},
       "runs": {}}
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    raw = synth.make_gl_torch(n_sites, n_ind, 3, torch.device("cuda", 0), mono_frac=0.2)
    g, p = os.path.join(d, "in.glf"), os.path.join(d, "in.pos")
    with open(g, "wb") as fh:
        for lo in range(0, n_sites, 20000):
            fh.write(raw[lo:lo + 20000].cpu().numpy().tobytes())
    del raw
    torch.cuda.empty_cache()
    synth.write_pos(p, chrs, pos)
    for budget in (None, 2.58 + 3 * planes_gb + 0.3, 2.58 + 1.5 * planes_gb, 5.16 + 0.25, 2.58 + planes_gb + 0.1):
        cmd = [capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p, "--max_kb_dist", "100",
               "--extend_out", "--n_threads", "16", "--verbose", "2", "--out", "/dev/null"]
        if budget is not None:
            cmd += ["--max_gpu_mem", f"{budget:.2f}"]
        times, err = [], ""
        for _ in range(2):
            time.sleep(0.5)
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True)
            times.append(round(time.perf_counter() - t0, 3))
            err = r.stderr
            if r.returncode != 0:
                break
        line = [ln for ln in err.splitlines() if "replayed in the reference" in ln or "Streaming" in ln or "ERROR" in ln.upper()]
        m = re.search(r"\((\d+) on the device, (\d+) on host threads\)", err)
        out["runs"]["unlimited" if budget is None else f"{budget:.2f} GB"] = {
            "seconds": times, "pairs_per_s": n_pairs / min(times), "returncode": r.returncode, "stderr": line[-3:],
            "pairs_on_host": int(m.group(2)) if m else None}
print(json.dumps(out, indent=1))
