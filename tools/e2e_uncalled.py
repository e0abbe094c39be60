#!/usr/bin/env python
"""GPU box: the drop-in binary end to end (binary GL file + positions in, extended TSV out to /dev/null) on configs[2]'s shape
with input that is NOT SNP-called (README.md:73): 20 % of the sites monomorphic / a log-uniform frequency spectrum, beside the
default generator.  NGSLD_TRACE shows where the exact store is built.  python tools/e2e_uncalled.py [n_sites]"""
import json
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ngsld_amd import capi, shard, synth  # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
n_ind = 500
threads = len(os.sched_getaffinity(0))
try:
    q = open("/sys/fs/cgroup/cpu.max").read().split()
    if q[0] != "max":
        threads = max(1, min(threads, int(float(q[0]) / float(q[1]) + 0.5)))
except (OSError, ValueError, IndexError):
    pass
out = {"n_sites": n_sites, "n_ind": n_ind, "threads": threads, "runs": {}}
chrs, pos = synth.make_positions(n_sites, 3)
n_pairs = int(shard.row_pair_counts(shard.pos_dist_from_positions(chrs, pos), 100, 0).sum())
out["pairs"] = n_pairs
only = os.environ.get("E2E_ONLY")
for name, kw in (("default", {}), ("mono20", {"mono_frac": 0.2}), ("sfs", {"sfs": True})):
    if only and name not in only.split(","):
        continue
    with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
        raw = synth.make_gl_torch(n_sites, n_ind, 3, torch.device("cuda", 0), **kw)
        if os.environ.get("E2E_JITTER"):   # every likelihood its own value: no triple repeats, the exact store's memo never hits
            g = torch.Generator(device=raw.device)
            g.manual_seed(99)
            raw *= 1.0 + 0.01 * torch.rand(raw.shape, generator=g, device=raw.device, dtype=raw.dtype)
        g, p = os.path.join(d, "in.glf"), os.path.join(d, "in.pos")
        with open(g, "wb") as fh:
            for lo in range(0, n_sites, 20000):
                fh.write(raw[lo:lo + 20000].cpu().numpy().tobytes())
        del raw
        torch.cuda.empty_cache()
        synth.write_pos(p, chrs, pos)
        cmd = [capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p, "--max_kb_dist", "100",
               "--extend_out", "--n_threads", str(threads), "--verbose", "2", "--out", "/dev/null"]
        times, last = [], ""
        for k in range(3):
            time.sleep(0.5)
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, NGSLD_TRACE="1") if k == 2 else None)
            times.append(round(time.perf_counter() - t0, 3))
            assert r.returncode == 0, r.stderr[-2000:]
            last = r.stderr
        os.makedirs("gpurun_out/r05_e2e", exist_ok=True)
        open(f"gpurun_out/r05_e2e/trace_{name}.txt", "w").write(last)
        out["runs"][name] = {"seconds": times, "pairs_per_s_best": n_pairs / min(times[:2]),
                             "stderr_tail": [ln for ln in last.splitlines() if "exact store" in ln or "replay" in ln.lower()][-6:]}
print(json.dumps(out))
