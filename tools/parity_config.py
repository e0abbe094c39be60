"""Every-pair parity of a BASELINE configuration against the oracle (GPU box; the oracle on all host cores).

    python tools/parity_config.py c1            # configs[1]: 5,000 x 100, all 12,497,500 pairs
    python tools/parity_config.py c2 [rows]     # configs[2]: 100,000 x 500, 100 kb window, rows [0, rows) (default 3000)
    python tools/parity_config.py c2mono [rows] # the same shape NOT SNP-called (20 % of the sites monomorphic): a third of the pairs are
                                                # replayed in the reference's operation order on the device -- bit for bit the oracle's
    python tools/parity_config.py c3 [rows]     # configs[3]: 50,000 x 1,000 all pairs, rows [0, rows) (default 24: 1.2e6 pairs)
    python tools/parity_config.py c4 [rows] [n_sites]  # configs[4]'s shape: n_sites (default 60,000) x 2,000, 500 kb window over
                                                # ~1 kb gaps, rows [0, rows) (default 1500)

Per pair: nIter and sample_size exact, hap / D / D' / r2 / r2_ExpG inside the tolerances written in tests/util.py.
Prints one JSON line (max absolute differences included); exit status 1 on any difference outside the tolerances.
Input is the same seeded generator the bench and the tests use (ngsld_amd.synth)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import torch  # noqa: E402,F401  (HIP runtime load order, see tests/conftest.py)

from ngsld_amd import capi, shard, synth  # noqa: E402
from oracle import orc  # noqa: E402
from util import check_records, pearson_tolerance  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c1"
    dev = torch.device("cuda", 0)
    if which == "c1":
        n_sites, n_ind, max_kb, seed, rows = 5000, 100, 0, 2, 5000
        pd = None
    elif which == "c3":
        n_sites, n_ind, max_kb, seed = 50_000, 1000, 0, 4
        rows = int(sys.argv[2]) if len(sys.argv) > 2 else 24
        pd = None
    elif which == "c4":
        n_sites, n_ind, max_kb, seed = (int(sys.argv[3]) if len(sys.argv) > 3 else 60_000), 2000, 500, 5
        rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
        chrs, pos = synth.make_positions(n_sites, seed, max_gap=2000)
        pd = shard.pos_dist_from_positions(chrs, pos)
    else:
        n_sites, n_ind, max_kb, seed = 100_000, 500, 100, 3
        rows = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
        chrs, pos = synth.make_positions(n_sites, seed)
        pd = shard.pos_dist_from_positions(chrs, pos)
    raw_t = synth.make_gl_torch(n_sites, n_ind, seed, dev, depth=10.0, mono_frac=0.2 if which == "c2mono" else 0.0)
    if pd is not None:  # the oracle only needs the rows and their halo
        ends = shard.row_ends(pd, max_kb, 0)
        n_have = int(ends[:rows].max())
        raw = raw_t[:n_have].cpu().numpy()
        pd_o = pd[:n_have].copy()
    else:
        raw, pd_o = raw_t.cpu().numpy(), None
    cores = len(os.sched_getaffinity(0))
    try:  # the cgroup CPU quota of the lease, when there is one (os.cpu_count() reports the machine)
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            cores = max(1, min(cores, int(float(q[0]) / float(q[1]) + 0.5)))
    except (OSError, ValueError, IndexError):
        pass
    t0 = time.perf_counter()
    o = orc.Oracle(raw, pd_o, max_kb_dist=max_kb, n_threads=cores)
    want = o.run(0, rows)
    t_cpu = time.perf_counter() - t0
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw_t.data_ptr(), n_sites=n_sites, n_ind=n_ind)
        if len(raw) == n_sites:
            eng.set_replay_source(raw)          # the exact-order replay reads the caller's values, as the CLI does
        elif which == "c2mono":                 # (there the replay is a third of the run: the whole matrix, as the binary holds it)
            whole = raw_t.cpu().numpy()
            eng.set_replay_source(whole)
        eng.set_pos_dist(pd)
        family = eng.pair_kernel()
        eng.plan(max_kb_dist=max_kb, extend_out=True)
        t0 = time.perf_counter()
        s1, s2, std, ext = eng.run(0, rows)
        t_gpu = time.perf_counter() - t0
        eng_replayed = [eng.replay_stats()[0]]
        eng_info = eng.replay_info()
    finally:
        eng.close()
    replayed = eng_replayed[0]
    out = {"config": which, "n_sites": n_sites, "n_ind": n_ind, "max_kb_dist": max_kb, "rows": rows, "pairs": int(len(want)),
           "pairs_replayed_exact_order": replayed, "replay": eng_info, "pair_kernel": family,
           "oracle_s": round(t_cpu, 1), "oracle_threads": cores, "gpu_sink_path_s": round(t_gpu, 2)}
    ok = len(want) == len(std) and np.array_equal(s1, want["s1"]) and np.array_equal(s2, want["s2"])
    out["pairs_and_order_equal"] = bool(ok)
    if ok:
        out["n_iter_equal"] = bool(np.array_equal(ext["n_iter"], want["n_iter"]))
        out["sample_size_equal"] = bool(np.array_equal(ext["n_ind_data"], want["n_ind_data"]))
        out["executed_iterations"] = int(np.minimum(want["n_iter"].astype(np.int64) + 1, 100).sum())
        with np.errstate(invalid="ignore"):
            for name, got, exp in (("hap", ext["hap"], want["hap"]), ("D", std["D"], want["D"]), ("Dp", std["Dp"], want["Dp"]),
                                   ("r2", std["r2"], want["r2"]), ("r2_ExpG", std["r2_ExpG"], want["r2pear"])):
                d = np.abs(np.asarray(got) - np.asarray(exp))
                out["max_abs_diff_" + name] = float(np.nanmax(d)) if d.size else 0.0
        try:
            check_records(std, ext, want, pearson_tol=pearson_tolerance(o.gl, s1, s2))
            out["within_tolerances"] = True
        except AssertionError as e:
            out["within_tolerances"] = False
            out["first_failure"] = str(e)[:300]
            ok = False
    print(json.dumps(out), flush=True)
    sys.exit(0 if ok and out.get("within_tolerances") else 1)


if __name__ == "__main__":
    main()
