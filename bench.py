#!/usr/bin/env python
"""bench.py -- SNP-pair EM-LD computations per second on MI355X (BASELINE.json metric).

Default workload = BASELINE.json configs[2], the n_ind = 500 configuration the metric is quoted on:
synthetic binary GL, 100,000 sites x 500 individuals PER GPU, depth-10 generator, positions with gaps
~ UniformInt[1,200] on one chromosome, --max_kb_dist 100 windowed, --extend_out records.  A "step" is one pass of the
pair kernels over every pair of the rank's rows, inputs already resident in HBM, records left in HBM
(ngsld_run_device + ngsld_finish_device: the exact-order replay of whatever the kernels flagged is inside the step).
N GPUs: rank 0 generates the matrix and broadcasts it once over RCCL (outside the timed region), ranks take
contiguous row ranges balanced by pair count and never communicate on the compute path.

--config picks another BASELINE configuration (same code path, same JSON line):
  c1  configs[1]   5,000 x 100 all pairs                       weak (each GPU its own 5,000 sites)
  c2  configs[2]   100,000 x 500, 100 kb window (default)      weak (N x 100,000 sites)
  c3  configs[3]   50,000 x 1,000 all pairs, 1.25e9 pairs      STRONG: the fixed problem sharded over N GPUs
  c4  configs[4]   1,000,000 x 2,000, 500 kb window, 1 kb gaps STRONG (48 GB matrix; --sites scales it down)

Prints ONE JSON line on rank 0 (contract in the task statement).  Beside the contract's fields:
  value_host_resident  SURVEY §8(d)'s metric to the letter: pairs / wall time from the first kernel launch to the last
                       record resident in HOST memory (ngsld_run: kernel || D2H into pinned buffers || sink), all ranks
  e2e_file_to_tsv_s    N = 1, configs[2]: the drop-in binary, binary GL file in -> TSV text out (to /dev/null)
  roofline             the pair kernel against the HBM roofline on ALGORITHMIC bytes ((48 n_ind + 72) B per pair, as
                       if every pair streamed both sites from HBM), kernel time from HIP events on the launch stream;
                       `traffic` is the HBM traffic measured with rocprofv3 PMC counters in a separate profiling run
                       (profiles/), not in this run; `fp64_valu` is the roofline that actually binds the kernel
  cpu_baseline         the host, on the threads this process may really use: kind "reference" = the reference's OWN program
                       (oracle/_ref: ngsLD.cpp's main + calc_pair_LD compiled whole minus the statements that need GSL) from a
                       file of the sample's first sites to TSV; beside it `port` = the oracle restatement (whole per-pair path,
                       with the whole-sample parity check against the GPU's records) and `reference` = the reference's
                       compiled haplo_freq alone (the EM only), each also on one thread
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X FP64 vector peak
STD_BYTES, EXT_BYTES = 32, 40

CONFIGS = {  # sites (per GPU when weak, total when strong), ind, max_kb, max_gap, scaling, steps, warmup
    "c1": dict(sites=5_000, ind=100, max_kb=0, max_gap=200, scaling="weak", steps=5, warmup=2,
               name="BASELINE configs[1]"),
    "c2": dict(sites=100_000, ind=500, max_kb=100, max_gap=200, scaling="weak", steps=3, warmup=1,
               name="BASELINE configs[2]"),
    "c3": dict(sites=50_000, ind=1_000, max_kb=0, max_gap=200, scaling="strong", steps=1, warmup=0,
               name="BASELINE configs[3]"),
    "c4": dict(sites=1_000_000, ind=2_000, max_kb=500, max_gap=2000, scaling="strong", steps=1, warmup=0,
               name="BASELINE configs[4]"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--sites", type=int, default=None, help="sites per GPU (weak) / in total (strong)")
    ap.add_argument("--ind", type=int, default=None)
    ap.add_argument("--max-kb", type=int, default=None)
    ap.add_argument("--depth", type=float, default=10.0)
    ap.add_argument("--max-gap", type=int, default=None, help="site gaps ~ UniformInt[1, max-gap]")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--pairs-per-item", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of each cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-sink", action="store_true", help="skip the host-resident leg (value_host_resident)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file -> TSV leg of the drop-in binary")
    ap.add_argument("--no-unfiltered", action="store_true",
                    help="skip the `unfiltered_input` leg (the same pass on 10,000 sites that are NOT SNP-called)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` leg (configs[1] whole, one rank's shard of configs[3] and configs[4], a pass each)")
    ap.add_argument("--ignore-miss", action="store_true", help="run the --ignore_miss_data kernels (not the headline config)")
    ap.add_argument("--rnd-sample", type=float, default=1.0, help="--rnd_sample of the plan (not the headline config)")
    ap.add_argument("--hard-calls", action="store_true",
                    help="hard-call the synthetic likelihoods (argmax -> 1/0/0 triples): the genotype-combination kernel (not the headline config)")
    ap.add_argument("--mono-frac", type=float, default=0.0,
                    help="that share of the sites monomorphic in the population: a matrix that is NOT SNP-called (the reference's "
                         "README.md:73); not the headline config")
    ap.add_argument("--sfs", action="store_true",
                    help="site frequencies log-uniform in [0.001, 0.5] instead of U(0.05, 0.5); not the headline config")
    ap.add_argument("--traffic-json", default=os.path.join(REPO, "profiles", "hbm_traffic.json"),
                    help="fallback for roofline.traffic when rocprofv3 is not available")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not re-run one launch under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for roofline.traffic")
    ap.add_argument("--balance", choices=["pairs", "work"], default="pairs",
                    help="how rows are dealt to the ranks: equal pair counts (default), or equal ESTIMATED work -- a row's "
                         "pairs x its executed EM iterations per pair, measured on every ~100th row before the timed region")
    ap.add_argument("--native-multi", action="store_true",
                    help="ONE process: the product's own multi-device path (ngsld_run_multi, `ngsLD --devices ...`) on --gpus "
                         "parts -- one part per device, or all on the devices there are -- with a discarding sink")
    a = ap.parse_args()
    preset = CONFIGS[a.config]
    a.custom = any(getattr(a, k) is not None for k in ("sites", "ind", "max_kb", "max_gap", "scaling"))
    for k in ("steps", "warmup", "sites", "ind", "max_kb", "max_gap", "scaling"):
        if getattr(a, k) is None:
            setattr(a, k, preset[k])
    return a


def host_cpus() -> dict:
    """The threads this process can really run on: the affinity mask, cut by the cgroup CPU quota when there is one
    (os.cpu_count() reports the machine, not the lease)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    usable = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return {"os_cpu_count": os.cpu_count(), "sched_affinity": aff, "cgroup_cpu_quota": quota, "threads_used": usable}


def cpu_baseline(raw_head: np.ndarray, pos_dist_head: np.ndarray, max_kb: int, target_s: float, gpu_rows=None) -> dict:
    """Rows [0, R) of the bench matrix on the host, R sized for ~target_s per leg.
    gpu_rows(R) -> (#pairs, sum of finite r2, sum of executed EM iterations) of the GPU records of the same rows:
    a whole-sample parity check (the iteration totals must be EQUAL, i.e. no convergence-threshold flip)."""
    from oracle import orc
    cpus = host_cpus()
    nt = cpus["threads_used"]
    n_have = raw_head.shape[0]
    o = orc.Oracle(raw_head, pos_dist_head, max_kb_dist=max_kb, n_threads=nt)
    ends = o.row_ends().astype(np.int64)
    halo = int((ends - np.arange(n_have)).max())
    usable = max(1, n_have - halo)                      # rows whose whole window lies inside the sample
    if max_kb == 0:                                     # all pairs: the sample is the sub-matrix's own pair space
        usable, gpu_rows = n_have - 1, None

    def timed(fn, threads, seconds):
        cal_rows = min(usable, max(threads, 16))
        t0 = time.perf_counter()
        fn(cal_rows, threads)
        t_cal = max(time.perf_counter() - t0, 1e-6)
        rows = int(min(usable, max(cal_rows, cal_rows * seconds / t_cal)))
        t0 = time.perf_counter()
        res = fn(rows, threads)
        return rows, res, time.perf_counter() - t0

    def port(rows, threads):
        o.p.n_threads = threads
        return o.bench(0, rows)                          # (#pairs, checksum, executed iterations)

    rows, (n, chk, iters), dt = timed(port, nt, target_s)
    out = {"value": n / dt, "unit": "pairs/s", "cores": nt, "kind": "port",
           "sample": f"rows 0..{rows} of the same matrix ({n} pairs, {dt:.1f} s, mean executed EM iterations "
                     f"{iters / max(n, 1):.2f}); oracle/liborc.so (the whole per-pair path: pearson_r, haplo_freq, D/D'/r2), "
                     f"{nt} pthreads",
           "host": cpus}
    rows1, (n1, _, _), dt1 = timed(port, 1, target_s / 3)
    out["one_thread"] = {"value": n1 / dt1, "pairs": int(n1), "seconds": round(dt1, 2)}
    out["threads_x_one_thread"] = nt * n1 / dt1          # what perfect scaling of the 1-thread rate would give
    if gpu_rows is not None:
        gn, gsum, giters = gpu_rows(rows)
        out["parity_on_sample"] = {"pairs": int(gn), "pairs_equal": bool(gn == n),
                                   "executed_iterations_cpu": int(iters), "executed_iterations_gpu": int(giters),
                                   "executed_iterations_equal": bool(giters == iters),
                                   "abs_diff_sum_r2": abs(gsum - chk), "sum_r2_cpu": chk}
    if o.bench_reference(0, 1, 1) is not None:          # the reference's own compiled EM (oracle/_ref travelled here)
        def ref(rows, threads):
            return o.bench_reference(0, rows, threads)
        rrows, (rn, rit, _), rdt = timed(ref, nt, target_s)
        r1rows, (rn1, _, _), rdt1 = timed(ref, 1, target_s / 3)
        out["reference"] = {"value": rn / rdt, "unit": "pairs/s", "cores": nt, "kind": "reference-subset",
                            "sample": f"rows 0..{rrows} of the same matrix ({rn} pairs, {rdt:.1f} s, mean executed EM "
                                      f"iterations {rit / max(rn, 1):.2f}): the reference's own haplo_freq "
                                      f"(shared/gen_func.cpp:1027, compiled from /root/reference into oracle/_ref) on {nt} "
                                      "pthreads, rows dealt round-robin as its thread pool deals calc_pair_LD jobs; pearson_r "
                                      "(GSL) and the fprintf are not in it, so this bounds the reference's rate from above",
                            "one_thread": {"value": rn1 / rdt1, "pairs": int(rn1), "seconds": round(rdt1, 2)},
                            "executed_iterations_equal_port": bool(rrows != rows or rit == iters)}
    else:
        out["reference"] = None
    # ---- the reference's own PROGRAM (oracle/_ref: ngsLD.cpp's main + calc_pair_LD compiled whole minus the statements that
    # need GSL, oracle/build_ref.sh): file -> reader -> est_maf -> thread pool -> calc_pair_LD -> fprintf, --n_threads = the
    # threads this process may use, on a file of the sample's first sites.  pearson_r (gsl_stats_correlation, ~4 % of its
    # per-pair time by SURVEY 8a) is the one thing it does not do: the column prints -nan.
    rp = reference_program(raw_head, pos_dist_head, max_kb, min(n_have, rows + halo), nt)
    out["reference_program"] = rp
    if rp and "value" in rp:   # the baseline of the line is the reference's own program where it can run; the port stays beside it
        out["port"] = {k: out[k] for k in ("value", "unit", "cores", "kind", "sample")}
        out.update(value=rp["value"], cores=rp["cores"], kind="reference", sample=rp["sample"],
                   baseline_definition=rp.get("baseline_definition"), value_without_pearson=rp.get("value_without_pearson"))
    return out


_REF_CHILD = r"""
import ctypes as C, sys, time
sys.path.insert(0, %r)
from oracle import orc
R = orc.ref()
argv = [b"ngsLD"] + [a.encode() for a in sys.argv[1:]]
arr = (C.c_char_p * (len(argv) + 1))(*argv, None)
R.ref_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
t0 = time.perf_counter()
rc = R.ref_main(len(argv), arr)
print("REF_MAIN_SECONDS", time.perf_counter() - t0, flush=True)
sys.exit(rc)
""" % REPO


def reference_program(raw_head: np.ndarray, pos_dist_head: np.ndarray, max_kb: int, n_file: int, threads: int) -> dict | None:
    from oracle import orc
    R = orc.ref()
    if R is None or not hasattr(R, "ref_main") or n_file < 2:
        return None
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as d:
        g, p = os.path.join(d, "sample.glf"), os.path.join(d, "sample.pos")
        np.ascontiguousarray(raw_head[:n_file]).tofile(g)
        pos = np.cumsum(np.where(np.isfinite(pos_dist_head[:n_file]), pos_dist_head[:n_file], 0.0)).astype(np.int64)
        with open(p, "w") as fh:
            fh.write("".join(f"chr1\t{int(x)}\n" for x in pos))
        oo = orc.Oracle(raw_head[:n_file], pos_dist_head[:n_file].copy(), max_kb_dist=max_kb, n_threads=threads)
        n_pairs = oo.count()
        cmd = [sys.executable, "-c", _REF_CHILD, "--geno", g, "--n_ind", str(raw_head.shape[1]), "--n_sites", str(n_file),
               "--max_kb_dist", str(max_kb), "--extend_out", "--n_threads", str(threads), "--verbose", "0", "--out", "/dev/null"]
        if max_kb > 0:
            cmd += ["--pos", p]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        except subprocess.TimeoutExpired:
            return {"error": "timeout"}
        sec = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("REF_MAIN_SECONDS")]
        if r.returncode != 0 or not sec:
            return {"error": r.stderr[-300:]}
        dt = float(sec[0])
        # the one thing that program leaves out, pearson_r (GSL's recurrence restated in the oracle), over the same pairs on the
        # same threads: added to its time, so that the baseline is what the reference's real binary would take
        t0 = time.perf_counter()
        pn, _ = oo.bench_pearson(0, n_file)
        dt_p = time.perf_counter() - t0
        assert pn == n_pairs
    return {"value": n_pairs / (dt + dt_p), "unit": "pairs/s", "cores": threads, "kind": "reference",
            # definition 2 (round 5 on): program time + pearson_r's; rounds 1-4 reported the program alone (definition 1 =
            # `value_without_pearson` here): speed-ups are comparable across rounds on that field only.  pearson_seconds is an
            # UPPER bound on what the real binary would add -- timed after the program, not inside its thread pool and I/O overlap
            "baseline_definition": 2,
            "value_without_pearson": n_pairs / dt, "program_seconds": round(dt, 2), "pearson_seconds": round(dt_p, 2),
            "sample": f"the first {n_file} sites of the same matrix as a binary GL file ({n_pairs} pairs, --extend_out, "
                      f"--n_threads {threads}): {dt:.1f} s from file to TSV on /dev/null for ngsLD.cpp's own main() and calc_pair_LD "
                      "compiled from /root/reference into oracle/_ref minus the statements that need GSL -- reader, est_maf, thread "
                      f"pool, walk, haplo_freq, D / D' / r2, fprintf -- PLUS {dt_p:.1f} s for the one thing that program leaves out, "
                      "pearson_r (gsl_stats_correlation: GSL's long double recurrence restated in oracle/ngsld_oracle.c), over the "
                      f"same pairs on the same {threads} threads; `value_without_pearson` is the program alone"}


def e2e_file_to_tsv(raw_dev, n_sites: int, n_ind: int, chrs, pos, max_kb: int, threads: int) -> dict | None:
    """The drop-in binary end to end: binary GL file + pos file in, extended TSV out (to /dev/null)."""
    from ngsld_amd import capi, synth
    if not os.path.exists(capi.CLI_PATH):
        return None
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    need = n_sites * n_ind * 24
    try:
        st = os.statvfs(base or tempfile.gettempdir())
        if st.f_bavail * st.f_frsize < need * 1.2:
            return {"skipped": "not enough scratch space for the input file"}
    except OSError:
        pass
    with tempfile.TemporaryDirectory(dir=base) as d:
        g, p = os.path.join(d, "in.glf"), os.path.join(d, "in.pos")
        with open(g, "wb") as fh:
            step = max(1, (256 << 20) // (n_ind * 24))
            for lo in range(0, n_sites, step):
                fh.write(raw_dev[lo:lo + step].cpu().numpy().tobytes())
        synth.write_pos(p, chrs, pos)
        cmd = [capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p,
               "--max_kb_dist", str(max_kb), "--extend_out", "--n_threads", str(threads), "--verbose", "0",
               "--out", "/dev/null"]
        runs = []
        for k in range(2):
            if k:  # (a process started while the driver is still tearing the previous one down pays ~0.15 s of runtime
                time.sleep(0.5)  # initialisation for it, profiles/r04/probe_init.txt: a user's run does not follow another by 0 ms)
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True)
            runs.append(time.perf_counter() - t0)
            if r.returncode != 0:
                return {"error": r.stderr[-300:]}
    return {"seconds": min(runs), "runs_s": [round(x, 4) for x in runs], "n_threads": threads,
            "what": "ngsld_amd/bin/ngsLD: file read, H2D, per-site prep, plan, pair kernels, device-side TSV, D2H of the text, "
                    "write to /dev/null (best of 2 runs half a second apart)"}


def measure_traffic(args) -> dict | None:
    """roofline.traffic, measured in THIS run: one launch of the same workload under `rocprofv3 --pmc FETCH_SIZE` and one
    under `--pmc WRITE_SIZE` (the two do not fit one pass; counters never combined with trace domains), outside the timed
    region, after the engine of the timed run is closed.  Bytes as MI355X_MICROARCH.md (HBM section) prescribes:
    FETCH_SIZE is in KiB and reports half the bytes of a wide (16 B per lane) coalesced stream on gfx950 -- the pair kernels'
    site copies are global_load_lds_dwordx4 -- so it is doubled; WRITE_SIZE in KiB as it comes.  The counters sit on the L2's
    fabric side: Infinity-Cache hits are counted, so this is an upper bound on HBM bytes."""
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    child = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--config", args.config, "--steps", "1",
             "--warmup", "0", "--no-cpu", "--no-sink", "--no-e2e", "--no-traffic", "--no-unfiltered", "--no-other-configs", "--sites", str(args.sites), "--ind",
             str(args.ind), "--max-kb", str(args.max_kb), "--max-gap", str(args.max_gap), "--scaling", args.scaling,
             "--depth", repr(args.depth), "--seed", str(args.seed), "--rnd-sample", repr(args.rnd_sample),
             "--mono-frac", repr(args.mono_frac)]
    if args.sfs:
        child.append("--sfs")
    if args.ignore_miss:
        child.append("--ignore-miss")
    if args.hard_calls:
        child.append("--hard-calls")
    if args.pairs_per_item:
        child += ["--pairs-per-item", str(args.pairs_per_item)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    got, names = {}, set()
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            try:
                r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "run", "--"] + child,
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
            except (OSError, subprocess.TimeoutExpired) as e:
                return {"error": f"rocprofv3 --pmc {counter}: {e!r}"}
            if r.returncode != 0:
                return {"error": f"rocprofv3 --pmc {counter} exited {r.returncode}: {r.stderr[-300:]}"}
            per_dispatch = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if "pair_ld" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                            key = (f, row.get("Dispatch_Id"))
                            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
                            names.add(row["Kernel_Name"].split("(")[0][:80])
            if not per_dispatch:
                return {"error": f"no pair-kernel rows in the {counter} pass"}
            got[counter] = sum(per_dispatch.values())          # the step's launches (one, unless the grid was cut)
    fetch, write = got["FETCH_SIZE"] * 1024.0 * 2.0, got["WRITE_SIZE"] * 1024.0
    return {"bytes_per_step": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write,
            "kernels": sorted(names), "seconds": round(time.perf_counter() - t0, 1),
            "source": "measured_this_run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two single-step passes of this "
                      "workload outside the timed region; FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-stream correction, "
                      "MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB x 1024; fabric-side counters: Infinity-Cache hits "
                      "included, an upper bound on HBM bytes"}


def unfiltered_input_leg(n_sites: int, n_ind: int, max_kb: int, max_gap: int, depth: float, dev_index: int, full_sites: int = 0) -> dict:
    """The same pass on a matrix that is NOT SNP-called (the reference's README.md:73: "these comparisons will show up as nan or
    inf"; its own examples/test.sh feeds such input): 20 % of the sites monomorphic in the population and, a second matrix, site
    frequencies log-uniform in [0.001, 0.5].  A third of the pairs (a fourteenth) are then pairs the reference's own rounding
    decides -- flagged by the pair kernels and replayed in the reference's operation order, on the DEVICE (ld_replay_lkl.hip) once the
    exact store is built (host libm, once per matrix: inside `first_pass_s`).  Small (n_sites sites) as in round 5's line; `full_sites`
    adds the first twin at the headline's own size (= `bench.py --mono-frac 0.2`)."""
    import torch
    from ngsld_amd import capi, shard, synth
    dev = torch.device("cuda", dev_index)
    out = {"n_sites": n_sites, "n_ind": n_ind, "max_kb_dist": max_kb}

    def measure(m_sites: int, kw: dict) -> dict:
        chrs, pos = synth.make_positions(m_sites, 33, max_gap=max_gap)
        pd = shard.pos_dist_from_positions(chrs, pos)
        raw = synth.make_gl_torch(m_sites, n_ind, 33, dev, depth=depth, **kw)
        host = raw.cpu().numpy()
        eng = capi.Engine(dev_index)
        try:
            eng.set_geno_raw(raw.data_ptr(), n_sites=m_sites, n_ind=n_ind)
            eng.set_replay_source(host)
            eng.set_pos_dist(pd)
            n_pairs = eng.plan(max_kb_dist=max_kb, extend_out=True)
            d_std = torch.empty(max(n_pairs, 1) * STD_BYTES, dtype=torch.uint8, device=dev)
            d_ext = torch.empty(max(n_pairs, 1) * EXT_BYTES, dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream().cuda_stream

            def one():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.run_device(0, m_sites, d_std.data_ptr(), d_ext.data_ptr(), st)
                eng.finish_device()
                torch.cuda.synchronize()
                return time.perf_counter() - t0
            first = one()
            best = min(one() for _ in range(3))
            info = eng.replay_info()
            eng.set_replay(False)
            eng.plan(max_kb_dist=max_kb, extend_out=True)
            one()
            off = min(one() for _ in range(2))
            return {"pairs": n_pairs, "value": n_pairs / best, "unit": "pairs/s", "ms_per_pass": best * 1e3,
                    "first_pass_s": round(first, 4), "value_replay_off": n_pairs / off, "ratio_to_replay_off": round(off / best, 4),
                    "replay": info}
        finally:
            eng.close()
            del raw, host
            torch.cuda.empty_cache()

    for name, kw in (("mono_frac_0.2", {"mono_frac": 0.2}), ("sfs", {"sfs": True})):
        out[name] = measure(n_sites, kw)
    if full_sites:
        # the same twin at the HEADLINE's size (configs[2]'s 100,000 sites): a launch of 3e6 flagged pairs is a dozen pairs a lane
        # and pays the lane kernel's ramp and tail (+6 ms of 67); this is the rate a job of the headline's size sees
        out["mono_frac_0.2_full_size"] = dict(measure(full_sites, {"mono_frac": 0.2}), n_sites=full_sites)
    return out


def other_configs_leg(dev_index: int, depth: float, seed: int, world: int = 8) -> dict:
    """The other BASELINE configurations in front of the driver, one pass each on this one device, outside the timed region:
    configs[1] at full size (5,000 x 100, all 12,497,500 pairs) and ONE rank's shard of the two 8-GPU configurations -- the rows
    shard.split_rows deals to rank world // 2 of `world`, with the sites of their windows -- configs[3] (50,000 x 1,000, all
    pairs) and configs[4] (1,000,000 x 2,000, 500 kb: the shard's own slab of sites is generated, 1 / world of the genome).  Per
    configuration: pairs/s of the pass (ngsld_run_device + ngsld_finish_device, replay inside), the pair kernel's launch time
    (HIP events on the launch stream), roofline.frac on algorithmic bytes, the FP64 VALU fraction (the binding roofline), and
    the wrap-around checksum of every record word."""
    import torch
    from ngsld_amd import capi, shard, synth
    dev = torch.device("cuda", dev_index)
    out = {}
    for key in ("c1", "c3", "c4"):
        cfg = CONFIGS[key]
        n_ind, max_kb = cfg["ind"], cfg["max_kb"]
        t_all = time.perf_counter()
        if key == "c1":
            n_sites, lo, hi, slab_lo = cfg["sites"], 0, cfg["sites"], 0
            chrs, pos = synth.make_positions(n_sites, seed, max_gap=cfg["max_gap"])
            pd = shard.pos_dist_from_positions(chrs, pos)
            raw = synth.make_gl_torch(n_sites, n_ind, seed, dev, depth=depth)
            share = "the whole configuration"
        else:
            total = cfg["sites"]
            chrs, pos = synth.make_positions(total, seed, max_gap=cfg["max_gap"])
            pd_all = shard.pos_dist_from_positions(chrs, pos)
            row_end = shard.row_ends(pd_all, max_kb, 0)
            counts = row_end - (np.arange(total, dtype=np.int64) + 1)
            lo, hi = shard.split_rows(counts, world)[world // 2]
            slab_lo, slab_hi = shard.slab_for_rows(row_end, lo, hi)
            n_sites = int(slab_hi - slab_lo)
            pd = pd_all[slab_lo:slab_hi].copy()
            # (the slab's own sites: a matrix of the slab's size with the configuration's generator -- the other ranks' sites are
            # never looked at by this rank's rows)
            raw = synth.make_gl_torch(n_sites, n_ind, seed + 1, dev, depth=depth)
            share = f"rank {world // 2} of {world}: rows [{int(lo)}, {int(hi)}) of {total}, sites [{int(slab_lo)}, {int(slab_hi)})"
        eng = capi.Engine(dev_index)
        try:
            eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
            if raw.numel() * 8 <= (2 << 30):
                host = raw.cpu().numpy()
                eng.set_replay_source(host)
            del raw
            torch.cuda.empty_cache()
            eng.set_pos_dist(pd)
            eng.plan(max_kb_dist=max_kb, extend_out=True)
            row_off, _ = eng.plan_rows()
            r0, r1 = int(lo - slab_lo), int(hi - slab_lo)
            n_pairs = int(row_off[r1] - row_off[r0])
            d_std = torch.empty(max(n_pairs, 1) * STD_BYTES, dtype=torch.uint8, device=dev)
            d_ext = torch.empty(max(n_pairs, 1) * EXT_BYTES, dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream().cuda_stream

            def one():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.run_device(r0, r1, d_std.data_ptr(), d_ext.data_ptr(), st)
                eng.finish_device()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                ms, nl, _ = eng.last_kernel_time()
                return dt, ms, nl
            one()
            passes = [one() for _ in range(3 if key == "c1" else 1)]
            dt, ms, nl = min(passes)
            ext_i32 = d_ext.view(torch.int32).view(-1, EXT_BYTES // 4)
            mean_exec = float(torch.clamp(ext_i32[:n_pairs, 9].to(torch.float64) + 1, max=100).mean()) if n_pairs else 0.0
            words = int(d_std.view(torch.int64)[:n_pairs * (STD_BYTES // 8)].sum()) + int(d_ext.view(torch.int64)[:n_pairs * (EXT_BYTES // 8)].sum())
            bytes_pair = 48 * n_ind + STD_BYTES + EXT_BYTES
            kernel_s = ms / 1e3
            tflops = 2.0 * n_pairs * n_ind * mean_exec * 20.0 / kernel_s / 1e12
            out[key] = {"workload": f"{cfg['name']}: {cfg['sites']} sites x {n_ind} ind, --max_kb_dist {max_kb}; {share}",
                        "pairs": n_pairs, "value": n_pairs / dt, "unit": "pairs/s", "ms_per_pass": dt * 1e3,
                        "kernel": eng.pair_kernel(), "kernel_ms_per_pass": ms, "kernel_launches": nl,
                        "mean_executed_em_iterations": round(mean_exec, 3),
                        "roofline": {"bound": "hbm", "algorithmic_bytes_per_pair": bytes_pair,
                                     "achieved": bytes_pair * n_pairs / kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": bytes_pair * n_pairs / kernel_s / 1e9 / HBM_PEAK_GBS},
                        "fp64_valu": {"achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS},
                        "records_checksum_u64": words % (1 << 64), "replay": eng.replay_info(),
                        "leg_seconds": None}
            del d_std, d_ext
        finally:
            eng.close()
        torch.cuda.empty_cache()
        out[key]["leg_seconds"] = round(time.perf_counter() - t_all, 2)
    return out


def estimate_row_work(raw, pos_dist, args, dev_index: int, n_sites: int, n_ind: int) -> tuple[np.ndarray, dict]:
    """Estimated work per row = its pairs x (executed EM iterations per pair + the per-pair prologue, ~1.1 iterations' worth):
    every ~100th row is computed for real on this device (the whole matrix is here: this runs before the slabs are cut) and
    the per-pair figure interpolated in between.  Not timed."""
    import torch
    from ngsld_amd import capi
    t0 = time.perf_counter()
    eng = capi.Engine(dev_index)
    try:
        eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind, ignore_miss_data=args.ignore_miss)
        eng.set_replay(False)
        eng.set_pos_dist(pos_dist)
        eng.plan(max_kb_dist=args.max_kb, extend_out=True, ignore_miss_data=args.ignore_miss, rnd_sample=args.rnd_sample, seed=12345)
        row_off, _ = eng.plan_rows()
        pairs = np.diff(row_off.astype(np.int64))
        m = int(min(n_sites, max(64, n_sites // 100)))
        rows = np.unique(np.linspace(0, max(n_sites - 2, 0), m).astype(np.int64))
        cap = int(pairs[rows].max()) if len(rows) else 1
        dev = torch.device("cuda", dev_index)
        d_std = torch.empty(max(cap, 1) * STD_BYTES, dtype=torch.uint8, device=dev)
        d_ext = torch.empty(max(cap, 1) * EXT_BYTES, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        per_pair = np.zeros(len(rows))
        for k, r in enumerate(rows):
            n = int(pairs[r])
            if n == 0:
                per_pair[k] = np.nan
                continue
            eng.run_device(int(r), int(r) + 1, d_std.data_ptr(), d_ext.data_ptr(), stream)
            eng.finish_device()
            it = d_ext.view(torch.int32).view(-1, EXT_BYTES // 4)[:n, 9].to(torch.float64)
            per_pair[k] = float(torch.clamp(it + 1, max=100).mean()) + 1.1
    finally:
        eng.close()
    ok = np.isfinite(per_pair)
    if not ok.any():
        return pairs.astype(np.float64), {"sampled_rows": 0}
    est = np.interp(np.arange(n_sites), rows[ok], per_pair[ok])
    info = {"sampled_rows": int(ok.sum()), "per_pair_min_max": [float(per_pair[ok].min()), float(per_pair[ok].max())],
            "seconds": round(time.perf_counter() - t0, 2)}
    return pairs.astype(np.float64) * est, info


def native_multi(args) -> None:
    """bench.py --native-multi: the product's OWN multi-device path in one process -- ngsld_run_multi, what `ngsLD --devices
    0-7` runs (multi.hip: one host thread + context per device, per-part slab upload or ONE RCCL broadcast, per-part sink).
    A step is the whole call: upload / broadcast, per-site prep, plan and the pair kernels of every part, the records handed
    to a discarding sink in host memory.  Not the line the driver takes (that is the process-per-GPU path below)."""
    import torch
    from ngsld_amd import capi, shard, synth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda is not available (no CPU fallback)")
    torch.cuda.set_device(0)
    n_dev = max(1, capi.device_count())
    torch.cuda.set_device(0)                      # (counting devices walks hipSetDevice up to the first failure)
    parts = max(1, args.gpus)
    devices = [k % n_dev for k in range(parts)]
    strong = args.scaling == "strong"
    n_sites = args.sites if strong else args.sites * parts
    n_ind = args.ind
    dev = torch.device("cuda", 0)
    chrs, pos = synth.make_positions(n_sites, args.seed, max_gap=args.max_gap, n_chr=1)
    pos_dist = shard.pos_dist_from_positions(chrs, pos)
    raw = synth.make_gl_torch(n_sites, n_ind, args.seed, dev, depth=args.depth, mono_frac=args.mono_frac, sfs=args.sfs)
    if args.hard_calls:
        raw = torch.nn.functional.one_hot(raw.argmax(dim=2), 3).to(torch.float64)
    raw = raw.cpu().numpy()
    torch.cuda.empty_cache()
    kw = dict(max_kb_dist=args.max_kb, extend_out=True, ignore_miss_data=args.ignore_miss, rnd_sample=args.rnd_sample, seed=12345)

    def run(check: bool):
        acc = {"pairs": 0, "iters": 0, "words": 0}
        import threading
        lock = threading.Lock()

        def sink(part, n_pairs, std, ext):
            if check:
                it = np.minimum(ext["n_iter"].astype(np.int64) + 1, 100).sum()
                w = (int(std.view(np.int64).sum()) + int(ext.view(np.int64).sum())) % (1 << 64)
                with lock:
                    acc["pairs"] += n_pairs
                    acc["iters"] += int(it)
                    acc["words"] = (acc["words"] + w) % (1 << 64)
            return 0
        t0 = time.perf_counter()
        per = capi.run_multi_sink(raw, pos_dist, devices, sink if check else None, **kw)
        return time.perf_counter() - t0, per, acc

    _, per, acc = run(True)                       # untimed: what the parts computed, as partition-invariant checksums
    for _ in range(args.warmup):
        run(False)
    times = [run(False)[0] for _ in range(args.steps)]
    total_pairs = int(sum(per))
    elapsed = float(sum(times))
    out = {
        "metric": f"SNP-pair EM-LD computations/sec @ n_ind={n_ind}", "value": total_pairs * args.steps / elapsed, "unit": "pairs/s",
        "n_gpus": parts, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"synthetic binary GL, {args.sites} sites{'' if strong else '/GPU'} x {n_ind} ind, depth {args.depth:g}, "
                               f"--max_kb_dist {args.max_kb} {'windowed' if args.max_kb else 'all pairs'}, --extend_out"
                               + (", hard-called" if args.hard_calls else ""),
                   "n_sites_total": n_sites, "pairs_per_step": total_pairs,
                   "parallelism": "one process, ngsld_run_multi (one host thread + context per part, records to a discarding "
                                  "sink in host memory); a step is the WHOLE call: matrix distribution, per-site prep, plan, pair "
                                  "kernels -- not comparable with the default line, whose inputs are resident before the step",
                   "devices": devices, "devices_visible": n_dev,
                   "matrix_distribution": capi.multi_last_distribution(),
                   "pairs_per_part": [int(v) for v in per],
                   "executed_iterations_total": acc["iters"], "records_checksum_u64": acc["words"],
                   "step_seconds": [round(t, 4) for t in times]},
    }
    print(json.dumps(out), flush=True)


def main():
    args = parse()
    if args.native_multi:
        native_multi(args)
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    from ngsld_amd import capi, shard, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda is not available (no CPU fallback)")
    # NGSLD_BENCH_ONE_DEVICE=1 (debug only): put every rank on GPU 0 with the gloo backend, so the N > 1 code
    # path can be exercised on a 1-GPU box; the driver's runs use one GPU per rank over RCCL (backend nccl).
    one_dev = os.environ.get("NGSLD_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_dev else local_rank
    if world > 1 and not one_dev and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py --gpus {world}: rank {rank} (local rank {local_rank}) has no device of its own -- "
                         f"{torch.cuda.device_count()} visible; one GPU per rank, or NGSLD_BENCH_ONE_DEVICE=1 for a dry run of the "
                         "N > 1 path on one device (not a scaling measurement)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # every rank adds 1: what the collective library itself saw of the job (the SCALE record checks itself)
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())

    strong = args.scaling == "strong"
    n_sites = args.sites if strong else args.sites * world
    n_ind = args.ind

    # ---- positions (host, identical on every rank) and the row shards ----
    chrs, pos = synth.make_positions(n_sites, args.seed, max_gap=args.max_gap, n_chr=1)
    pos_dist = shard.pos_dist_from_positions(chrs, pos)
    row_end = shard.row_ends(pos_dist, args.max_kb, 0)
    counts = row_end - (np.arange(n_sites, dtype=np.int64) + 1)

    # ---- the GL matrix: rank 0 generates, one RCCL broadcast distributes (not timed) ----
    t_gen = time.perf_counter()
    if rank == 0:
        raw = synth.make_gl_torch(n_sites, n_ind, args.seed, dev, depth=args.depth, mono_frac=args.mono_frac, sfs=args.sfs)
    else:
        raw = torch.empty((n_sites, n_ind, 3), dtype=torch.float64, device=dev)
    if args.hard_calls and rank == 0:
        raw = torch.nn.functional.one_hot(raw.argmax(dim=2), 3).to(torch.float64).contiguous()
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    t_bc = time.perf_counter()
    shard.broadcast_matrix(raw, src=0)
    torch.cuda.synchronize()
    t_bc = time.perf_counter() - t_bc

    # ---- which rows are whose: equal pair counts, or (--balance work) equal estimated work -- rank 0 measures every ~100th
    # row on the whole matrix (everybody holds it at this point) and hands the estimate to the others ----
    balance_info = None
    if args.balance == "work" and world > 1:
        est = torch.zeros(n_sites, dtype=torch.float64, device=dev)
        if rank == 0:
            w, balance_info = estimate_row_work(raw, pos_dist, args, dev_index, n_sites, n_ind)
            est.copy_(torch.from_numpy(w))
        dist.broadcast(est, src=0)
        lo, hi = shard.split_rows_weighted(est.cpu().numpy(), world)[rank]
    else:
        lo, hi = shard.split_rows(counts, world)[rank]
    slab_lo, slab_hi = shard.slab_for_rows(row_end, lo, hi)

    # ---- engine: this rank's slab (its rows + halo) goes through the device prep kernel ----
    eng = capi.Engine(dev_index)
    slab = raw[slab_lo:slab_hi]
    torch.cuda.synchronize()
    t_prep = time.perf_counter()
    eng.set_geno_raw(slab.data_ptr(), n_sites=slab_hi - slab_lo, n_ind=n_ind,
                     ignore_miss_data=args.ignore_miss)                          # per-site prep kernel (one-off)
    t_prep = time.perf_counter() - t_prep
    # the exact-order replay reads flagged pairs' sites from host memory (as the drop-in binary does); a slab too large to
    # mirror on the host is read back from the device's planes instead
    if slab.numel() * 8 <= (8 << 30):
        slab_host = slab.cpu().numpy()
        eng.set_replay_source(slab_host)
    local_pd = pos_dist[slab_lo:slab_hi].copy()
    eng.set_pos_dist(local_pd)
    if args.pairs_per_item:
        eng.set_tuning(pairs_per_item=args.pairs_per_item)
    t_plan = time.perf_counter()
    eng.plan(max_kb_dist=args.max_kb, extend_out=True, ignore_miss_data=args.ignore_miss,
             rnd_sample=args.rnd_sample, seed=12345)                                # pair-space plan (one-off)
    t_plan = time.perf_counter() - t_plan
    row_off, _ = eng.plan_rows()
    n_rows = hi - lo
    n_pairs = int(row_off[n_rows] - row_off[0])
    assert args.rnd_sample < 1.0 or n_pairs == int(counts[lo:hi].sum()), "engine plan and host mirror disagree on the pair count"

    uncalled = args.mono_frac > 0 or args.sfs
    headline = world == 1 and not args.ignore_miss and args.rnd_sample >= 1.0 and not args.hard_calls and not uncalled
    raw_head = None
    family = eng.pair_kernel()
    if rank == 0 and headline and not args.no_cpu:
        head = min(n_sites, 12_000 if n_ind <= 500 else 6_000)
        raw_head = raw[:head].cpu().numpy()
    e2e = None
    if rank == 0 and headline and not args.no_e2e and args.config == "c2" and not args.custom:
        torch.cuda.synchronize()
        e2e = e2e_file_to_tsv(raw, n_sites, n_ind, chrs, pos, args.max_kb, host_cpus()["threads_used"])
        if e2e is not None and "seconds" in e2e and not args.no_unfiltered:
            # the same through the binary on the un-called twin of the matrix (20 % of the sites monomorphic, README.md:73): text in
            # groups, the flagged third of the pairs replayed on the device (engine_run.hip, run_grouped)
            raw_u = synth.make_gl_torch(n_sites, n_ind, args.seed, dev, depth=args.depth, mono_frac=0.2)
            torch.cuda.synchronize()
            e2e["uncalled_mono_frac_0.2"] = e2e_file_to_tsv(raw_u, n_sites, n_ind, chrs, pos, args.max_kb, host_cpus()["threads_used"])
            del raw_u
            torch.cuda.empty_cache()
    unfiltered = None
    profiled = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCTX")) for k in os.environ)  # (its kernels would land in the profile)
    if rank == 0 and headline and not args.no_unfiltered and args.config == "c2" and not args.custom and not profiled:
        unfiltered = unfiltered_input_leg(10_000, n_ind, args.max_kb, args.max_gap, args.depth, dev_index,
                                          full_sites=args.sites if args.sites >= 50_000 else 0)
    del slab, raw
    torch.cuda.empty_cache()
    other_configs = None
    if rank == 0 and headline and not args.no_other_configs and args.config == "c2" and not args.custom and not profiled:
        other_configs = other_configs_leg(dev_index, args.depth, args.seed)

    d_std = torch.empty(max(n_pairs, 1) * STD_BYTES, dtype=torch.uint8, device=dev)
    d_ext = torch.empty(max(n_pairs, 1) * EXT_BYTES, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    replayed = [0]

    def step():
        eng.run_device(0, n_rows, d_std.data_ptr(), d_ext.data_ptr(), stream)
        eng.finish_device()              # waits for the kernels; exact-order replay of the pairs they flagged
        replayed[0] = eng.replay_stats()[0]

    # the first pass, timed by itself: on a matrix that is not SNP-called it is the one that builds the exact store of the
    # device-side replay (host libm, once per matrix) -- part of the warm-up when there is one, of the timed steps otherwise
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    first_in_warmup = args.warmup > 0
    if first_in_warmup:
        step()
        torch.cuda.synchronize()
    t_first = time.perf_counter() - t_first

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup - (1 if first_in_warmup else 0)):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    kernel_ms, launches = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        ms, nl, _ = eng.last_kernel_time()   # HIP events on the launch stream
        kernel_ms += ms
        launches += nl
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    replay_info = eng.replay_info()
    replay_off = None
    if uncalled and world == 1:          # the same pass with the exact-order replay off: what the replay costs on this input
        eng.set_replay(False)
        eng.plan(max_kb_dist=args.max_kb, extend_out=True, ignore_miss_data=args.ignore_miss, rnd_sample=args.rnd_sample, seed=12345)
        step()
        torch.cuda.synchronize()
        t_off = time.perf_counter()
        step()
        torch.cuda.synchronize()
        t_off = time.perf_counter() - t_off
        replay_off = {"value": n_pairs / t_off, "ms_per_step": t_off * 1e3}
        eng.set_replay(True)
        eng.plan(max_kb_dist=args.max_kb, extend_out=True, ignore_miss_data=args.ignore_miss, rnd_sample=args.rnd_sample, seed=12345)
        step()                            # (the records of the line's checksums are the replayed ones)

    # ---- SURVEY §8(d)'s metric to the letter: last record resident in HOST memory (kernel || D2H || sink) ----
    sink_elapsed, sink_passes = 0.0, max(1, min(args.steps, 3))
    if not args.no_sink:
        eng.run_discard(0, n_rows)                               # warm-up pass: pinned buffers, host copy of the plan
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for _ in range(sink_passes):                             # consecutive passes, timed together like the steps above
            got = eng.run_discard(0, n_rows)
            assert got == n_pairs
        torch.cuda.synchronize()
        barrier()
        sink_elapsed = (time.perf_counter() - t1) / sink_passes

    # ---- aggregate over ranks: MAX time, SUM pairs ----
    stats = torch.tensor([elapsed, kernel_ms / max(launches, 1), float(n_pairs), sink_elapsed, -elapsed, -float(n_pairs)],
                         dtype=torch.float64, device=dev)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed_max, total_pairs, sink_max = float(mx[0]), int(sm[2]), float(mx[3])
        elapsed_min, pairs_min, pairs_max = -float(mx[4]), int(-float(mx[5])), int(mx[2])
    else:
        elapsed_max, total_pairs, sink_max = elapsed, n_pairs, sink_elapsed
        elapsed_min, pairs_min, pairs_max = elapsed, n_pairs, n_pairs

    # mean executed EM iterations (n_iter is the 0-based index of the converging iteration; 100 = cap)
    ext_i32 = d_ext.view(torch.int32).view(-1, EXT_BYTES // 4)
    n_iter = ext_i32[:n_pairs, 9].to(torch.float64)
    mean_exec = float(torch.clamp(n_iter + 1, max=100).mean()) if n_pairs else 0.0

    # ---- what this rank computed, as partition-invariant checksums: the executed-iteration total and the wrap-around sum
    # of every 8-byte word of its records.  Summed over the ranks they must equal a 1-rank run of the same matrix bit for
    # bit -- sharding changes who computes a pair, never its result (tests/test_gpu_multi_ranks.py holds the driver's
    # N > 1 launch to that) ----
    words_std = d_std.view(torch.int64)[:n_pairs * (STD_BYTES // 8)]
    words_ext = d_ext.view(torch.int64)[:n_pairs * (EXT_BYTES // 8)]
    r2_col = d_std.view(torch.float64).view(-1, 4)[:n_pairs, 3]
    mine = {"rank": rank, "rows": [int(lo), int(hi)], "sites_held": [int(slab_lo), int(slab_hi)], "pairs": n_pairs,
            "executed_iterations": int(torch.clamp(ext_i32[:n_pairs, 9].to(torch.int64) + 1, max=100).sum()) if n_pairs else 0,
            "sum_r2_finite": float(r2_col[torch.isfinite(r2_col)].sum()) if n_pairs else 0.0,
            "records_checksum_u64": (int(words_std.sum()) + int(words_ext.sum())) % (1 << 64) if n_pairs else 0,
            "mean_executed_iterations": round(mean_exec, 4), "kernel_ms_per_launch": kernel_ms / max(launches, 1),
            "seconds": elapsed, "device": torch.cuda.get_device_name(dev), "device_index": dev_index,
            "gl_broadcast_s": round(t_bc, 4)}
    rank_records = [mine]
    if world > 1:
        rank_records = [None] * world
        dist.all_gather_object(rank_records, mine)

    if rank == 0:
        value = total_pairs * args.steps / elapsed_max
        bytes_pair = 48 * n_ind + STD_BYTES + EXT_BYTES
        launch_s = (kernel_ms / max(launches, 1)) / 1e3
        pairs_per_launch = n_pairs * args.steps / max(launches, 1)
        achieved = bytes_pair * pairs_per_launch / launch_s / 1e9
        # FP64 view: per individual and executed iteration 9 FMA (s) + 8 FMA (R) + 3 of the shared-reciprocal tree
        dp_ops = pairs_per_launch * n_ind * mean_exec * 20.0
        fp64_tflops = 2.0 * dp_ops / launch_s / 1e12
        preset = CONFIGS[args.config]
        is_preset = not args.custom and not args.hard_calls and not uncalled
        out = {
            "metric": f"SNP-pair EM-LD computations/sec @ n_ind={n_ind}", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic binary GL, {args.sites} sites{'' if strong else '/GPU'} x {n_ind} ind, depth {args.depth:g}, "
                                   f"--max_kb_dist {args.max_kb} {'windowed' if args.max_kb else 'all pairs'}, --extend_out"
                                   + (", hard-called" if args.hard_calls else "")
                                   + (f", NOT SNP-called: {args.mono_frac:g} of the sites monomorphic" if args.mono_frac > 0 else "")
                                   + (", site frequencies log-uniform in [0.001, 0.5]" if args.sfs else "")
                                   + (f" ({preset['name']})" if is_preset else ""),
                       "n_sites_total": n_sites, "pairs_per_step": total_pairs,
                       "mean_executed_em_iterations": round(mean_exec, 3),
                       "parallelism": f"rows sharded by {'estimated work (pairs x executed iterations, --balance work)' if args.balance == 'work' and world > 1 else 'pair count'} over {world} GPU(s), no data-path collective",
                       "balance": args.balance, "balance_estimate": balance_info,
                       "pairs_per_rank_min_max": [pairs_min, pairs_max],
                       "rank_seconds_min_max": [elapsed_min, elapsed_max],
                       "rank_records": rank_records,
                       "backend": ("gloo, every rank on GPU 0 (NGSLD_BENCH_ONE_DEVICE=1: a dry run of the N > 1 path, not a "
                                   "scaling measurement)" if one_dev else "nccl (RCCL)") if world > 1 else None,
                       "rccl_ranks_seen": ranks_seen,   # an all_reduce of ones over the job's backend (1 when there is no collective to run)
                       "pairs_replayed_exact_order_rank0_last_step": replayed[0],
                       "replay_rank0_last_step": replay_info,
                       "first_pass_s_rank0": round(t_first, 4) if first_in_warmup else None,
                       "replay_off": replay_off,
                       "gl_generate_s": round(t_gen, 3), "gl_broadcast_s": round(t_bc, 3),
                       "one_off_prep_ms_rank0": round(t_prep * 1e3, 2), "one_off_plan_ms_rank0": round(t_plan * 1e3, 2)},
            "value_host_resident": (total_pairs / sink_max) if sink_max > 0 else None,
            "host_resident_note": "SURVEY 8(d): pairs / wall time from the first kernel launch to the last record resident "
                                  "in host memory (ngsld_run: the pair kernels write the 72 B per pair straight into pinned host "
                                  f"buffers over the host link, two batches in turn); mean of {sink_passes} consecutive passes, all "
                                  "ranks, MAX over ranks",
            "e2e_file_to_tsv_s": e2e,
            "unfiltered_input": unfiltered,
            "other_configs": other_configs,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                         "traffic_per_pair": None, "traffic_detail": None,
                         "frac_is": "algorithmic bytes per pair x pairs per launch / kernel time, over the HBM peak -- NOT the "
                                    "HBM utilisation: the row vector is shared through LDS and the window is re-read from L2 / "
                                    "Infinity Cache (`traffic`); the binding roofline is fp64_valu",
                         "kernel": {"group": "pair_ld_group_kernel", "run": "pair_ld_run_kernel", "ab": "pair_ld_ab_kernel",
                                    "multi": "pair_ld_kernel (multi-wavefront)", "multi-ab": "pair_ld_abm_kernel (multi-wavefront, a/b form)",
                                    "stream": "pair_ld_stream_kernel" if os.environ.get("NGSLD_PAIR_KERNEL") == "stream"
                                    else "pair_ld_bres_kernel (streaming, candidate's vector resident)",
                                    "hard": "pair_ld_hard_kernel (genotype-combination counts)"}.get(family, family),
                         "kernel_ms_per_launch": launch_s * 1e3,
                         "algorithmic_bytes_per_pair": bytes_pair,
                         "fp64_valu": {"achieved": fp64_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": fp64_tflops / FP64_PEAK_TFLOPS,
                                       "note": "20 f64 VALU ops per individual per executed EM iteration "
                                               "(FMA counted as 2 flop); the binding roofline"}},
        }
        if raw_head is not None:
            def gpu_rows(rows):
                k = int(row_off[rows] - row_off[0])
                r2 = d_std.view(torch.float64).view(-1, 4)[:k, 3]
                it = ext_i32[:k, 9].to(torch.int64)
                fin = torch.isfinite(r2)
                return k, float(r2[fin].sum()), int(torch.clamp(it + 1, max=100).sum())
            out["cpu_baseline"] = cpu_baseline(raw_head, pos_dist[:raw_head.shape[0]].copy(), args.max_kb,
                                               args.cpu_seconds, gpu_rows)
        else:
            out["cpu_baseline"] = None
        traffic, traffic_src, traffic_detail = None, None, None
        launches_per_step = max(launches, 1) / max(args.steps, 1)
        under_profiler = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCTX")) for k in os.environ)
        if world == 1 and not args.no_traffic and not under_profiler:   # (never a profiler inside a profiler)
            eng.close()                                   # (everything the line needs from it has been taken)
            eng = None
            traffic_detail = measure_traffic(args)
            if traffic_detail and "bytes_per_step" in traffic_detail:
                traffic = traffic_detail["bytes_per_step"] / launches_per_step
                traffic_src = traffic_detail["source"]
        if traffic is None:
            try:
                with open(args.traffic_json) as fh:
                    tj = json.load(fh)
                if tj.get("workload") == f"{args.sites}x{n_ind}@{args.max_kb}kb":
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_src = ("from_profile: rocprofv3 PMC passes of an earlier run of this workload (" +
                                   os.path.relpath(args.traffic_json, REPO) + "), not measured in this run")
            except (OSError, ValueError):
                pass
        out["roofline"].update(traffic=traffic, traffic_source=traffic_src, traffic_detail=traffic_detail,
                               traffic_per_pair=(traffic / pairs_per_launch) if traffic else None)
        print(json.dumps(out), flush=True)

    if eng is not None:
        eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
