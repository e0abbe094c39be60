"""The drop-in binary against the reference's own program at the scale of a BASELINE configuration (GPU box).

    python tools/cli_vs_ref_config.py c1            # configs[1]: 5,000 x 100, all 12,497,500 pairs, extended columns
    python tools/cli_vs_ref_config.py c2 [sites]    # the first `sites` (default 12,000) sites of configs[2]'s matrix: x 500, 100 kb window
    python tools/cli_vs_ref_config.py c2call [sites]  # the same with --call_geno --N_thresh 0.4 --call_thresh 0.8
    python tools/cli_vs_ref_config.py c2mono [sites]  # the same shape NOT SNP-called: 20 % of the sites monomorphic (a third of the rows are
                                                      # replayed in the reference's operation order, on the device: ld_replay_lkl.hip)
    python tools/cli_vs_ref_config.py c2sfs [sites]   # ... site frequencies log-uniform in [0.001, 0.5]
    python tools/cli_vs_ref_config.py c3 [sites]    # configs[3]'s shape: sites (default 3,000) x 1,000, all pairs
    python tools/cli_vs_ref_config.py c4 [sites]    # configs[4]'s shape: sites (default 5,000) x 2,000, 500 kb window

Both programs get the same argv over the same files (binary likelihoods + positions).  `ref_main` is ngsLD.cpp's main() +
calc_pair_LD compiled as they stand minus the GSL statements (oracle/build_ref.sh), on every host core; its r2_ExpG column is
looked up in the oracle's records of the same run (GSL is not in the image).  The two tables are compared as files: first line
equal, bodies equal after `LC_ALL=C sort` (the reference's rows come out in thread order) -- line counts and md5 of both.
A third program runs when it is built: the reference's own main() with its thread-pool section replaced by
integration/ngsld_binding.h (oracle/_ref/libngsld_ref_hip.so) -- the reference's reader, est_maf and exp() loop on one host
thread, the pairs on the device.  Prints one JSON line; exit status 1 when any table differs."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import torch  # noqa: E402,F401

from ngsld_amd import capi, shard, synth  # noqa: E402
from oracle import orc  # noqa: E402
from util import have_patched_ref_program, run_patched_ref_program, run_ref_program  # noqa: E402


def sorted_body_md5(path: str, d: str) -> tuple[str, int, str]:
    """(first line, number of body lines, md5 of the sorted body) without holding the table in memory."""
    with open(path, "rb") as fh:
        head = fh.readline()
    body = os.path.join(d, os.path.basename(path) + ".sorted")
    env = dict(os.environ, LC_ALL="C")
    with open(body, "wb") as out:
        tail = subprocess.Popen(["tail", "-n", "+2", path], stdout=subprocess.PIPE)
        subprocess.run(["sort", "-S", "4G", "-T", d], stdin=tail.stdout, stdout=out, env=env, check=True)
        tail.wait()
    h, n = hashlib.md5(), 0
    with open(body, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 24), b""):
            h.update(chunk)
            n += chunk.count(b"\n")
    os.remove(body)
    return head.decode(), n, h.hexdigest()


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c1"
    dev = torch.device("cuda", 0)
    call = None
    if which == "c1":
        n_sites, n_ind, max_kb, seed = 5000, 100, 0, 2
        raw = synth.make_gl_torch(n_sites, n_ind, seed, dev, depth=10.0).cpu().numpy()
        chrs, pos = synth.make_positions(n_sites, seed)
    elif which == "c3":      # configs[3]'s shape (x 1,000, all pairs: two wavefronts per pair), `sites` of it
        n_sites, n_ind, max_kb, seed = (int(sys.argv[2]) if len(sys.argv) > 2 else 3000), 1000, 0, 4
        raw = synth.make_gl_torch(n_sites, n_ind, seed, dev, depth=10.0).cpu().numpy()
        chrs, pos = synth.make_positions(n_sites, seed)
    elif which == "c4":      # configs[4]'s shape (x 2,000, 500 kb window over ~1 kb gaps: four wavefronts per pair)
        n_sites, n_ind, max_kb, seed = (int(sys.argv[2]) if len(sys.argv) > 2 else 5000), 2000, 500, 5
        raw = synth.make_gl_torch(n_sites, n_ind, seed, dev, depth=10.0).cpu().numpy()
        chrs, pos = synth.make_positions(n_sites, seed, max_gap=2000)
    else:                    # c2, or c2call: the same matrix with --call_geno (the genotype-combination kernel, replay on the device)
        n_all, n_ind, max_kb, seed = 100_000, 500, 100, 3
        n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 12_000
        raw = synth.make_gl_torch(n_sites if which in ("c2mono", "c2sfs") else n_all, n_ind, seed, dev, depth=10.0,
                                  mono_frac=0.2 if which == "c2mono" else 0.0, sfs=which == "c2sfs")[:n_sites].cpu().numpy()
        chrs, pos = synth.make_positions(n_all, seed)
        chrs, pos = chrs[:n_sites], pos[:n_sites]
        if which == "c2call":
            call = (0.4, 0.8)
    pd = shard.pos_dist_from_positions(chrs, pos)
    cores = len(os.sched_getaffinity(0))
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            cores = max(1, min(cores, int(float(q[0]) / float(q[1]) + 0.5)))
    except (OSError, ValueError, IndexError):
        pass
    with tempfile.TemporaryDirectory(dir=os.environ.get("NGSLD_TMP", None)) as d:
        g, p = os.path.join(d, "in.glf"), os.path.join(d, "in.pos")
        raw.tofile(g)
        synth.write_pos(p, chrs, pos)
        flags = ["--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p, "--max_kb_dist", str(max_kb),
                 "--min_maf", "0", "--extend_out", "--verbose", "0"]
        if call is not None:
            flags += ["--probs", "--call_geno", "--N_thresh", repr(call[0]), "--call_thresh", repr(call[1])]
        t0 = time.perf_counter()
        rec = orc.Oracle(raw, pd, max_kb_dist=max_kb, n_threads=cores, call_geno=call).run()
        t_orc = time.perf_counter() - t0
        out_ref = os.path.join(d, "ref.tsv")
        t0 = time.perf_counter()
        r = run_ref_program(rec, n_sites, flags, out_ref, d, threads=cores, timeout=3000)
        t_ref = time.perf_counter() - t0
        if r.returncode != 0:
            print(r.stderr[-2000:], file=sys.stderr)
            sys.exit(2)
        del rec
        out_hip = os.path.join(d, "hip.tsv")
        t0 = time.perf_counter()
        h = subprocess.run([capi.CLI_PATH, *flags, "--n_threads", str(cores), "--out", out_hip], capture_output=True, text=True)
        t_hip = time.perf_counter() - t0
        if h.returncode != 0:
            print(h.stderr[-2000:], file=sys.stderr)
            sys.exit(2)
        size = os.path.getsize(out_ref)
        head_r, n_r, md5_r = sorted_body_md5(out_ref, d)
        head_h, n_h, md5_h = sorted_body_md5(out_hip, d)
        # third program: the reference's own main() with its thread-pool section replaced by integration/ngsld_binding.h
        patched = None
        if have_patched_ref_program():
            out_p = os.path.join(d, "patched.tsv")
            t0 = time.perf_counter()
            pp = run_patched_ref_program(flags, out_p, threads=cores, timeout=3000)
            t_p = time.perf_counter() - t0
            if pp.returncode != 0:
                print(pp.stderr[-2000:], file=sys.stderr)
                sys.exit(2)
            head_p, n_p, md5_p = sorted_body_md5(out_p, d)
            patched = {"seconds": round(t_p, 2), "rows": n_p, "md5_sorted_body": md5_p,
                       "identical": head_p == head_r and n_p == n_r and md5_p == md5_r}
    same = head_r == head_h and n_r == n_h and md5_r == md5_h and (patched is None or patched["identical"])
    print(json.dumps({"config": which, "n_sites": n_sites, "n_ind": n_ind, "max_kb_dist": max_kb, "rows": n_r, "rows_hip": n_h,
                      "tsv_bytes": size, "first_line_equal": head_r == head_h, "md5_sorted_body_ref": md5_r,
                      "md5_sorted_body_hip": md5_h, "identical": same, "host_threads": cores,
                      "seconds_reference_program": round(t_ref, 2), "seconds_hip_binary": round(t_hip, 2),
                      "seconds_oracle_for_r2_ExpG_table": round(t_orc, 2),
                      "reference_main_with_the_binding": patched}))
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
