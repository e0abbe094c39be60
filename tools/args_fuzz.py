"""Dev tool (CPU): random command lines through the reference's own parser (parse_args.cpp compiled whole, oracle/_ref) and through
the drop-in binary -- the comparison of tests/test_cli_args_vs_ref.py over generated argv: options in any order, single or double
dash, abbreviated to any prefix (unique or not), repeated, with values that atoi / atof take apart in their own way ("5x", "7.9",
"1e2", "-3", "", "abc", "nan"), flags that imply others, missing arguments.  Valid arguments: the reference's echo block must
head the binary's stderr line for line.  Invalid ones: both exit with -1 and print the same text up to the end of the ERROR block.
python tools/args_fuzz.py [first] [last]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
from ngsld_amd import capi  # noqa: E402
from test_cli_args_vs_ref import _CHILD, _norm  # noqa: E402

INTS = ["5", "7", "0", "1", "-3", "5x", "7.9", "abc", "", "1e2", "99999999999", "+4", " 6", "0x10", "010"]
REALS = ["0.5", "1", "0", "1.5", "-0.1", "abc", "1e-2", "nan", "inf", ".3", "0.9x", "", "1.0000001", "1e0"]
NAMES = ["in.glf", "x", "", "a b", "-", "--geno"]
OPTS = [("geno", NAMES), ("probs", None), ("log_scale", None), ("n_ind", INTS), ("n_sites", INTS), ("pos", NAMES), ("posH", NAMES),
        ("max_kb_dist", INTS), ("max_snp_dist", INTS), ("min_maf", REALS), ("ignore_miss_data", None), ("call_geno", None),
        ("N_thresh", REALS), ("call_thresh", REALS), ("rnd_sample", REALS), ("seed", INTS), ("extend_out", None), ("out", NAMES),
        ("n_threads", INTS), ("verbose", ["0", "1", "2", "5", "x", "-1"])]


def make_argv(k):
    rng = np.random.default_rng(930_000 + k)
    argv = []
    picked = [o for o in OPTS if rng.random() < (0.85 if o[0] in ("geno", "n_ind", "n_sites", "pos", "seed") else 0.25)]
    if rng.random() < 0.2:
        picked += [OPTS[int(rng.integers(0, len(OPTS)))]]           # a repeated option
    order = rng.permutation(len(picked))
    for i in order:
        name, pool = picked[int(i)]
        if rng.random() < 0.15:
            name = name[:int(rng.integers(1, len(name) + 1))]       # a prefix: getopt_long accepts a unique one
        dash = "--" if rng.random() < 0.8 else "-"
        if pool is None:
            argv.append(dash + name)
        elif rng.random() < 0.1:
            argv.append(dash + name + "=" + str(rng.choice(pool)))  # --name=value
        else:
            argv.append(dash + name)
            if rng.random() < 0.97:
                argv.append(str(rng.choice(pool)))
    if rng.random() < 0.05:
        argv.append("--" + str(rng.choice(["nothing", "outH", "help", "version"])))
    if rng.random() < 0.05:
        argv.append("stray")
    # keep the reference's echo deterministic: without --seed it draws one from the clock
    if not any(a.lstrip("-").startswith("se") for a in argv):
        argv += ["--seed", "3"]
    return argv


def compare(argv, d):
    want = subprocess.run([sys.executable, "-c", _CHILD, *argv], capture_output=True, text=True, cwd=d, timeout=120)
    got = subprocess.run([capi.CLI_PATH, *argv], capture_output=True, text=True, cwd=d, timeout=120)
    w, g = _norm(want.stderr), _norm(got.stderr)
    # the binary's own options (--device, --devices, --max_gpu_mem) lengthen getopt's list of possibilities for a prefix that is
    # ambiguous in the reference already; they make no prefix ambiguous that the reference accepts
    g = [ln.replace(" '--max_gpu_mem'", "").replace(" '-max_gpu_mem'", "") if "is ambiguous" in ln else ln for ln in g]
    if want.returncode == 0 and "PARSED" in want.stdout:
        if g[:len(w)] != w:
            return f"echo differs:\n  hip {g[:len(w)]}\n  ref {w}"
        if "parse_cmd_args" in "\n".join(g[len(w):]):
            return "the binary raises an argument error where the reference's parser returns"
        return None
    if got.returncode != want.returncode:
        return f"exit status {got.returncode} against the reference's {want.returncode}\n  hip {g[-6:]}\n  ref {w[-6:]}"
    cut = lambda lines: lines[:max((i for i, ln in enumerate(lines) if ln.startswith("=====")), default=len(lines) - 1) + 1]  # noqa: E731
    if cut(g) != cut(w):
        return f"stderr differs:\n  hip {cut(g)}\n  ref {cut(w)}"
    return None


def main():
    first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 300)
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        for k in range(first, last):
            argv = make_argv(k)
            diff = compare(argv, d)
            if diff:
                bad += 1
                print(f"case {k}: {argv}\n  {diff}", flush=True)
    print(f"args fuzz: cases {first}..{last - 1}, {bad} differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
