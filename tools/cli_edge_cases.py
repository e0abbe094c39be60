"""Dev tool (GPU box): hand-built EDGE cases through both programs -- the drop-in binary (ngsld_amd/bin/ngsLD) and the reference's
own program (oracle/_ref ref_main) -- same argv, same files.  Where the reference program writes a table, the binary must write
the same one (header equal, sorted bodies byte-identical); where the reference program ends in an error, so must the binary (exit
status non-zero, the same "[function] ERROR: ..." line).  The fuzz generator (tests/test_gpu_fuzz.py) starts at three sites and
sorted, well-formed files; this is what lies below and beside it: one / two sites, one / two individuals, nothing but monomorphic
sites, nothing but missing data, windows that hold no pair, one chromosome per site, thresholds that drop everything, files
shorter or longer than --n_sites says.
python tools/cli_edge_cases.py [--json out.json] [--binary PATH]   (--binary oracle/ngsld_oracle: the oracle's CLI instead, no GPU)"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
from ngsld_amd import capi, shard, synth  # noqa: E402
from oracle import orc  # noqa: E402
from util import run_ref_program  # noqa: E402


def same_tsv(got: str, want: str):
    gl, wl = got.splitlines(keepends=True), want.splitlines(keepends=True)
    if not gl and not wl:
        return None
    if len(gl) != len(wl):
        return f"{len(gl)} lines against the reference's {len(wl)}"
    if gl[0] != wl[0]:
        return f"first line {gl[0]!r} against {wl[0]!r}"
    a, b = sorted(gl[1:]), sorted(wl[1:])
    for x, y in zip(a, b):
        if x != y:
            return f"{sum(1 for p, q in zip(a, b) if p != q)} rows differ, first:\n  hip {x!r}\n  ref {y!r}"
    return None


def error_line(stderr: str) -> str:
    for ln in stderr.splitlines():
        if "ERROR" in ln:   # (the reference's main() is compiled under the name ref_main: its __FUNCTION__ says so)
            return ln.strip().replace("[ref_main]", "[main]")
    return ""


def cases():
    """(name, raw [sites][ind][3], (chrs, pos) or None, oracle keywords, extra flags, n_sites override or None)"""
    rng = np.random.default_rng(424242)
    gl = lambda s, i, seed, depth=4.0: synth.make_gl_numpy(s, i, seed, depth=depth)  # noqa: E731
    posn = lambda s, seed, gap=200, n_chr=1: synth.make_positions(s, seed, max_gap=gap, n_chr=n_chr)  # noqa: E731
    out = []
    out.append(("one site", gl(1, 5, 1), posn(1, 1), {}, [], None))
    out.append(("two sites, one individual", gl(2, 1, 2), posn(2, 2), {}, [], None))
    out.append(("two sites, two individuals", gl(2, 2, 3), posn(2, 3), {}, [], None))
    out.append(("three sites, one individual, no positions", gl(3, 1, 4), None, {}, [], None))
    mono = np.tile(np.array([1.0, 0.0, 0.0]), (6, 20, 1))
    out.append(("every site monomorphic", mono, posn(6, 5), {}, [], None))
    mono2 = np.tile(np.array([0.0, 0.0, 1.0]), (6, 20, 1))
    out.append(("every site fixed for the other allele", mono2, posn(6, 6), {}, [], None))
    het = np.tile(np.array([0.0, 1.0, 0.0]), (5, 12, 1))
    out.append(("every genotype a called heterozygote", het, posn(5, 7), {}, [], None))
    miss = np.full((6, 20, 3), 1.0 / 3.0)
    out.append(("nothing but missing data", miss, posn(6, 8), {}, [], None))
    out.append(("nothing but missing data, --ignore_miss_data", miss, posn(6, 8), dict(ignore_miss_data=True), [], None))
    half = gl(8, 30, 9)
    half[:, 15:] = 1.0 / 3.0
    out.append(("half the individuals without data, --ignore_miss_data", half, posn(8, 9), dict(ignore_miss_data=True), [], None))
    out.append(("window that holds no pair", gl(8, 10, 10), (["chr1"] * 8, np.arange(1, 9, dtype=np.int64) * 5000),
                dict(max_kb_dist=1), [], None))
    out.append(("one chromosome per site, all pairs", gl(6, 10, 11), posn(6, 11, n_chr=6), {}, [], None))
    out.append(("one chromosome per site, windowed", gl(6, 10, 11), posn(6, 11, n_chr=6), dict(max_kb_dist=10), [], None))
    out.append(("--min_maf 0.5", gl(10, 40, 12), posn(10, 12), dict(min_maf=0.5), [], None))
    out.append(("--min_maf 0.499", gl(10, 40, 12), posn(10, 12), dict(min_maf=0.499), [], None))
    out.append(("--max_snp_dist 1", gl(10, 40, 13), posn(10, 13), dict(max_snp_dist=1), [], None))
    out.append(("--max_snp_dist 1 and --max_kb_dist 1", gl(10, 40, 13), posn(10, 13, gap=900), dict(max_snp_dist=1, max_kb_dist=1), [], None))
    zeros = gl(5, 9, 14)
    zeros[2, 3] = 0.0
    zeros[4] = 0.0
    out.append(("all-zero triples (one individual; one whole site)", zeros, posn(5, 14), {}, [], None))
    big = gl(5, 9, 15)
    big[1] *= 1e300
    big[3] *= 1e-300
    out.append(("likelihoods near the ends of the double range", big, posn(5, 15), {}, [], None))
    with np.errstate(divide="ignore"):
        hard_log = np.log(np.eye(3)[rng.integers(0, 3, size=(6, 11))])
    out.append(("log scale with -inf entries", hard_log, posn(6, 16), dict(log_scale=True), [], None))
    out.append(("file longer than --n_sites", gl(10, 7, 17), posn(6, 17), {}, [], 6))
    out.append(("file shorter than --n_sites", gl(4, 7, 18), posn(6, 18), {}, [], 6))
    out.append(("positions file shorter than --n_sites", gl(6, 7, 19), posn(5, 19), {}, [], None))
    out.append(("positions file longer than --n_sites", gl(6, 7, 20), posn(7, 20), {}, [], None))
    chrs, pos = posn(6, 21)
    pos = pos.copy()
    pos[3] = pos[2]
    out.append(("two sites at one position", gl(6, 7, 21), (chrs, pos), {}, [], None))
    pos2 = posn(6, 22)[1].copy()
    pos2[4] = pos2[3] - 1
    out.append(("positions going backwards", gl(6, 7, 22), (["chr1"] * 6, pos2), {}, [], None))
    out.append(("--call_geno with both thresholds 0", gl(8, 25, 23), posn(8, 23), dict(call_geno=(0.0, 0.0)),
                ["--probs", "--call_geno", "--N_thresh", "0.0", "--call_thresh", "0.0"], None))
    out.append(("--call_geno with both thresholds 1", gl(8, 25, 24), posn(8, 24), dict(call_geno=(1.0, 1.0)),
                ["--probs", "--call_geno", "--N_thresh", "1.0", "--call_thresh", "1.0"], None))
    return out


def text_cases():
    """(name, file text, gz?, n_ind, n_sites flag, --probs?, extra flags): text genotype input as read_data.cpp:50-104 takes it apart
    -- header by field count, blank lines, the LAST n_ind (x 3) columns, (int) truncation of a genotype, EOF checks."""
    rng = np.random.default_rng(515151)
    g = rng.integers(0, 3, size=(7, 9)).astype(float)
    g[rng.random(g.shape) < 0.15] = -1.0
    rows = ["\t".join(str(int(x)) for x in r) for r in g]
    body = "\n".join(rows) + "\n"
    out = []
    out.append(("text: called genotypes, plain file", body, False, 9, 7, False, []))
    out.append(("text: gzipped", body, True, 9, 7, False, []))
    out.append(("text: header line with fewer fields", "marker\tref\talt\n" + body, True, 9, 7, False, []))
    out.append(("text: blank lines between and after the rows", "\n".join(rows[:3]) + "\n\n" + "\n".join(rows[3:]) + "\n\n\n", True, 9, 7, False, []))
    out.append(("text: CRLF line ends", "\r\n".join(rows) + "\r\n", True, 9, 7, False, []))
    out.append(("text: no newline after the last row", "\n".join(rows), True, 9, 7, False, []))
    out.append(("text: leading label columns", "".join(f"chr1_{k}\tA\tC\t{r}\n" for k, r in enumerate(rows)), True, 9, 7, False, []))
    out.append(("text: space-separated", body.replace("\t", " "), True, 9, 7, False, []))
    out.append(("text: a genotype of 3", body.replace("2", "3", 1), True, 9, 7, False, []))
    out.append(("text: genotypes with fractions (1.9, 0.4, -0.5, 2.7)", "\n".join("\t".join(["1.9", "0.4", "-0.5", "2.7", "1", "0", "2", "-1", "1"]) for _ in range(7)) + "\n",
                True, 9, 7, False, []))
    out.append(("text: a word among the genotypes", body.replace("1", "NA", 1), True, 9, 7, False, []))
    out.append(("text: a later row with fewer fields", "\n".join(rows[:4]) + "\n" + "\t".join(rows[4].split("\t")[:5]) + "\n" + "\n".join(rows[5:]) + "\n",
                True, 9, 7, False, []))
    out.append(("text: the header line repeated among the rows", "\n".join(rows[:3]) + "\nmarker\tref\talt\n" + "\n".join(rows[3:]) + "\n", True, 9, 7, False, []))
    out.append(("text: two header lines at the top, the second with a few numbers", "marker\tref\talt\n1\t2\tx\n" + body, True, 9, 7, False, []))
    out.append(("text: a genotype of 3 in an early row AND a row too many", body.replace("2", "3", 1) + rows[0] + "\n", True, 9, 7, False, []))
    out.append(("text: an empty line in a row's place (as many lines as sites)", "\n".join(rows[:3]) + "\n\n" + "\n".join(rows[4:]) + "\n", True, 9, 7, False, []))
    out.append(("text: more rows than --n_sites", body, True, 9, 5, False, []))
    out.append(("text: fewer rows than --n_sites", body, True, 9, 9, False, []))
    out.append(("text: empty file", "", True, 9, 7, False, []))
    out.append(("text: every genotype missing", "\n".join("\t".join(["-1"] * 9) for _ in range(7)) + "\n", True, 9, 7, False, []))
    out.append(("text: one individual", "\n".join(str(int(x)) for x in g[:, 0]) + "\n", True, 1, 7, False, []))
    raw = synth.make_gl_numpy(7, 6, 99, depth=3.0)
    raw /= raw.sum(axis=2, keepdims=True)
    raw[2, 1] = 0.0
    raw[5, 4] = np.array([1.0, 0.0, 0.0])
    probs = "\n".join("\t".join(repr(float(x)) for x in r.reshape(-1)) for r in raw) + "\n"
    out.append(("text: likelihood triples with an all-zero one", probs, True, 6, 7, True, []))
    out.append(("text: likelihood triples read as called genotypes (no --probs)", probs, True, 18, 7, False, []))
    out.append(("text: likelihood triples, --call_geno", probs, True, 6, 7, True, ["--call_geno", "--N_thresh", "0.3", "--call_thresh", "0.8"]))
    return out


def main():
    results, bad = [], 0
    binary = sys.argv[sys.argv.index("--binary") + 1] if "--binary" in sys.argv else capi.CLI_PATH
    for extend in (False, True):
        for name, raw, pp, kw, extra, n_sites_flag in cases():
            raw = np.ascontiguousarray(raw, dtype=np.float64)
            n_file, n_ind = raw.shape[:2]
            n_sites = n_sites_flag if n_sites_flag is not None else n_file
            with tempfile.TemporaryDirectory() as d:
                g = os.path.join(d, "in.glf")
                raw.tofile(g)
                flags = ["--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--verbose", "0"]
                pd = None
                if pp is not None:
                    p = os.path.join(d, "in.pos")
                    synth.write_pos(p, list(pp[0]), pp[1])
                    flags += ["--pos", p]
                    if len(pp[1]) == n_sites:
                        pd = shard.pos_dist_from_positions(list(pp[0]), pp[1])
                okw = dict(kw)
                flags += ["--max_kb_dist", str(okw.get("max_kb_dist", 0)), "--max_snp_dist", str(okw.get("max_snp_dist", 0)),
                          "--min_maf", repr(float(okw.get("min_maf", 0.0)))]
                if okw.get("log_scale"):
                    flags.append("--log_scale")
                if okw.get("ignore_miss_data"):
                    flags.append("--ignore_miss_data")
                flags += extra
                if extend:
                    flags.append("--extend_out")
                # the oracle's records: only for the reference program's r2_ExpG column (GSL is not in the image)
                rec = np.zeros(0, dtype=orc.PAIR_DTYPE)
                usable = n_file >= n_sites and (pp is None or pd is not None) and (pd is None or not np.any(pd[np.isfinite(pd)] < 1))
                if usable:
                    try:
                        with np.errstate(all="ignore"):
                            rec = orc.Oracle(raw[:n_sites], pd, n_threads=2, **okw).run()
                    except (ValueError, RuntimeError):
                        rec = np.zeros(0, dtype=orc.PAIR_DTYPE)
                out_ref, out_hip = os.path.join(d, "ref.tsv"), os.path.join(d, "hip.tsv")
                r = run_ref_program(rec, n_sites, flags, out_ref, d, threads=2)
                h = subprocess.run([binary, *flags, "--n_threads", "2", "--out", out_hip], capture_output=True, text=True,
                                   timeout=600)
                verdict = None
                if r.returncode == 0:
                    if h.returncode != 0:
                        verdict = f"reference program wrote a table, the binary ended with {h.returncode}: {h.stderr[-300:]}"
                    else:
                        verdict = same_tsv(open(out_hip).read(), open(out_ref).read())
                    what = f"table of {max(0, len(open(out_ref).read().splitlines()) - 1)} rows"
                else:
                    if h.returncode == 0:
                        verdict = f"reference program ended with {r.returncode} ({error_line(r.stderr)}), the binary wrote a table"
                    elif error_line(r.stderr) != error_line(h.stderr):
                        verdict = f"error lines differ: ref {error_line(r.stderr)!r} hip {error_line(h.stderr)!r}"
                    what = f"error: {error_line(r.stderr)}"
                tag = name + (", --extend_out" if extend else "")
                results.append({"case": tag, "reference": what, "same": verdict is None, "difference": verdict})
                bad += verdict is not None
                print(f"{'same     ' if verdict is None else 'DIFFERENT'}  {tag}: {what}" + ("" if verdict is None else f"\n           {verdict}"),
                      flush=True)
    import ctypes as C
    import gzip
    for name, text, gz, n_ind, n_sites, probs, extra in text_cases():
        with tempfile.TemporaryDirectory() as d:
            g = os.path.join(d, "in.geno" + (".gz" if gz else ""))
            with (gzip.open(g, "wt", newline="") if gz else open(g, "w", newline="")) as fh:
                fh.write(text)
            flags = ["--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--verbose", "0", "--extend_out",
                     "--max_kb_dist", "0", "--max_snp_dist", "0"]
            if probs:
                flags.append("--probs")
            flags += extra
            rec = np.zeros(0, dtype=orc.PAIR_DTYPE)
            gl = np.empty((n_sites, n_ind, 3))
            err = C.create_string_buffer(256)
            if orc.lib().orc_read_geno_text(g.encode(), int(probs), 0, n_ind, n_sites, orc.dp(gl), err, 256) == 0:
                call = (0.3, 0.8) if "--call_geno" in extra else None
                with np.errstate(all="ignore"):
                    rec = orc.Oracle(gl, None, already_normalised_log=True, n_threads=2, call_geno=call).run()
            out_ref, out_hip = os.path.join(d, "ref.tsv"), os.path.join(d, "hip.tsv")
            r = run_ref_program(rec, n_sites, flags, out_ref, d, threads=2)
            h = subprocess.run([binary, *flags, "--n_threads", "2", "--out", out_hip], capture_output=True, text=True, timeout=600)
            verdict = None
            if r.returncode == 0:
                if h.returncode != 0 and "empty line" in name and "empty line in GENO file" in error_line(h.stderr):
                    pass  # the reference goes on with the site an empty line left uninitialised (read_data.cpp:58-59)
                elif h.returncode != 0:
                    verdict = f"reference program wrote a table, the binary ended with {h.returncode}: {h.stderr[-300:]}"
                else:
                    verdict = same_tsv(open(out_hip).read(), open(out_ref).read())
                what = f"table of {max(0, len(open(out_ref).read().splitlines()) - 1)} rows"
            else:
                if h.returncode == 0:
                    verdict = f"reference program ended with {r.returncode} ({error_line(r.stderr)}), the binary wrote a table"
                elif error_line(r.stderr) != error_line(h.stderr) and not (
                        "empty line" in name and "empty line in GENO file" in error_line(h.stderr)):
                    # (an empty line that takes a site's place is an error here -- the reference leaves that
                    # site uninitialised, read_data.cpp:58-59, and goes on)
                    verdict = f"error lines differ: ref {error_line(r.stderr)!r} hip {error_line(h.stderr)!r}"
                what = f"error: {error_line(r.stderr)}"
            results.append({"case": name, "reference": what, "same": verdict is None, "difference": verdict})
            bad += verdict is not None
            print(f"{'same     ' if verdict is None else 'DIFFERENT'}  {name}: {what}" + ("" if verdict is None else f"\n           {verdict}"), flush=True)
    # no --out: the table goes to standard output (parse_args.cpp:26), with every thread count the same rows
    import util
    for threads in (1, 3):
        with tempfile.TemporaryDirectory() as d:
            raw = synth.make_gl_numpy(9, 30, 77, depth=3.0)
            chrs, pos = synth.make_positions(9, 77, max_gap=300, n_chr=2)
            g, p = os.path.join(d, "in.glf"), os.path.join(d, "in.pos")
            raw.tofile(g)
            synth.write_pos(p, chrs, pos)
            flags = ["--geno", g, "--n_ind", "30", "--n_sites", "9", "--pos", p, "--verbose", "0", "--max_kb_dist", "0", "--extend_out",
                     "--n_threads", str(threads)]
            rec = orc.Oracle(raw, shard.pos_dist_from_positions(chrs, pos), n_threads=2).run()
            first = np.zeros(10, dtype=np.uint64)
            np.add.at(first, rec["s1"].astype(np.int64) + 1, 1)
            tab = os.path.join(d, "r2_table.npz")
            np.savez(tab, first=np.cumsum(first).astype(np.uint64), s2=rec["s2"].astype(np.uint64), val=rec["r2pear"].astype(np.float64))
            r = subprocess.run([sys.executable, "-c", util._REF_CHILD, tab, *flags], capture_output=True, text=True, timeout=600)
            h = subprocess.run([binary, *flags], capture_output=True, text=True, timeout=600)
            verdict = None
            if r.returncode != 0 or h.returncode != 0:
                verdict = f"exit status {h.returncode} against the reference program's {r.returncode}: {h.stderr[-300:]}"
            else:
                verdict = same_tsv(h.stdout, r.stdout)
            name = f"no --out: the table on standard output, --n_threads {threads}"
            what = f"table of {max(0, len(r.stdout.splitlines()) - 1)} rows"
            results.append({"case": name, "reference": what, "same": verdict is None, "difference": verdict})
            bad += verdict is not None
            print(f"{'same     ' if verdict is None else 'DIFFERENT'}  {name}: {what}" + ("" if verdict is None else f"\n           {verdict}"), flush=True)
    # files: compressed inputs, and the ones that are not there or cannot be written
    import gzip as _gz
    for name, mode in (("binary genotype file gzipped", "geno_gz"), ("positions file gzipped", "pos_gz"), ("genotype file that does not exist", "no_geno"),
                       ("positions file that does not exist", "no_pos"), ("positions file with a row of three fields among rows of two", "pos_fields"),
                       ("positions file with comment lines and blank lines", "pos_comments"), ("positions file with CRLF line ends", "pos_crlf"),
                       ("positions file without a usable line", "pos_empty"), ("positions file with one line too many AND a row of three fields", "pos_fields_and_count"),
                       ("positions written with leading zeros (strtoul reads them as octal, read_data.cpp:211)", "pos_octal"),
                       ("positions in scientific notation (strtod and strtoul disagree, read_data.cpp:204,211)", "pos_sci"),
                       ("output path that cannot be written", "bad_out")):
        with tempfile.TemporaryDirectory() as d:
            raw = synth.make_gl_numpy(8, 12, 78, depth=3.0)
            chrs, pos = synth.make_positions(8, 78, max_gap=300)
            g, p = os.path.join(d, "in.glf"), os.path.join(d, "in.pos")
            raw.tofile(g)
            synth.write_pos(p, chrs, pos)
            if mode == "geno_gz":
                with open(g, "rb") as fi, _gz.open(g + ".gz", "wb") as fo:
                    fo.write(fi.read())
                g += ".gz"
            if mode == "pos_gz":
                with open(p, "rb") as fi, _gz.open(p + ".gz", "wb") as fo:
                    fo.write(fi.read())
                p += ".gz"
            if mode == "pos_fields":
                lines = open(p).read().splitlines()
                lines[4] += "\textra"
                open(p, "w").write("\n".join(lines) + "\n")
            if mode == "pos_comments":
                lines = open(p).read().splitlines()
                open(p, "w").write("# positions\n" + "\n".join(lines[:3]) + "\n\n#another\n" + "\n".join(lines[3:]) + "\n\n")
            if mode == "pos_crlf":
                crlf = open(p).read().replace("\n", "\r\n")
                open(p, "w", newline="").write(crlf)
            if mode == "pos_empty":
                open(p, "w").write("# nothing but a comment\n\n")
            if mode == "pos_fields_and_count":
                lines = open(p).read().splitlines()
                lines[6] += "\textra"
                open(p, "w").write("\n".join(lines + ["chr1\t999999"]) + "\n")
            if mode == "pos_octal":
                open(p, "w").write("".join(f"chr1\t0{10 * (k + 1)}\n" for k in range(8)))
            if mode == "pos_sci":
                open(p, "w").write("".join(f"chr1\t{k + 1}e2\n" for k in range(8)))
            if mode == "no_geno":
                g = os.path.join(d, "nothing.glf")
            if mode == "no_pos":
                p = os.path.join(d, "nothing.pos")
            flags = ["--geno", g, "--n_ind", "12", "--n_sites", "8", "--pos", p, "--verbose", "0", "--max_kb_dist", "0", "--extend_out"]
            rec = orc.Oracle(raw, shard.pos_dist_from_positions(chrs, pos), n_threads=2).run()
            out_ref, out_hip = os.path.join(d, "ref.tsv"), os.path.join(d, "hip.tsv")
            if mode == "bad_out":
                out_ref = out_hip = os.path.join(d, "no_such_directory", "out.tsv")
            r = run_ref_program(rec, 8, flags, out_ref, d, threads=2)
            h = subprocess.run([binary, *flags, "--n_threads", "2", "--out", out_hip], capture_output=True, text=True, timeout=600)
            verdict = None
            if r.returncode == 0:
                if h.returncode != 0:
                    verdict = f"reference program wrote a table, the binary ended with {h.returncode}: {h.stderr[-300:]}"
                else:
                    verdict = same_tsv(open(out_hip).read(), open(out_ref).read())
                what = f"table of {max(0, len(open(out_ref).read().splitlines()) - 1)} rows"
            else:
                if h.returncode == 0:
                    verdict = f"reference program ended with {r.returncode} ({error_line(r.stderr)}), the binary wrote a table"
                elif error_line(r.stderr) != error_line(h.stderr):
                    verdict = f"error lines differ: ref {error_line(r.stderr)!r} hip {error_line(h.stderr)!r}"
                what = f"error: {error_line(r.stderr)}"
            results.append({"case": name, "reference": what, "same": verdict is None, "difference": verdict})
            bad += verdict is not None
            print(f"{'same     ' if verdict is None else 'DIFFERENT'}  {name}: {what}" + ("" if verdict is None else f"\n           {verdict}"), flush=True)
    print(f"edge cases: {len(results)} through both programs, {bad} differ")
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump(results, fh, indent=1)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
