"""GPU: the drop-in binary against the reference's OWN program, same argv, same files, same bytes.

`ngsld_amd/bin/ngsLD` (HIP) and `ref_main` (oracle/_ref: ngsLD.cpp's main() + calc_pair_LD compiled as they stand but for the GSL
statements, oracle/build_ref.sh) are handed the SAME command line over the SAME input files; the TSV of one must be the TSV of
the other -- header equal, body equal as sorted lines (the reference's rows come out in thread order, examples/test.sh:16 sorts
them too).  Fixtures first, then the fuzz generator's cases written out as files (binary likelihoods, chromosome breaks, every
filter, --ignore_miss_data, --log_scale, --call_geno, allele-frequency thresholds that sit ON a site's frequency).  The
reference program's r2_ExpG column is the oracle's (GSL is not in the image): the one column this does not pin; --rnd_sample
needs gsl_rng and is left to the fixtures of test_gpu_golden.py.  (tools/cli_soak.py runs the same comparison over any range.)"""
import os
import subprocess

import numpy as np
import pytest

from ngsld_amd import capi, synth
from oracle import orc
from test_gpu_fuzz import _case_full, pick_min_maf
from util import Fixture, fixtures, have_ref_program, run_ref_program

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not have_ref_program(), reason="oracle/_ref predates ref_main (rebuild with oracle/build_ref.sh)")]


def same_tsv(got: str, want: str) -> str | None:
    """None when the two programs wrote the same table; else where they differ."""
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


def both_programs(flags: list[str], rec, n_sites: int, d: str, threads: int = 2, hip_flags=(), hip_env=None):
    """hip_flags / hip_env: options only the drop-in binary knows (--devices, NGSLD_TEST_SLAB_SITES ...), for it alone."""
    out_ref = os.path.join(d, "ref.tsv")
    r = run_ref_program(rec, n_sites, flags, out_ref, d, threads)
    assert r.returncode == 0, r.stderr[-2000:]
    out_hip = os.path.join(d, "hip.tsv")
    env = dict(os.environ)
    env.update(hip_env or {})
    h = subprocess.run([capi.CLI_PATH, *flags, *hip_flags, "--n_threads", str(threads), "--out", out_hip], capture_output=True,
                       text=True, timeout=600, env=env)
    assert h.returncode == 0, h.stderr[-2000:]
    return open(out_hip).read(), open(out_ref).read()


@pytest.mark.parametrize("name", [n for n in fixtures() if Fixture(n).rnd_sample >= 1 and Fixture(n).n_ind <= 500])
@pytest.mark.parametrize("extend", [False, True])
def test_fixture_through_both_programs(name, extend, tmp_path):
    fx = Fixture(name)
    d = str(tmp_path)
    g, p = fx.write_inputs(d)
    flags = ["--geno", g, "--n_ind", str(fx.n_ind), "--n_sites", str(fx.n_sites), "--verbose", "0"]
    if p:
        flags += ["--posH" if fx.header else "--pos", p]
    flags += fx.cli_flags(extend)
    got, want = both_programs(flags, fx.oracle().run(), fx.n_sites, d)
    assert same_tsv(got, want) is None, same_tsv(got, want)


def case_files(k: int, d: str):
    """Fuzz case k as the files and flags of a command line (None: a case the reference program cannot run here)."""
    raw, pd, kw, call, chrs, pos = _case_full(k)
    n_sites, n_ind = raw.shape[:2]
    o0 = orc.Oracle(raw, pd, log_scale=kw["log_scale"], call_geno=call)
    min_maf = pick_min_maf(o0.maf, k)
    o = orc.Oracle(raw, pd, min_maf=min_maf, n_threads=4, call_geno=call, log_scale=kw["log_scale"],
                   ignore_miss_data=kw["ignore_miss_data"], max_kb_dist=kw["max_kb_dist"], max_snp_dist=kw["max_snp_dist"])
    g, p = os.path.join(d, "in.glf"), os.path.join(d, "in.pos")
    raw.tofile(g)
    header = k % 4 == 1
    synth.write_pos(p, chrs, pos, header=header, extra_col=k % 5 == 2)
    flags = ["--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--verbose", "0", "--posH" if header else "--pos", p,
             "--max_kb_dist", str(kw["max_kb_dist"]), "--max_snp_dist", str(kw["max_snp_dist"]), "--min_maf", repr(min_maf)]
    if kw["log_scale"]:
        flags.append("--log_scale")
    if kw["ignore_miss_data"]:
        flags.append("--ignore_miss_data")
    if call is not None:
        flags += ["--probs", "--call_geno", "--N_thresh", repr(float(call[0])), "--call_thresh", repr(float(call[1]))]
    if k % 3 != 1:
        flags.append("--extend_out")
    return flags, o.run(), n_sites


@pytest.mark.parametrize("k", list(range(0, 48)) + list(range(10_000, 10_006)) + list(range(30_000, 30_006)))   # (tools/cli_soak.py: 2,900 more)
def test_random_case_through_both_programs(k, tmp_path):
    d = str(tmp_path)
    flags, rec, n_sites = case_files(k, d)
    got, want = both_programs(flags, rec, n_sites, d, threads=1 + k % 3)
    assert same_tsv(got, want) is None, f"case {k}: {same_tsv(got, want)}\n{' '.join(flags)}"


def _geno_text(mat, header: bool, prefix_cols: bool) -> str:
    """A beagle-like text genotype file (as tests/golden/make_golden.py writes the f8_* fixtures): optional header line,
    optional non-numeric leading columns; small integers as integers, everything else with every digit."""
    lines = []
    if header:
        lines.append("marker\tallele1\tallele2\t" + "\t".join(f"Ind{i}" for i in range(mat.shape[1])))
    for s in range(mat.shape[0]):
        pre = f"chr1_{s}\tA\tC\t" if prefix_cols else ""
        lines.append(pre + "\t".join(str(int(x)) if np.isfinite(x) and float(x) == int(x) and abs(x) <= 9 else repr(float(x))
                                     for x in mat[s].reshape(-1)))
    return "\n".join(lines) + "\n"


def text_case_files(k: int, d: str):
    """Text (.gz) genotype input, case k: called genotypes (0 / 1 / 2, -1 = no data) or likelihood triples (normal or log
    scale, optionally hardened by --call_geno) -- the text branch of read_geno (read_data.cpp:50-104)."""
    import ctypes as C
    import gzip
    rng = np.random.default_rng(88_000 + k)
    called = k % 2 == 0
    n_ind = int(rng.choice([1, 2, 7, 20, 64, 65, 100, 130, 257, 500, 520]))
    n_sites = int(rng.integers(4, 50 if n_ind <= 130 else 16))
    log_scale = (not called) and rng.random() < 0.3
    call = None
    if called:
        p_miss = float(rng.choice([0.0, 0.1, 0.5]))
        q = rng.uniform(0.05, 0.5, size=n_sites)
        g = rng.binomial(2, q[:, None], size=(n_sites, n_ind)).astype(float)
        g[rng.random((n_sites, n_ind)) < p_miss] = -1.0
        if rng.random() < 0.3:
            g[int(rng.integers(0, n_sites))] = 0.0                      # a monomorphic site
        mat, prefix = g, bool(rng.random() < 0.5)
    else:
        raw = synth.make_gl_numpy(n_sites, n_ind, 88_500 + k, depth=float(rng.choice([0.5, 2.0, 8.0])))
        raw /= raw.sum(axis=2, keepdims=True)
        if rng.random() < 0.4:
            hc = rng.random((n_sites, n_ind)) < 0.1
            raw[hc] = np.eye(3)[rng.integers(0, 3, size=int(hc.sum()))]   # exact zeros: log(0) stays -inf in the text branch
        if log_scale:
            with np.errstate(divide="ignore"):
                raw = np.log(raw)
        if rng.random() < 0.3:
            call = tuple(sorted(rng.random(2)))
        mat, prefix = raw.reshape(n_sites, -1), True
    gpath, ppath = os.path.join(d, "in.geno.gz"), os.path.join(d, "in.pos")
    with gzip.open(gpath, "wt") as fh:
        fh.write(_geno_text(mat, header=bool(rng.random() < 0.5), prefix_cols=prefix))
    chrs, pos = synth.make_positions(n_sites, 88_900 + k, max_gap=int(rng.choice([5, 200, 3000])), n_chr=int(rng.integers(1, 3)))
    synth.write_pos(ppath, chrs, pos)
    from ngsld_amd import shard
    pd = shard.pos_dist_from_positions(chrs, pos)
    ignore = bool(rng.random() < 0.5)
    max_kb, max_snp = int(rng.choice([0, 0, 1, 50])), int(rng.choice([0, 0, 5]))
    gl = np.empty((n_sites, n_ind, 3))
    err = C.create_string_buffer(256)
    rc = orc.lib().orc_read_geno_text(gpath.encode(), int(not called), int(log_scale), n_ind, n_sites, orc.dp(gl), err, 256)
    assert rc == 0, err.value
    o = orc.Oracle(gl, pd, already_normalised_log=True, ignore_miss_data=ignore, max_kb_dist=max_kb, max_snp_dist=max_snp,
                   n_threads=4, call_geno=call)
    flags = ["--geno", gpath, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--verbose", "0", "--pos", ppath,
             "--max_kb_dist", str(max_kb), "--max_snp_dist", str(max_snp), "--min_maf", "0"]
    if not called:
        flags.append("--probs")
    if log_scale:
        flags.append("--log_scale")
    if ignore:
        flags.append("--ignore_miss_data")
    if call is not None:
        flags += ["--call_geno", "--N_thresh", repr(float(call[0])), "--call_thresh", repr(float(call[1]))]
    if k % 3 != 1:
        flags.append("--extend_out")
    return flags, o.run(), n_sites


@pytest.mark.parametrize("k", range(24))   # (tools/cli_soak.py ... text: 400 more)
def test_random_text_input_through_both_programs(k, tmp_path):
    d = str(tmp_path)
    flags, rec, n_sites = text_case_files(k, d)
    got, want = both_programs(flags, rec, n_sites, d, threads=1 + k % 3)
    assert same_tsv(got, want) is None, f"text case {k}: {same_tsv(got, want)}\n{' '.join(flags)}"


@pytest.mark.parametrize("k", list(range(200, 214)) + list(range(10_020, 10_022)))
@pytest.mark.parametrize("how", ["slabs", "parts"])
def test_streamed_and_multi_part_runs_through_both_programs(k, how, tmp_path):
    """The drop-in binary's own ways of cutting a job -- row slabs streamed through two alternating contexts
    (NGSLD_TEST_SLAB_SITES: what --max_gpu_mem does to a matrix beyond the budget) and several parts in one process
    (--devices 0,0,0: three parts on this box's one GPU) -- write the reference program's table too."""
    d = str(tmp_path)
    flags, rec, n_sites = case_files(k, d)
    if how == "slabs":
        got, want = both_programs(flags, rec, n_sites, d, threads=2, hip_env={"NGSLD_TEST_SLAB_SITES": str(max(2, n_sites // 4))})
    else:
        got, want = both_programs(flags, rec, n_sites, d, threads=2, hip_flags=("--devices", "0,0,0"))
    assert same_tsv(got, want) is None, f"case {k} ({how}): {same_tsv(got, want)}\n{' '.join(flags)}"


def test_edge_cases_through_both_programs():
    """tools/cli_edge_cases.py: what lies below and beside the fuzz generator's cases (one / two sites, one / two individuals,
    nothing but monomorphic sites or missing data, empty windows, a chromosome per site, thresholds that drop everything, files
    shorter or longer than --n_sites, positions that repeat or go backwards) -- the binary writes the reference program's table
    or ends with the reference program's error line (94 runs, twelve over compressed / absent / unwritable files and positions files with comments, CRLF, octal-looking or scientific numbers, no usable line, uneven fields, two with the table on standard output, 20 of them text genotype files: headers, blank lines, CRLF, label columns, fractions, words, too few / too many rows)."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(capi.REPO_DIR, "tools", "cli_edge_cases.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "94 through both programs, 0 differ" in r.stdout
