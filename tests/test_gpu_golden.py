"""GPU: the HIP path (through the C-ABI) against the committed golden vectors -- hap / nIter / sample_size
produced by the reference's own compiled EM, the derived columns by the oracle."""
import hashlib
import subprocess
import tempfile

import numpy as np
import pytest

from ngsld_amd import capi
from util import MAF_TOL, Fixture, check_records, close, fixtures

pytestmark = pytest.mark.gpu


def _want(fx):
    return {"hap": fx["ref_hap"], "n_iter": fx["ref_n_iter"], "n_ind_data": fx["ref_n_ind_data"], "D": fx["orc_D"],
            "Dp": fx["orc_Dp"], "r2": fx["orc_r2"], "r2pear": fx["orc_r2pear"], "hap_maf": fx["orc_hap_maf"]}


@pytest.mark.parametrize("name", fixtures())
def test_hip_matches_golden(engine, name, tmp_path):
    fx = Fixture(name)
    fx.engine_load(engine, str(tmp_path))
    assert np.all(close(engine.maf(), fx["ref_maf"], MAF_TOL)), "est_maf vs reference"
    n = engine.plan(fx.max_kb, fx.max_snp, fx.min_maf, fx.ignore_miss, True, fx.rnd_sample, fx.seed)
    assert n == len(fx["orc_s1"])
    s1, s2, std, ext = engine.run()
    assert np.array_equal(s1, fx["orc_s1"]) and np.array_equal(s2, fx["orc_s2"])
    check_records(std, ext, _want(fx))


def _parse(txt):
    rows = [l.rstrip("\n").split("\t") for l in txt.splitlines()[1:]]
    return rows


@pytest.mark.parametrize("name", [n for n in fixtures() if "orc_tsv_std_md5" in Fixture(n)])
@pytest.mark.parametrize("extend", [False, True])
def test_cli_text_parity(name, extend):
    """The ngsLD drop-in binary end to end: same flags in, same TSV out as the oracle's text -- md5 of the sorted body,
    as examples/test.sh does -- on EVERY fixture, the degenerate ones included: the -nan / 0.000000 / inf of a
    monomorphic site and the sign of a rounded zero are the reference's own (exact-order replay)."""
    fx = Fixture(name)
    tag = "ext" if extend else "std"
    with tempfile.TemporaryDirectory() as d:
        g, p = fx.write_inputs(d)
        cmd = [capi.CLI_PATH, "--geno", g, "--n_ind", str(fx.n_ind), "--n_sites", str(fx.n_sites), "--verbose", "0"]
        if p:
            cmd += ["--posH" if fx.header else "--pos", p]
        r = subprocess.run(cmd + fx.cli_flags(extend), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines(keepends=True)
    assert lines[0] == str(fx[f"orc_tsv_{tag}_header"])
    md5 = hashlib.md5((lines[0] + "".join(sorted(lines[1:]))).encode()).hexdigest()
    if md5 != str(fx[f"orc_tsv_{tag}_md5"]) and f"orc_tsv_{tag}" in fx:      # say where, then fail
        want, got = _parse(str(fx[f"orc_tsv_{tag}"])), _parse(r.stdout)
        assert len(got) == len(want)
        diff = [(k, a, b) for k, (a, b) in enumerate(zip(got, want)) if a != b]
        assert not diff, f"{len(diff)} lines differ from the oracle's text, first: {diff[0]}"
    assert md5 == str(fx[f"orc_tsv_{tag}_md5"])


def test_cli_errors_like_the_reference(tmp_path):
    fx = Fixture("f3_degenerate")
    g, p = fx.write_inputs(str(tmp_path))
    base = [capi.CLI_PATH, "--geno", g, "--n_ind", str(fx.n_ind), "--verbose", "0"]
    r = subprocess.run(base + ["--n_sites", str(fx.n_sites)], capture_output=True, text=True)
    assert r.returncode == 255 and "position file necessary in order to filter by maximum distance!" in r.stderr
    r = subprocess.run(base + ["--n_sites", str(fx.n_sites + 1), "--pos", p], capture_output=True, text=True)
    assert r.returncode == 255 and "invalid/corrupt genotype input file!" in r.stderr
    r = subprocess.run(base + ["--pos", p], capture_output=True, text=True)
    assert r.returncode == 255 and "number of sites (--n_sites) missing!" in r.stderr
    r = subprocess.run(base + ["--n_sites", str(fx.n_sites), "--pos", p, "--outH", "x"], capture_output=True, text=True)
    assert r.returncode == 255                                        # declared flag without a case: exit(-1)
    bad = fx.raw.copy()
    bad[1, 2, :] = -1.0                                               # log(-1) = NaN
    bad.tofile(g)
    r = subprocess.run(base + ["--n_sites", str(fx.n_sites), "--pos", p], capture_output=True, text=True)
    assert r.returncode == 255 and "NaN found! Is the file format correct?" in r.stderr
