"""GPU: the drop-in boundary in situ -- the reference's OWN main() with the library plugged in.

oracle/build_ref.sh compiles ngsLD.cpp's main() from where it lies with its thread-pool section (ngsLD.cpp:153-198: pool creation,
one calc_pair_LD job per site, wait, destroy) replaced by the one call of integration/ngsld_binding.h -- the patch INTEGRATION.md
section 2 shows a maintainer -- and links libngsld.so (oracle/_ref/libngsld_ref_hip.so, entry ref_main_hip).  Argument parsing,
read_geno, call_geno, est_maf, the exp() loop, read_dist, labels, the output file and its header are the reference's text; the
pairs come from the device through ngsld_set_geno_lkl / ngsld_plan / ngsld_run, the rows either formatted on the device
(default) or printed by the reference's own print block from the records (NGSLD_BINDING_TEXT=0).  Its TSV must be the
UNPATCHED reference program's (tests/util.py run_ref_program) over the same argv and files -- first line equal, bodies equal
as sorted lines -- and, for --rnd_sample (which the unpatched build cannot run here: gsl_rng), the golden md5 of the oracle's."""
import hashlib
import os

import pytest

from test_gpu_vs_ref_program import case_files, same_tsv, text_case_files
from util import Fixture, fixtures, have_patched_ref_program, have_ref_program, run_patched_ref_program, run_ref_program

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (have_ref_program() and have_patched_ref_program()),
                                 reason="oracle/_ref predates ref_main_hip (rebuild with oracle/build_ref.sh)")]

MODES = [("text", {}), ("records", {"NGSLD_BINDING_TEXT": "0"})]


def patched_and_unpatched(flags, rec, n_sites, d, threads, env):
    out_hip, out_ref = os.path.join(d, "patched.tsv"), os.path.join(d, "ref.tsv")
    h = run_patched_ref_program(flags, out_hip, threads, env=env)
    assert h.returncode == 0, h.stderr[-2000:]
    r = run_ref_program(rec, n_sites, flags, out_ref, d, threads)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out_hip).read(), open(out_ref).read()


def fixture_flags(fx, extend, d):
    g, p = fx.write_inputs(d)
    flags = ["--geno", g, "--n_ind", str(fx.n_ind), "--n_sites", str(fx.n_sites), "--verbose", "0"]
    if p:
        flags += ["--posH" if fx.header else "--pos", p]
    return flags + fx.cli_flags(extend)


@pytest.mark.parametrize("name", [n for n in fixtures() if Fixture(n).rnd_sample >= 1 and Fixture(n).n_ind <= 500])
@pytest.mark.parametrize("extend", [False, True])
@pytest.mark.parametrize("mode,env", MODES)
def test_patched_main_on_a_fixture(name, extend, mode, env, tmp_path):
    fx = Fixture(name)
    d = str(tmp_path)
    got, want = patched_and_unpatched(fixture_flags(fx, extend, d), fx.oracle().run(), fx.n_sites, d, 2, env)
    assert same_tsv(got, want) is None, same_tsv(got, want)


@pytest.mark.parametrize("name", [n for n in fixtures() if Fixture(n).rnd_sample < 1 and "orc_tsv_ext_md5" in Fixture(n)])
@pytest.mark.parametrize("mode,env", MODES)
def test_patched_main_with_rnd_sample_writes_the_golden_table(name, mode, env, tmp_path):
    """--rnd_sample / --seed: the per-row Tausworthe streams are the library's (the patched main hands it rnd_sample and seed,
    ngsLD.cpp:160-166 is part of the section the binding replaces); the unpatched build has no gsl_rng here, the golden md5 is
    the oracle's."""
    fx = Fixture(name)
    d = str(tmp_path)
    out = os.path.join(d, "patched.tsv")
    h = run_patched_ref_program(fixture_flags(fx, True, d), out, 2, env=env)
    assert h.returncode == 0, h.stderr[-2000:]
    lines = open(out).read().splitlines(keepends=True)
    assert lines[0] == str(fx["orc_tsv_ext_header"])
    assert hashlib.md5((lines[0] + "".join(sorted(lines[1:]))).encode()).hexdigest() == str(fx["orc_tsv_ext_md5"])


@pytest.mark.parametrize("k", list(range(0, 16)) + list(range(10_000, 10_002)) + list(range(30_000, 30_002)))
@pytest.mark.parametrize("mode,env", MODES)
def test_patched_main_on_a_random_case(k, mode, env, tmp_path):
    d = str(tmp_path)
    flags, rec, n_sites = case_files(k, d)
    got, want = patched_and_unpatched(flags, rec, n_sites, d, 1 + k % 3, env)
    assert same_tsv(got, want) is None, f"case {k} ({mode}): {same_tsv(got, want)}\n{' '.join(flags)}"


@pytest.mark.parametrize("k", range(10))
def test_patched_main_on_a_random_text_input(k, tmp_path):
    d = str(tmp_path)
    flags, rec, n_sites = text_case_files(k, d)
    got, want = patched_and_unpatched(flags, rec, n_sites, d, 1 + k % 3, MODES[k % 2][1])
    assert same_tsv(got, want) is None, f"text case {k}: {same_tsv(got, want)}\n{' '.join(flags)}"
