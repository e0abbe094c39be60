"""The reference's OWN program flow, run here on the fixtures (CPU).

oracle/build_ref.sh compiles ngsLD.cpp's main() and calc_pair_LD as they stand but for the statements that need GSL: the six
gsl_rng statements and the --rnd_sample block are dropped (fixtures with --rnd_sample are skipped), the one pearson_r call is
replaced by a lookup of the value the caller supplies for that pair.  ref_main(argc, argv) is then the reference's argument
parsing, file checks, reader, call_geno loop, est_maf loop, exp / expected genotypes, positions and labels, thread pool,
per-row walk, EM, statistics and fprintf on real files.  Its TSV must be the oracle CLI's byte for byte (one thread: same
order; several: as sorted lines) -- the assembly of the pieces the other tests pin one by one.  The r2_ExpG column is the
oracle's own (GSL), handed in: the one column this does not pin."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import orc
from util import Fixture, fixtures, have_ref_program, run_ref_program

pytestmark = pytest.mark.skipif(not have_ref_program(), reason="oracle/_ref predates ref_main (rebuild with oracle/build_ref.sh)")

NAMES = [n for n in fixtures() if Fixture(n).rnd_sample >= 1 and Fixture(n).n_ind <= 500]


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("extend,threads", [(False, 1), (True, 1), (True, 3)])
def test_reference_program_flow_writes_the_oracles_tsv(name, extend, threads, tmp_path):
    fx = Fixture(name)
    d = str(tmp_path)
    g, p = fx.write_inputs(d)
    rec = fx.oracle().run()
    base = ["--geno", g, "--n_ind", str(fx.n_ind), "--n_sites", str(fx.n_sites), "--verbose", "0"]
    if p:
        base += ["--posH" if fx.header else "--pos", p]
    flags = base + fx.cli_flags(extend)
    out_ref = os.path.join(d, "ref.tsv")
    r = run_ref_program(rec, fx.n_sites, flags, out_ref, d, threads)
    assert r.returncode == 0, r.stderr[-2000:]
    want = subprocess.run([orc.ORC_CLI, *flags], check=True, capture_output=True, text=True).stdout
    got = open(out_ref).read()
    if threads == 1:
        assert got == want, "the reference's program flow and the oracle CLI write different bytes"
    else:
        gl, wl = got.splitlines(keepends=True), want.splitlines(keepends=True)
        assert gl[0] == wl[0] and sorted(gl[1:]) == sorted(wl[1:])
    tag = "ext" if extend else "std"
    if f"orc_tsv_{tag}_md5" in fx:   # and the golden md5 (sorted body, as the reference's own test sorts: examples/test.sh:16)
        lines = got.splitlines(keepends=True)
        assert hashlib.md5((lines[0] + "".join(sorted(lines[1:]))).encode()).hexdigest() == str(fx[f"orc_tsv_{tag}_md5"])


@pytest.mark.parametrize("k", list(range(0, 48)) + list(range(10_000, 10_006)) + list(range(30_000, 30_006)))
def test_reference_program_on_random_cases_writes_the_oracles_tsv(k, tmp_path):
    """The fuzz generator's cases as files (tests/test_gpu_vs_ref_program.py hands the same command lines to the HIP binary):
    every filter, chromosome breaks, --log_scale, --ignore_miss_data, --call_geno, thresholds ON a site's frequency."""
    from test_gpu_vs_ref_program import case_files, same_tsv
    d = str(tmp_path)
    flags, rec, n_sites = case_files(k, d)
    out_ref = os.path.join(d, "ref.tsv")
    r = run_ref_program(rec, n_sites, flags, out_ref, d, threads=1 + k % 3)
    assert r.returncode == 0, r.stderr[-2000:]
    want = subprocess.run([orc.ORC_CLI, *flags], check=True, capture_output=True, text=True).stdout
    assert same_tsv(open(out_ref).read(), want) is None, f"case {k}: {same_tsv(open(out_ref).read(), want)}\n{' '.join(flags)}"


@pytest.mark.parametrize("k", range(24))
def test_reference_program_on_random_text_inputs_writes_the_oracles_tsv(k, tmp_path):
    """Text (.gz) genotype files -- called genotypes, likelihood triples in normal and log scale, --call_geno -- through the
    reference's program and the oracle CLI (the HIP binary gets the same command lines in tests/test_gpu_vs_ref_program.py)."""
    from test_gpu_vs_ref_program import same_tsv, text_case_files
    d = str(tmp_path)
    flags, rec, n_sites = text_case_files(k, d)
    out_ref = os.path.join(d, "ref.tsv")
    r = run_ref_program(rec, n_sites, flags, out_ref, d, threads=1 + k % 3)
    assert r.returncode == 0, r.stderr[-2000:]
    want = subprocess.run([orc.ORC_CLI, *flags], check=True, capture_output=True, text=True).stdout
    assert same_tsv(open(out_ref).read(), want) is None, f"text case {k}: {same_tsv(open(out_ref).read(), want)}\n{' '.join(flags)}"


def test_patched_reference_main_fails_loudly_without_a_device(tmp_path):
    """oracle/_ref/libngsld_ref_hip.so -- the reference's main() with its thread-pool section replaced by the binding of
    integration/ngsld_binding.h -- on a box without a GPU: the reference's own code parses, reads and estimates, then
    ngsld_create refuses and the program ends through the reference's error() (exit status 255).  No CPU fallback.
    (With a GPU: tests/test_gpu_ref_main_patched.py.)"""
    import torch
    from util import have_patched_ref_program, run_patched_ref_program
    if not have_patched_ref_program():
        pytest.skip("oracle/_ref predates ref_main_hip (rebuild with oracle/build_ref.sh)")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    fx = Fixture(NAMES[0])
    d = str(tmp_path)
    g, p = fx.write_inputs(d)
    flags = ["--geno", g, "--n_ind", str(fx.n_ind), "--n_sites", str(fx.n_sites), "--verbose", "0"] + (["--pos", p] if p else []) + fx.cli_flags(True)
    r = run_patched_ref_program(flags, os.path.join(d, "out.tsv"))
    assert r.returncode == 255 and "no HIP device available" in r.stderr and "no CPU fallback" in r.stderr


def test_edge_cases_through_the_reference_program_and_the_oracle_cli():
    """tools/cli_edge_cases.py with the oracle's CLI in the binary's place: one / two sites, one / two individuals, nothing but
    monomorphic sites or missing data, empty windows, a chromosome per site, thresholds that drop everything, files shorter or
    longer than --n_sites, positions that repeat or go backwards -- same table, or the same error line (94 runs, twelve over compressed / absent / unwritable files and positions files with comments, CRLF, octal-looking or scientific numbers, no usable line, uneven fields, two with the table on standard output, 20 of them text genotype files: headers, blank lines, CRLF, label columns, fractions, words, too few / too many rows)."""
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(orc.HERE), "tools", "cli_edge_cases.py"), "--binary", orc.ORC_CLI],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "94 through both programs, 0 differ" in r.stdout
