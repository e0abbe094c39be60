"""The oracle against the committed golden vectors (CPU).  `ref_*` fields were produced by the reference's
own compiled functions and -- D / D' / r2 / hap_maf / chi2, the s2 walk, the TSV rows -- by the GSL-free lines of
ngsLD.cpp compiled from where they lie (tests/golden/make_golden.py, oracle/build_ref.sh), so this pins the oracle
wherever the reference can be built; the `orc_*` fields without a `ref_*` twin (r2_ExpG: gsl_stats_correlation, and
the --rnd_sample draws: gsl_rng_taus) freeze the oracle's own output for the one unpinned dependency, GSL."""
import hashlib
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import orc
from util import Fixture, fixtures


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", fixtures())
def test_oracle_reproduces_golden(name):
    fx = Fixture(name)
    o = fx.oracle()
    # reference-pinned stages: reader (+ call_geno), maf, preprocessing (bit for bit)
    assert _sha(o.gl_log) == str(fx["ref_reader_sha"])
    assert _sha(o.gl) == str(fx["ref_gl_sha"]) and _sha(o.expg) == str(fx["ref_expg_sha"])
    assert np.array_equal(o.maf, fx["ref_maf"], equal_nan=True)
    rec = o.run()
    assert np.array_equal(rec["s1"], fx["orc_s1"]) and np.array_equal(rec["s2"], fx["orc_s2"])
    # reference-pinned EM
    assert np.array_equal(rec["hap"], fx["ref_hap"], equal_nan=True)
    assert np.array_equal(rec["n_iter"], fx["ref_n_iter"]) and np.array_equal(rec["n_ind_data"], fx["ref_n_ind_data"])
    # reference-pinned statistics: ngsLD.cpp:296-306 and :328-333, compiled from the reference's text -- bit for bit
    def bits(a):
        a = np.ascontiguousarray(a)
        return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)
    for col, key in (("D", "ref_D"), ("Dp", "ref_Dp"), ("r2", "ref_r2"), ("hap_maf", "ref_hap_maf"), ("chi2", "ref_chi2")):
        assert np.array_equal(bits(rec[col].copy()), bits(fx[key])), col
    # reference-pinned walk (ngsLD.cpp:240-275): which pairs, and their running distance
    if float(fx["rnd_sample"]) >= 1:
        assert np.array_equal(rec["s1"], fx["ref_walk_s1"]) and np.array_equal(rec["s2"], fx["ref_walk_s2"])
        assert np.array_equal(bits(rec["dist"].copy()), bits(fx["ref_walk_dist"]))
    # frozen oracle output (r2_ExpG is the unpinned one: GSL; the rest repeats the reference-pinned fields above)
    for col, key in (("dist", "orc_dist"), ("r2pear", "orc_r2pear"), ("D", "orc_D"), ("Dp", "orc_Dp"), ("r2", "orc_r2"),
                     ("hap_maf", "orc_hap_maf"), ("chi2", "orc_chi2")):
        assert np.array_equal(rec[col], fx[key], equal_nan=True), col
    if "orc_tsv_std_md5" in fx:   # the generator held every TSV row to the reference's own fprintf lines (ngsLD.cpp:314-351)
        assert bool(fx["ref_tsv_std_rows_equal"]) and bool(fx["ref_tsv_ext_rows_equal"])


@pytest.mark.parametrize("name", [n for n in fixtures() if "orc_tsv_std_md5" in Fixture(n)])
@pytest.mark.parametrize("extend", [False, True])
def test_oracle_cli_text(name, extend):
    fx = Fixture(name)
    tag = "ext" if extend else "std"
    with tempfile.TemporaryDirectory() as d:
        g, p = fx.write_inputs(d)
        cmd = [orc.ORC_CLI, "--geno", g, "--n_ind", str(fx.n_ind), "--n_sites", str(fx.n_sites), "--verbose", "0"]
        if p:
            cmd += ["--posH" if fx.header else "--pos", p]
        txt = subprocess.run(cmd + fx.cli_flags(extend), check=True, capture_output=True, text=True).stdout
    lines = txt.splitlines(keepends=True)
    assert lines[0] == str(fx[f"orc_tsv_{tag}_header"])
    md5 = hashlib.md5((lines[0] + "".join(sorted(lines[1:]))).encode()).hexdigest()
    assert md5 == str(fx[f"orc_tsv_{tag}_md5"])


def test_pearson_against_textbook():
    """gsl_stats_correlation is restated from GSL's published algorithm (GSL itself is absent): hold it to the
    textbook formula at 1e-12 (parity unpinned at the GSL boundary, see oracle/ngsld_oracle.h)."""
    rng = np.random.default_rng(3)
    for n in (2, 3, 24, 500, 2000):
        x, y = rng.random(n) * 2, rng.random(n) * 2
        got = orc.lib().orc_correlation(orc.dp(x), orc.dp(y), n)
        assert abs(got - np.corrcoef(x, y)[0, 1]) < 1e-12
    x = np.full(10, 0.7)
    assert np.isnan(orc.lib().orc_correlation(orc.dp(x), orc.dp(rng.random(10)), 10))   # zero variance -> 0/0


def test_em_known_answers():
    """Closed-form cases of the 4-haplotype EM (independent of any reference build)."""
    L = orc.lib()
    import ctypes as C
    n = 40
    # hard calls, every individual double-homozygous: haplotype counts are observed directly
    g1 = np.array([0] * 10 + [2] * 30)
    g2 = np.array([0] * 10 + [2] * 10 + [0] * 20)
    a, b = np.eye(3)[g1], np.eye(3)[g2]
    hap, nn, e = np.zeros(4), C.c_uint64(), C.c_int(0)
    it = L.orc_haplo_freq(orc.dp(hap), C.byref(nn), orc.dp(np.ascontiguousarray(a)), orc.dp(np.ascontiguousarray(b)),
                          0.75, 0.25, n, 0, C.byref(e))
    # haplotypes: 10 ind x (0,0) -> h00, 10 ind x (1,1) -> h11, 20 ind x (1,0) -> h10
    assert np.allclose(hap, [0.25, 0.0, 0.5, 0.25], atol=1e-12) and nn.value == n and it <= 2 and e.value == 0


def test_taus_known_answer_and_row_seeds():
    """gsl_rng_taus restated from GSL's published algorithm (GSL is absent): GSL's own self-test value
    (rng/test.c: seed 1, 10000th output 2733957125) pins the generator; row seeds are the serial master stream."""
    import ctypes as C
    L = orc.lib()
    st = (C.c_uint32 * 3)()
    L.orc_taus_set(st, 1)
    v = 0
    for _ in range(10000):
        v = L.orc_taus_get(st)
    assert v == 2733957125
    L.orc_taus_set(st, 0)                      # seed 0 is replaced by 1
    first0 = L.orc_taus_get(st)
    L.orc_taus_set(st, 1)
    assert first0 == L.orc_taus_get(st)
    seeds = (C.c_uint64 * 5)()
    L.orc_row_seeds(42, 5, seeds)
    L.orc_taus_set(st, 42)
    for k in range(5):
        assert seeds[k] == int(L.orc_taus_get(st) / 4294967296.0 * 1e15)


def test_pearson_against_exact_rational_arithmetic():
    """How far ANY faithful evaluation of Pearson's r can be from the oracle's: r^2 = Sxy^2 / (Sxx Syy) evaluated EXACTLY on
    the doubles (Python fractions: every double is a rational) and rounded once, against the square of orc_correlation (GSL's
    one-pass recurrence with long double accumulators, restated -- GSL itself is not in the image).  Expected-genotype
    vectors of the shapes the path sees: values in [0, 2], cohorts of 2..500, including nearly constant ones (the
    ill-conditioned case).  The oracle is within 2e-15 of the exact value of r^2 (a handful of ulps of 1) -- so is any
    implementation that carries the sums in extended precision, GSL's included -- five orders of magnitude inside the
    1e-9 the north star asks for: what "parity unpinned at the GSL boundary" can cost is bounded by that."""
    from fractions import Fraction
    rng = np.random.default_rng(11)
    worst = 0.0
    for case in range(120):
        n = int(rng.choice([2, 3, 8, 24, 100, 500]))
        if case % 3 == 0:      # expected genotypes of a well-typed site pair
            x, y = rng.random(n) * 2, rng.random(n) * 2
        elif case % 3 == 1:    # correlated pair
            x = rng.random(n) * 2
            y = np.clip(x + rng.normal(0, 0.05, n), 0, 2)
        else:                  # nearly constant vectors: 1 / (std1 std2) large
            x = 1.0 + rng.normal(0, 1e-6, n)
            y = 0.3 + rng.normal(0, 1e-6, n)
        fx, fy = [Fraction(float(v)) for v in x], [Fraction(float(v)) for v in y]
        mx, my = sum(fx) / n, sum(fy) / n
        sxy = sum((a - mx) * (b - my) for a, b in zip(fx, fy))
        sxx = sum((a - mx) ** 2 for a in fx)
        syy = sum((b - my) ** 2 for b in fy)
        if sxx == 0 or syy == 0:
            continue
        exact = float(sxy * sxy / (sxx * syy))
        got = orc.lib().orc_correlation(orc.dp(np.ascontiguousarray(x)), orc.dp(np.ascontiguousarray(y)), n) ** 2
        err = abs(got - exact)
        if case % 3 != 2:
            assert err < 2e-15, (case, n, got, exact)
            worst = max(worst, err)
        else:  # the cancellation in the means costs digits in proportion to 1 / std: still far inside 1e-9
            assert err < 1e-9, (case, n, got, exact)
    print(f"oracle r2_ExpG vs exact rational arithmetic: worst absolute difference {worst:.2e} (well-conditioned cases)")
