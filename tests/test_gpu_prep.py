"""GPU: the per-site prep kernels (ld_prep.hip).  Likelihood triples take a quotient fast path, a_k = raw_k / sum, wherever
the reference's chain of logs (read_data.cpp:37-45, gen_func.cpp:974-1009, ngsLD.cpp:110) has no special behaviour, and
the chain itself everywhere else; NGSLD_TEST_PREP_EXACT=1 sends every triple through the chain.  Both must agree with the
reference's compiled est_maf (the oracle, bit-checked against it) to 1e-12 and with each other far inside that, on ordinary
triples and on every special one: zeros, all-zero triples, called genotypes, denormal and huge values."""
import os

import numpy as np
import pytest

from ngsld_amd import capi, synth
from oracle import orc
from util import MAF_TOL, check_records, close

pytestmark = pytest.mark.gpu


def _special_matrix(n_sites, n_ind, seed):
    raw = synth.make_gl_numpy(n_sites, n_ind, seed, depth=4.0)
    rng = np.random.default_rng(seed)
    raw[1, ::4] = [1.0, 0.0, 0.0]                      # zeros inside a triple (fast path: exact 1 / 0)
    raw[2, 1::5] = 0.0                                 # all-zero triples: the chain's -1e15 arithmetic (0.3247 each)
    raw[3] = np.eye(3)[rng.integers(0, 3, n_ind)]      # a called site
    raw[4, ::3] *= 1e-305                              # denormal range: the chain
    raw[5, ::3] *= 1e300                               # huge: the chain
    raw[6] = 1.0 / 3.0                                 # nobody has data
    raw[7, ::2] = [0.2, 0.2, 0.2]                      # un-normalised "no data"
    raw[8] *= rng.uniform(1e-30, 1e30, size=(n_ind, 1))  # wildly different scales per individual
    return raw


@pytest.mark.parametrize("n_ind", [24, 100, 500, 700, 1000, 2500, 6000])
@pytest.mark.parametrize("ignore_miss", [False, True])
def test_unnormalised_triples_at_a_relabelled_site(n_ind, ignore_miss):
    """All-zero natural-scale triples leave the reference's chain as 0.3247 three times: they do not sum to 1.  At a site
    whose allele frequency is above 1/2 the pair kernels relabel the alleles and take the Pearson moment from 2 - e, which
    such a triple breaks (r2_ExpG off in the second decimal): those sites carry a sign in their rsx and their pairs are
    replayed.  Sites 2 and 9 here have maf > 1/2 AND all-zero triples -- as row site and as candidate of every kernel family;
    site 5 has the triples with maf < 1/2 (no relabelling, no replay needed)."""
    n_sites = 24
    raw = synth.make_gl_numpy(n_sites, n_ind, 8100 + n_ind, depth=6.0)
    for s in (2, 9):
        raw[s] = raw[s, :, ::-1]                        # swap genotypes 0 and 2: frequency 1 - q
        raw[s, 1::5] = 0.0
    raw[5, ::4] = 0.0
    o = orc.Oracle(raw, ignore_miss_data=ignore_miss, n_threads=8)
    assert o.maf[2] > 0.5 and o.maf[9] > 0.5 and o.maf[5] < 0.5
    want = o.run()
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw, ignore_miss_data=ignore_miss)
        eng.set_pos_dist(None)
        assert eng.plan(ignore_miss_data=ignore_miss) == len(want)
        s1, s2, std, ext = eng.run()
        replayed = eng.replay_stats()[0]
    finally:
        eng.close()
    check_records(std, ext, want)
    touched = int(np.sum((want["s1"] == 2) | (want["s2"] == 2) | (want["s1"] == 9) | (want["s2"] == 9)))
    assert replayed >= touched                         # every pair of the two relabelled sites went through the replay


@pytest.mark.parametrize("n_ind", [24, 100, 500, 700, 1500, 2500])
@pytest.mark.parametrize("ignore_miss", [False, True])
def test_fast_path_and_chain_agree_with_the_reference(n_ind, ignore_miss):
    """n_ind 24..1500: one wavefront per site (8 / 16 / 32 individuals per lane); 2500: one workgroup per site."""
    n_sites = 40
    raw = _special_matrix(n_sites, n_ind, 7000 + n_ind)
    o = orc.Oracle(raw, ignore_miss_data=ignore_miss, n_threads=8)
    want = o.run()
    got = {}
    for mode in ("fast", "chain"):
        if mode == "chain":
            os.environ["NGSLD_TEST_PREP_EXACT"] = "1"
        try:
            eng = capi.Engine(0)
            try:
                eng.set_geno_raw(raw, ignore_miss_data=ignore_miss)
                maf = eng.maf()
                eng.set_pos_dist(None)
                assert eng.plan(ignore_miss_data=ignore_miss) == len(want)
                s1, s2, std, ext = eng.run()
            finally:
                eng.close()
        finally:
            os.environ.pop("NGSLD_TEST_PREP_EXACT", None)
        assert np.all(close(maf, o.maf, MAF_TOL)), mode            # 1e-12 against the reference's est_maf
        check_records(std, ext, want)
        got[mode] = (maf, std, ext)
    both = np.isfinite(got["fast"][0]) & np.isfinite(got["chain"][0])
    assert np.array_equal(np.isfinite(got["fast"][0]), np.isfinite(got["chain"][0]))
    assert np.max(np.abs(got["fast"][0][both] - got["chain"][0][both]), initial=0.0) < 2e-14
    assert np.array_equal(got["fast"][2]["n_iter"], got["chain"][2]["n_iter"])
    assert np.array_equal(got["fast"][2]["n_ind_data"], got["chain"][2]["n_ind_data"])


def test_nan_input_is_reported_on_both_paths():
    raw = synth.make_gl_numpy(20, 64, 7100, depth=4.0)
    raw[11, 5, 1] = -0.25                               # log of a negative value: NaN -> "NaN found!" (read_data.cpp:42-45)
    for exact in ("0", "1"):
        os.environ["NGSLD_TEST_PREP_EXACT"] = exact
        try:
            eng = capi.Engine(0)
            try:
                with pytest.raises(capi.NgsldError) as e:
                    eng.set_geno_raw(raw)
                assert e.value.code == capi.ERR_NAN
            finally:
                eng.close()
        finally:
            os.environ.pop("NGSLD_TEST_PREP_EXACT", None)


def test_values_at_the_edges_of_the_double_range_end_like_the_reference_reader():
    """tools/probe_special_values.py: zeros, denormals, 1e300, negative numbers, -0.0 in both scales through ngsld_set_geno_raw --
    the device's prep fails with the reader's "NaN found" where the oracle's reader (held to the reference's on the same kind of
    matrices, tests/test_oracle_vs_ref.py) fails, and gives its allele frequencies where it does not."""
    import os
    import subprocess
    import sys
    from ngsld_amd import capi
    r = subprocess.run([sys.executable, os.path.join(capi.REPO_DIR, "tools", "probe_special_values.py"), "150"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and ", 0 differ" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
