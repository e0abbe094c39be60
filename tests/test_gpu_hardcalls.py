"""GPU: larger shapes with hard-called genotypes, rare variants and allele frequencies above one half -- exact zeros in
P and in the haplotype frequencies, hap 0 near 1 or near 0, hundreds of EMs at the iteration cap -- every pair against
the oracle.  (The three-value EM step needs its allele relabelling and its switch to the full step exactly here.)"""
import numpy as np
import pytest

from ngsld_amd import synth
from oracle import orc
from util import MAF_TOL, check_records, close, pearson_tolerance

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case,n_sites,n_ind,mode,ignore", [
    (0, 700, 40, "called", False), (1, 900, 300, "called", False), (2, 500, 600, "called", False),
    (3, 800, 120, "rare", False), (4, 700, 300, "rare", True), (5, 400, 1100, "rare", False),
    (6, 600, 300, "flipped", False), (7, 600, 64, "called_miss", True)])
def test_hard_calls_and_rare_variants(engine, case, n_sites, n_ind, mode, ignore):
    rng = np.random.default_rng(100 + case)
    raw = synth.make_gl_numpy(n_sites, n_ind, 100 + case, depth=8.0)
    if mode.startswith("called"):                       # hard calls: likelihood 1 for the called genotype, 0 elsewhere
        raw = np.eye(3)[raw.argmax(axis=2)]
        if mode == "called_miss":
            raw[rng.random((n_sites, n_ind)) < 0.3] = 1.0 / 3.0
    elif mode == "rare":                                # rare variants: most individuals certain homozygous reference
        q = rng.uniform(0.0, 0.02, size=n_sites)
        g = ((rng.random((n_sites, n_ind)) < q[:, None]).astype(int) +
             (rng.random((n_sites, n_ind)) < q[:, None]).astype(int))
        d = rng.poisson(6.0, size=(n_sites, n_ind))
        k = rng.binomial(d, np.asarray(synth.P_ALT)[g])
        for x, px in enumerate(synth.P_ALT):
            raw[:, :, x] = np.power(px, k) * np.power(1.0 - px, d - k)
    elif mode == "flipped":                             # allele frequencies above one half at every other site
        raw[::2] = raw[::2, :, ::-1]
    o = orc.Oracle(raw, None, ignore_miss_data=ignore, n_threads=32)
    rec = o.run()
    engine.set_geno_raw(raw, ignore_miss_data=ignore)
    engine.set_pos_dist(None)
    assert np.all(close(engine.maf(), o.maf, MAF_TOL))
    n = engine.plan(0, 0, 0.0, ignore, True)
    assert n == len(rec) == n_sites * (n_sites - 1) // 2
    s1, s2, std, ext = engine.run()
    check_records(std, ext, rec, pearson_tol=pearson_tolerance(o.gl, s1, s2))
