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
    # called genotypes (with or without missing data) take the genotype-combination kernel, everything else the
    # per-individual kernels
    assert (engine.pair_kernel() == "hard") == mode.startswith("called")
    assert np.all(close(engine.maf(), o.maf, MAF_TOL))
    n = engine.plan(0, 0, 0.0, ignore, True)
    assert n == len(rec) == n_sites * (n_sites - 1) // 2
    s1, s2, std, ext = engine.run()
    check_records(std, ext, rec, pearson_tol=pearson_tolerance(o.gl, s1, s2))


@pytest.mark.parametrize("case,n_sites,n_ind,miss,ignore", [
    (0, 500, 24, 0.0, False), (1, 400, 500, 0.2, False), (2, 400, 500, 0.2, True), (3, 300, 2000, 0.05, True),
    (4, 200, 5000, 0.1, False), (5, 300, 64, 0.5, True), (6, 300, 129, 0.0, False)])
def test_called_genotypes_on_both_kernel_paths(case, n_sites, n_ind, miss, ignore, monkeypatch):
    """Hard-called matrices against the oracle twice: on the genotype-combination kernel (ld_pair_hard.hip) and, with
    NGSLD_TEST_HARD_KERNEL=0, on the per-individual kernels; the two paths agree with each other far inside the tolerance."""
    from ngsld_amd import capi
    rng = np.random.default_rng(700 + case)
    raw = np.eye(3)[synth.make_gl_numpy(n_sites, n_ind, 700 + case, depth=4.0).argmax(axis=2)]
    if miss:
        raw[rng.random((n_sites, n_ind)) < miss] = 1.0 / 3.0
        raw[5, :] = 1.0 / 3.0                                    # a site nobody has data for
    raw[7] = np.eye(3)[0]                                        # a monomorphic site
    chrs, pos = synth.make_positions(n_sites, 700 + case, n_chr=2)
    from ngsld_amd import shard
    pd = shard.pos_dist_from_positions(chrs, pos)
    o = orc.Oracle(raw, pd, ignore_miss_data=ignore, max_kb_dist=8, min_maf=0.02, n_threads=32)
    rec = o.run()
    got = {}
    for path in ("hard", "generic"):
        if path == "generic":
            monkeypatch.setenv("NGSLD_TEST_HARD_KERNEL", "0")
        eng = capi.Engine(0)
        try:
            eng.set_geno_raw(raw, ignore_miss_data=ignore)
            assert (eng.pair_kernel() == "hard") == (path == "hard")
            eng.set_pos_dist(pd)
            assert np.all(close(eng.maf(), o.maf, MAF_TOL))
            n = eng.plan(8, 0, 0.02, ignore, True)
            assert n == len(rec)
            s1, s2, std, ext = eng.run()
        finally:
            eng.close()
        assert np.array_equal(s1, rec["s1"]) and np.array_equal(s2, rec["s2"])
        check_records(std, ext, rec, pearson_tol=pearson_tolerance(o.gl, s1, s2))
        got[path] = (std, ext)
    assert np.array_equal(got["hard"][1]["n_iter"], got["generic"][1]["n_iter"])
    assert np.all(close(got["hard"][1]["hap"], got["generic"][1]["hap"], 1e-12))


@pytest.mark.parametrize("case,n_sites,n_ind,miss,ignore", [
    (0, 400, 100, 0.0, False), (1, 400, 100, 0.1, False), (2, 400, 100, 0.1, True), (3, 300, 37, 0.3, False),
    (4, 200, 2000, 0.2, True), (5, 300, 10, 0.5, False)])
def test_maf_is_bit_identical_on_called_genotypes(engine, case, n_sites, n_ind, miss, ignore):
    """est_maf of called genotypes (with missing data) equals the oracle's BIT FOR BIT -- its terms are 0, 1, 2, the sums are
    exact whatever their order -- so a round --min_maf (0.05 with 100 individuals: a frequency many sites have exactly)
    keeps and drops the same sites as the reference (deviations)."""
    rng = np.random.default_rng(case)
    raw = np.eye(3)[synth.make_gl_numpy(n_sites, n_ind, 50 + case, depth=3.0).argmax(axis=2)]
    raw[rng.random((n_sites, n_ind)) < miss] = 1.0 / 3.0
    o = orc.Oracle(raw, None, ignore_miss_data=ignore)
    engine.set_geno_raw(raw, ignore_miss_data=ignore)
    m = engine.maf()
    assert np.all((m == o.maf) | (np.isnan(m) & np.isnan(o.maf)))
    if not ignore:                                       # a threshold that IS the frequency of some sites resolves alike
        thr = float(np.sort(o.maf[np.isfinite(o.maf)])[n_sites // 3])
        assert np.count_nonzero(o.maf == thr) >= 1
        o2 = orc.Oracle(raw, None, min_maf=thr)
        engine.set_pos_dist(None)
        assert engine.plan(0, 0, thr, False, True) == o2.count()


@pytest.mark.parametrize("n_ind,call,ignore,miss", [(500, None, False, 0.0), (1000, None, False, 0.0), (500, (0.0, 0.0), False, 0.0),
                                                  (500, (0.4, 0.4), False, 0.02), (500, (0.4, 0.4), True, 0.02),
                                                  (2000, None, True, 0.0), (500, None, False, 0.02), (24, None, False, 0.0)])
def test_device_replay_of_called_genotypes_is_the_hosts(n_ind, call, ignore, miss, monkeypatch):
    """Called genotypes make exact ties of eps with EPSILON common (D moves in steps of 1 / (4 n^2)): thousands of flagged
    pairs.  They are replayed ON THE DEVICE (ld_replay.hip: the reference's sequential order, one lane per pair) where the
    values are the host's bits -- and must come out as the host's replay does, bit for bit in hap / D / D' / r2 / nIter /
    sample_size (r2_ExpG stays the pair kernel's), and both as the oracle.  With --call_geno the "no data" individuals are
    part of it (the host's constants for their triple); a matrix that arrives called with missing data (miss > 0, no
    --call_geno) leaves the pairs of such sites to the host."""
    from ngsld_amd import capi
    n_sites = 600 if n_ind <= 1000 else 300
    raw = np.eye(3)[synth.make_gl_numpy(n_sites, n_ind, 4200 + n_ind, depth=8.0).argmax(axis=2)]
    if miss:  # (under --call_geno an all-equal triple becomes call_geno's own "no data" triple, gen_func.cpp:897-905)
        raw[np.random.default_rng(5).random((n_sites, n_ind)) < miss] = 1.0 / 3.0
    raw[7] = np.eye(3)[0]                                # a monomorphic site and a nearly monomorphic one: D' and r2 of their
    raw[19] = np.eye(3)[0]                               # pairs are 0/0-type quotients (-nan, inf): replayed, bit for bit
    raw[19, 3] = np.eye(3)[1]
    o = orc.Oracle(raw, None, ignore_miss_data=ignore, n_threads=32, call_geno=call)
    rec = o.run()
    got = {}
    for where in ("device", "host"):
        monkeypatch.setenv("NGSLD_REPLAY_DEVICE", "1" if where == "device" else "0")
        eng = capi.Engine(0)
        try:
            eng.set_geno_raw(raw, ignore_miss_data=ignore, call_geno=call)
            assert eng.pair_kernel() == "hard"
            eng.set_pos_dist(None)
            assert eng.plan(0, 0, 0.0, ignore, True) == len(rec)
            s1, s2, std, ext = eng.run()
            replayed = eng.replay_stats()[0]
        finally:
            eng.close()
        check_records(std, ext, rec)
        got[where] = (std, ext, replayed)
    (sd, ed, rd), (sh, eh, rh) = got["device"], got["host"]
    assert rd == rh and (rd > 0 or n_ind < 100), (rd, rh)           # the same pairs were flagged and replayed
    for col in ("D", "Dp", "r2"):
        assert np.array_equal(sd[col].view(np.uint64), sh[col].view(np.uint64)), col
    assert np.array_equal(ed["hap"].view(np.uint64), eh["hap"].view(np.uint64))
    assert np.array_equal(ed["n_iter"], eh["n_iter"]) and np.array_equal(ed["n_ind_data"], eh["n_ind_data"])
    assert np.all(close(sd["r2_ExpG"], sh["r2_ExpG"], 1e-12))
    ties = int(np.sum(rec["n_iter"] <= 2))
    print(f"\n[device replay] n_ind {n_ind} call {call} ignore {ignore} miss {miss}: {len(rec)} pairs, {rd} replayed, {ties} with nIter <= 2")


@pytest.mark.parametrize("n_ind,call,ignore,miss,tiny_list", [(200, None, False, 0.0, False), (500, (0.4, 0.4), False, 0.03, False),
                                                            (500, (0.4, 0.4), True, 0.03, False), (500, None, False, 0.02, False),
                                                            (200, None, False, 0.0, True), (500, (0.4, 0.4), False, 0.03, True)])
def test_a_launch_that_overflows_its_flag_list_is_replayed_on_the_device_too(n_ind, call, ignore, miss, tiny_list, monkeypatch):
    """A called-genotype matrix with MONOMORPHIC sites (a VCF that was never SNP-filtered; the reference's README.md:73) flags
    every pair of such a site: more than the launch's list holds.  Rounds 2-4 left ALL flagged pairs of such a launch to the host's
    threads; now the bitmap is turned into a list of located pairs on the device and a second kernel replays those
    (ld_replay.hip: replay_hard_list_kernel).  Device against host replay, bit for bit; both against the oracle.  A matrix that
    arrives called with missing data and no --call_geno (last case) still leaves the pairs of such sites to the host."""
    from ngsld_amd import capi
    n_sites = 400
    rng = np.random.default_rng(77 + n_ind)
    raw = np.eye(3)[synth.make_gl_numpy(n_sites, n_ind, 5200 + n_ind, depth=8.0).argmax(axis=2)]
    mono = rng.random(n_sites) < 0.3
    raw[mono] = np.eye(3)[0]
    if miss:
        raw[rng.random((n_sites, n_ind)) < miss] = 1.0 / 3.0
    rec = orc.Oracle(raw, None, ignore_miss_data=ignore, n_threads=32, call_geno=call).run()
    got = {}
    if tiny_list:
        # the list of located pairs overflows too (in earnest only a launch of more than 2^26 flagged pairs does: ngsld_run_device on
        # configs[3]'s size): what it cannot hold stays in the bitmap and must reach the host -- it used to be dropped, silently
        monkeypatch.setenv("NGSLD_TEST_REPLAY_LIST_CAP", "3000")
    for where in ("device", "host"):
        monkeypatch.setenv("NGSLD_REPLAY_DEVICE", "1" if where == "device" else "0")
        eng = capi.Engine(0)
        try:
            eng.set_geno_raw(raw, ignore_miss_data=ignore, call_geno=call)
            assert eng.pair_kernel() == "hard"
            eng.set_pos_dist(None)
            assert eng.plan(0, 0, 0.0, ignore, True) == len(rec)
            s1, s2, std, ext = eng.run()
            info = eng.replay_info()
        finally:
            eng.close()
        check_records(std, ext, rec)
        got[where] = (std, ext, info)
    (sd, ed, idv), (sh, eh, ih) = got["device"], got["host"]
    assert idv["pairs_flagged"] == ih["pairs_flagged"] > max(4096, len(rec) // 256)          # (the list did overflow)
    assert ih["pairs_on_device"] == 0
    if tiny_list:
        assert 0 < idv["pairs_on_device"] <= 3000 and idv["pairs_on_host"] == idv["pairs_flagged"] - idv["pairs_on_device"], idv
    elif call is not None or not miss:
        assert idv["pairs_on_device"] > idv["pairs_flagged"] * 0.9, idv
    for col in ("D", "Dp", "r2"):
        assert np.array_equal(sd[col].view(np.uint64), sh[col].view(np.uint64)), col
    assert np.array_equal(ed["hap"].view(np.uint64), eh["hap"].view(np.uint64))
    assert np.array_equal(ed["n_iter"], eh["n_iter"]) and np.array_equal(ed["n_ind_data"], eh["n_ind_data"])
    assert np.all(close(sd["r2_ExpG"], sh["r2_ExpG"], 1e-12))


@pytest.mark.parametrize("n_ind,ignore,mono", [(500, False, 0.3), (500, True, 0.3), (130, True, 0.3), (1000, False, 0.2)])
def test_text_genotypes_with_missing_calls_are_replayed_on_the_device(n_ind, ignore, mono, monkeypatch):
    """A text genotype file ({-1, 0, 1, 2}, no --call_geno): read_geno stores log(1) for the called genotype over -INF and
    log(1/3) three times for a missing call, then post_prob (read_data.cpp:88-98).  The prep pass sees that every individual without
    data is that very triple (PrepArgs::odd_missing), so the device-side replay knows its likelihood and est_maf posterior -- two
    constants from the host's libm (replay_missing_constants_text) -- and takes the pairs of sites with missing calls too (up to
    round 5 they were the host's: 98,000 pairs of a 30,000-site matrix, 3.5x on the pass).  Device against host replay bit for
    bit, both against the oracle; a matrix whose missing triple is anything else (1e-3 off) stays with the host."""
    from ngsld_amd import capi
    n_sites = 400
    rng = np.random.default_rng(900 + n_ind)
    g = synth.make_gl_numpy(n_sites, n_ind, 6100 + n_ind, depth=8.0).argmax(axis=2)
    raw = np.full((n_sites, n_ind, 3), -1e15)
    np.put_along_axis(raw, g[..., None], 0.0, axis=2)
    if mono:
        raw[rng.random(n_sites) < mono] = np.array([0.0, -1e15, -1e15])
    missing = rng.random((n_sites, n_ind)) < 0.04
    raw[missing] = capi.missing_call_log()                         # (log(1/3) as the reader stores it)
    rec = orc.Oracle(raw, None, log_scale=True, ignore_miss_data=ignore, n_threads=32).run()
    got = {}
    for where in ("device", "host", "odd"):
        monkeypatch.setenv("NGSLD_REPLAY_DEVICE", "0" if where == "host" else "1")
        eng = capi.Engine(0)
        try:
            m = raw
            if where == "odd":
                m = raw.copy()
                m[missing] = capi.missing_call_log() + 1e-3             # still "no data" to every kernel, but not the reader's triple
            eng.set_geno_raw(m, log_scale=True, ignore_miss_data=ignore, text=True)
            assert eng.pair_kernel() == "hard"
            eng.set_pos_dist(None)
            assert eng.plan(0, 0, 0.0, ignore, True) == len(rec)
            s1, s2, std, ext = eng.run()
            info = eng.replay_info()
        finally:
            eng.close()
        if where != "odd":
            check_records(std, ext, rec)
        got[where] = (std, ext, info)
    (sd, ed, idv), (sh, eh, ih), (_, _, io) = got["device"], got["host"], got["odd"]
    assert idv["pairs_flagged"] == ih["pairs_flagged"] > 100 and ih["pairs_on_device"] == 0
    assert idv["pairs_on_device"] > idv["pairs_flagged"] * 0.9, idv
    assert io["pairs_on_host"] > io["pairs_flagged"] * 0.5, io       # (sites with missing calls: nearly all of them)
    for col in ("D", "Dp", "r2"):
        assert np.array_equal(sd[col].view(np.uint64), sh[col].view(np.uint64)), col
    assert np.array_equal(ed["hap"].view(np.uint64), eh["hap"].view(np.uint64))
    assert np.array_equal(ed["n_iter"], eh["n_iter"]) and np.array_equal(ed["n_ind_data"], eh["n_ind_data"])
    assert np.all(close(sd["r2_ExpG"], sh["r2_ExpG"], 1e-12))
    print(f"\n[text genotypes, missing calls] n_ind {n_ind} ignore {ignore} mono {mono}: {len(rec)} pairs, {idv['pairs_flagged']} flagged, "
          f"{idv['pairs_on_device']} on the device ({io['pairs_on_device']} when the missing triple is not the reader's)")
