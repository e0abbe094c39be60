"""GPU parity: the HIP pair kernel (through the C-ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): |delta| <= 1e-9 on r2 / D / D' (and on hap, r2_ExpG), sample_size and
nIter bit-exact.  Tolerance is absolute, written here as TOL.
"""
import numpy as np
import pytest

from ngsld_amd import synth
from oracle import orc

from util import MAF_TOL, check_records, close

pytestmark = pytest.mark.gpu


def check_against_oracle(engine, raw, pos_dist=None, log_scale=False, ignore_miss=False, max_kb=0, max_snp=0,
                         min_maf=0.0, via_lkl=False, rnd_sample=1.0, seed=0):
    o = orc.Oracle(raw, pos_dist, log_scale=log_scale, ignore_miss_data=ignore_miss, max_kb_dist=max_kb,
                   max_snp_dist=max_snp, min_maf=min_maf, n_threads=4, rnd_sample=rnd_sample, seed=seed)
    rec = o.run()
    if via_lkl:
        engine.set_geno_lkl(o.gl, o.maf)
    else:
        engine.set_geno_raw(raw, log_scale=log_scale, ignore_miss_data=ignore_miss)
    engine.set_pos_dist(pos_dist)
    assert np.all(close(engine.maf(), o.maf, MAF_TOL)), "est_maf differs by more than 1e-12"
    n = engine.plan(max_kb, max_snp, min_maf, ignore_miss, True, rnd_sample, seed)
    assert n == len(rec), f"pair count {n} != oracle {len(rec)}"
    s1, s2, std, ext = engine.run()
    assert np.array_equal(s1, rec["s1"]) and np.array_equal(s2, rec["s2"])
    check_records(std, ext, rec)
    return rec


def _forced(kernel):
    """A context of its own with NGSLD_PAIR_KERNEL=<kernel> (read when the context is created)."""
    import os
    from ngsld_amd import capi
    os.environ["NGSLD_PAIR_KERNEL"] = kernel
    try:
        return capi.Engine(0)
    finally:
        del os.environ["NGSLD_PAIR_KERNEL"]


def multi_ab_shape(n_ind, masked):
    """pair_config's rule (ld_pair_w1.hip): does the cohort run on several wavefronts per pair in the a/b form?"""
    for w in (2, 4, 8):
        slots = -(-n_ind // (64 * w))
        lo, hi = (9, 13 if w == 2 else 14) if masked else ((11 if w == 2 else 9), 15)
        if lo <= slots <= hi:
            return True
    return False


def test_selftest(engine):
    engine.selftest()


@pytest.mark.parametrize("n_sites,n_ind,depth,seed", [
    (100, 24, 2.0, 1),      # C1 shape, slow convergence incl. nIter == 100
    (128, 100, 5.0, 7),     # 16-lane groups, padded last slot
    (64, 500, 10.0, 6),     # headline n_ind, 8 slots
    (48, 64, 10.0, 11),     # exactly one full slot
    (40, 65, 10.0, 12),     # one individual in the second slot
    (24, 512, 10.0, 13),    # 8 full slots
])
def test_all_pairs_single_wave(engine, n_sites, n_ind, depth, seed):
    raw = synth.make_gl_numpy(n_sites, n_ind, seed, depth=depth)
    check_against_oracle(engine, raw)


@pytest.mark.parametrize("n_sites,n_ind,seed", [(32, 1000, 21), (16, 2000, 22), (20, 513, 23), (12, 1030, 24),
                                                (10, 4096, 25)])
def test_all_pairs_multi_wave(engine, n_sites, n_ind, seed):
    raw = synth.make_gl_numpy(n_sites, n_ind, seed, depth=10.0)
    check_against_oracle(engine, raw)


@pytest.mark.parametrize("n_sites,n_ind,seed,ignore_miss", [(12, 1100, 26, False), (12, 1152, 27, True), (10, 2200, 28, False),
                                                            (9, 2304, 29, True), (8, 4300, 30, False), (7, 4608, 31, True),
                                                            (12, 1153, 32, False), (12, 1200, 33, True), (11, 1280, 34, False),
                                                            (9, 2305, 35, True), (9, 2560, 36, False), (7, 4609, 37, False),
                                                            (6, 4700, 38, True), (6, 4700, 41, False), (6, 5000, 39, False),
                                                            (6, 5000, 42, True), (6, 5120, 40, True)])
def test_nine_and_ten_slots_per_lane(engine, n_sites, n_ind, seed, ignore_miss):
    """Just past a doubling of the wavefronts per pair the kernels hold NINE or TEN individuals per lane on half as many
    wavefronts (2 / 4 / 8 x 9 / 10 x 64; 2 x 10 under --ignore_miss_data runs as 4 x 5 on the same planes): with and without
    --ignore_miss_data, empty slots inside and at the end of the last wavefront.  4,609..5,120 individuals used to run on the
    streaming kernel."""
    raw = synth.make_gl_numpy(n_sites, n_ind, seed, depth=4.0)
    if ignore_miss:
        miss = np.random.default_rng(seed).random((n_sites, n_ind)) < 0.15
        raw[miss] = 1.0 / 3.0
    engine.set_geno_raw(raw[:2], ignore_miss_data=ignore_miss)
    assert engine.pair_kernel() == ("multi-ab" if multi_ab_shape(n_ind, ignore_miss) else "multi")
    check_against_oracle(engine, raw, ignore_miss=ignore_miss)
    eng = _forced("multi")      # the P form on these shapes (the default wherever the a/b form has none or measured behind)
    try:
        eng.set_geno_raw(raw[:2], ignore_miss_data=ignore_miss)
        assert eng.pair_kernel() == "multi"
        check_against_oracle(eng, raw, ignore_miss=ignore_miss)
    finally:
        eng.close()


@pytest.mark.parametrize("n_sites,n_ind,seed,ignore_miss", [(12, 1300, 43, False), (10, 1536, 44, True), (10, 1400, 47, False),
                                                            (8, 2600, 45, False), (8, 3072, 46, True), (8, 1290, 48, True)])
def test_five_and_six_slots_keep_the_row_slice_in_registers(n_sites, n_ind, seed, ignore_miss):
    """Several wavefronts per pair with five or six individuals per lane (1,281..1,536 on four wavefronts, 2,561..3,072 on
    eight): the wavefront's slice of the row vector is loaded once per item and held in registers for its 64 candidates --
    including the relabelling of a row site with maf > 1/2, a monomorphic site and a site without data for some."""
    raw = synth.make_gl_numpy(n_sites, n_ind, seed, depth=4.0)
    raw[1] = raw[1][:, ::-1]                       # maf > 1/2 at a row site: planes 0 and 2 trade places (Relabel)
    raw[3] = [1.0, 0.0, 0.0]
    if ignore_miss:
        miss = np.random.default_rng(seed).random((n_sites, n_ind)) < 0.15
        raw[miss] = 1.0 / 3.0
    eng = _forced("multi")      # (these cohort sizes run in the a/b form by default: test_gpu_run_kernel.py and the fuzz cover it)
    try:
        eng.set_geno_raw(raw[:2], ignore_miss_data=ignore_miss)
        assert eng.pair_kernel() == "multi"
        check_against_oracle(eng, raw, ignore_miss=ignore_miss)
    finally:
        eng.close()


def test_windowed_and_snp_dist(engine):
    raw = synth.make_gl_numpy(300, 50, 31, depth=8.0)
    chrs, pos = synth.make_positions(300, 31, max_gap=200, n_chr=2)
    pd = np.empty(300)
    pd[0] = pos[0]
    for s in range(1, 300):
        pd[s] = np.inf if chrs[s] != chrs[s - 1] else pos[s] - pos[s - 1]
    check_against_oracle(engine, raw, pd, max_kb=2)
    check_against_oracle(engine, raw, pd, max_kb=0, max_snp=7)
    check_against_oracle(engine, raw, pd, max_kb=5, max_snp=20)
    check_against_oracle(engine, raw, pd, max_kb=0)  # all pairs across the chromosome break (dist inf)


def test_min_maf_break_and_skip(engine):
    raw = synth.make_gl_numpy(120, 40, 41, depth=6.0)
    o = orc.Oracle(raw)
    thr = float(np.quantile(o.maf, 0.3))
    rec = check_against_oracle(engine, raw, min_maf=thr)
    assert 0 < len(rec) < 120 * 119 // 2


def test_ignore_miss_data(engine):
    rng = np.random.default_rng(51)
    raw = synth.make_gl_numpy(60, 90, 51, depth=4.0)
    miss = rng.random((60, 90)) < 0.15
    raw[miss] = 1.0 / 3.0                    # all-equal triple = missing (gen_func.cpp:862-868)
    raw[5, :, :] = 0.25                      # a site missing for everybody
    for ignore in (False, True):
        rec = check_against_oracle(engine, raw, ignore_miss=ignore)
        if ignore:
            assert rec["n_ind_data"].min() == 0 and rec["n_ind_data"].max() < 90


def test_degenerate_sites(engine):
    """monomorphic / hard-called sites: exact zeros in the GLs, NaN and inf results."""
    n_ind = 12
    raw = synth.make_gl_numpy(10, n_ind, 61, depth=3.0)
    raw[2] = np.array([1.0, 0.0, 0.0])                       # monomorphic ref, hard calls
    raw[3] = np.array([0.0, 0.0, 1.0])                       # monomorphic alt
    g = np.random.default_rng(61).integers(0, 3, size=n_ind)  # hard-called polymorphic site
    raw[4] = np.eye(3)[g]
    raw[6] = raw[4]                                          # perfect LD with site 4
    for ignore in (False, True):
        check_against_oracle(engine, raw, ignore_miss=ignore)


def test_log_scale_input(engine):
    raw = synth.make_gl_numpy(50, 30, 71, depth=5.0)
    with np.errstate(divide="ignore"):
        lg = np.log(raw)
    check_against_oracle(engine, lg, log_scale=True)


def test_reference_contract_entry(engine):
    """ngsld_set_geno_lkl: the caller hands over geno_lkl + maf exactly as `params` holds them."""
    raw = synth.make_gl_numpy(70, 200, 81, depth=10.0)
    check_against_oracle(engine, raw, via_lkl=True)


def test_batched_run_matches_single_batch(engine):
    raw = synth.make_gl_numpy(200, 64, 91, depth=10.0)
    engine.set_geno_raw(raw)
    engine.set_pos_dist(None)
    engine.set_tuning(pairs_per_item=16, batch_pairs=1 << 23)
    engine.plan(extend_out=True)
    a = engine.run()
    engine.set_tuning(pairs_per_item=5, batch_pairs=1000)   # many small batches, odd item size
    engine.plan(extend_out=True)
    b = engine.run()
    engine.set_tuning(pairs_per_item=16, batch_pairs=1 << 23)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)     # bit-identical: the per-pair reduction order is fixed


@pytest.mark.parametrize("rnd,seed", [(0.5, 1), (0.1, 99), (0.9, 4294967296 + 5), (1.0, 3)])
def test_rnd_sample_streams(engine, rnd, seed):
    """--rnd_sample / --seed: the same pairs survive as in the oracle's per-row Tausworthe streams."""
    raw = synth.make_gl_numpy(150, 30, 33, depth=5.0)
    chrs, pos = synth.make_positions(150, 33, n_chr=2)
    pd = np.empty(150)
    pd[0] = pos[0]
    for s in range(1, 150):
        pd[s] = np.inf if chrs[s] != chrs[s - 1] else pos[s] - pos[s - 1]
    rec = check_against_oracle(engine, raw, pd, max_kb=8, rnd_sample=rnd, seed=seed)
    full = check_against_oracle(engine, raw, pd, max_kb=8)
    assert (len(rec) == len(full)) == (rnd == 1.0)


@pytest.mark.parametrize("n_sites,n_ind,seed", [
    (2, 1, 201), (3, 2, 202), (1, 40, 203),            # degenerate sizes (one site: no pair at all)
    (40, 7, 210), (40, 8, 211), (40, 9, 212),          # 8-lane groups: partial / full first slot, one lane into the second
    (30, 16, 204), (30, 17, 205), (30, 63, 213),       # 8-lane groups, up to 8 slots
    (30, 64, 214), (30, 65, 215),                      # last shape of the 8-lane groups / first of the 16-lane groups
    (20, 128, 206), (20, 129, 207),                    # last shape of the 16-lane groups / first of the 32-lane groups
    (40, 130, 208), (24, 160, 216), (24, 161, 217),    # 32-lane groups (5 slots) / wavefront kernel (161..192)
    (24, 193, 218), (24, 224, 219), (24, 225, 220),    # 32-lane groups (7 slots) / wavefront kernel again
    (16, 448, 209),                                    # wavefront kernel, partial last slot
])
def test_kernel_family_boundaries(engine, n_sites, n_ind, seed):
    raw = synth.make_gl_numpy(n_sites, n_ind, seed, depth=6.0)
    rec = check_against_oracle(engine, raw)
    assert len(rec) == n_sites * (n_sites - 1) // 2


def test_long_rows_and_many_items(engine):
    """All pairs of 3000 sites x 20 ind: rows of up to 2999 candidates = 47 items each, claimed dynamically."""
    raw = synth.make_gl_numpy(3000, 20, 211, depth=3.0)
    engine.set_geno_raw(raw)
    engine.set_pos_dist(None)
    n = engine.plan(extend_out=True)
    assert n == 3000 * 2999 // 2
    s1, s2, std, ext = engine.run()
    o = orc.Oracle(raw[[5, 2990]], None)           # spot-check two far-apart sites against the oracle
    r = o.run()[0]
    k = int(np.flatnonzero((s1 == 5) & (s2 == 2990))[0])
    assert ext["n_iter"][k] == r["n_iter"] and np.all(close(ext["hap"][k], r["hap"]))
    assert np.all(close([std["D"][k], std["r2"][k], std["r2_ExpG"][k]], [r["D"], r["r2"], r["r2pear"]]))
    assert float(np.abs(ext["hap"].sum(axis=1) - 1).max()) < 1e-12


@pytest.mark.parametrize("n_sites,n_ind,seed,ignore", [(6, 5121, 301, False), (5, 6000, 302, True), (5, 6000, 304, False),
                                                       (4, 9001, 303, False), (4, 10000, 305, True), (4, 10000, 306, False),
                                                       (5, 5633, 307, True), (4, 7000, 308, False), (4, 7681, 309, True),
                                                       (4, 8200, 310, False), (3, 10240, 311, False), (3, 10241, 312, False),
                                                       (3, 10300, 313, True), (3, 12000, 314, False), (3, 9217, 315, True),
                                                       (3, 9800, 316, False), (3, 20000, 317, False), (3, 33000, 318, True)])
def test_streaming_kernel_large_cohorts(engine, n_sites, n_ind, seed, ignore):
    """n_ind > 5120: the streaming kernel -- up to 10,240 individuals the candidate's vector stays in registers and every EM
    iteration re-reads the row vector only (11 .. 20 blocks of 64 individuals per wavefront: 5,121 / 5,633 / ... / 10,240 sit
    on block-count borders); beyond that 20 blocks per wavefront (10,240 individuals) stay resident and the rest of both
    vectors is re-read."""
    raw = synth.make_gl_numpy(n_sites, n_ind, seed, depth=8.0)
    if ignore:
        raw[np.random.default_rng(seed).random((n_sites, n_ind)) < 0.05] = 1.0 / 3.0
    eng = _forced("bres")       # (up to 7,680 individuals the default is eight wavefronts per pair in the a/b form)
    try:
        eng.set_geno_raw(raw[:2], ignore_miss_data=ignore)
        assert eng.pair_kernel() == "stream"
        check_against_oracle(eng, raw, ignore_miss=ignore)
    finally:
        eng.close()
    engine.set_geno_raw(raw[:2], ignore_miss_data=ignore)
    assert engine.pair_kernel() == ("multi-ab" if multi_ab_shape(n_ind, ignore) else "stream")
    check_against_oracle(engine, raw, ignore_miss=ignore)


@pytest.mark.parametrize("n_sites,n_ind,seed,ignore", [(5, 5121, 331, False), (4, 6000, 332, True), (3, 10241, 333, False),
                                                       (3, 16000, 334, True)])
def test_plain_streaming_kernel(n_sites, n_ind, seed, ignore):
    """NGSLD_PAIR_KERNEL=stream: nothing resident, both site vectors re-read in every EM iteration (the kernel the resident
    form is measured against; held to the same bars)."""
    import os
    from ngsld_amd import capi
    raw = synth.make_gl_numpy(n_sites, n_ind, seed, depth=8.0)
    if ignore:
        raw[np.random.default_rng(seed).random((n_sites, n_ind)) < 0.05] = 1.0 / 3.0
    os.environ["NGSLD_PAIR_KERNEL"] = "stream"
    try:
        eng = capi.Engine(0)
    finally:
        del os.environ["NGSLD_PAIR_KERNEL"]
    try:
        check_against_oracle(eng, raw, ignore_miss=ignore)
        assert eng.pair_kernel() == "stream"
    finally:
        eng.close()


def test_a_matrix_set_without_the_flag_can_be_planned_with_it(engine):
    """ngsld_set_geno_lkl carries no --ignore_miss_data: the plane layout (and with it the kernel family) follows "every individual
    counts", and ngsld_plan may still ask for the flag.  1,700 individuals: two wavefronts x 14 per lane in the a/b form either way
    (set WITH the flag the same cohort takes four wavefronts x 7 in the P form)."""
    n_sites, n_ind = 10, 1700
    raw = synth.make_gl_numpy(n_sites, n_ind, 341, depth=3.0)
    raw[np.random.default_rng(341).random((n_sites, n_ind)) < 0.1] = 1.0 / 3.0
    o = orc.Oracle(raw, None, ignore_miss_data=True, n_threads=4)
    rec = o.run()
    engine.set_geno_lkl(o.gl, o.maf)
    engine.set_pos_dist(None)
    assert engine.plan(0, 0, 0.0, True, True, 1.0, 0) == len(rec)
    assert engine.pair_kernel() == "multi-ab"
    s1, s2, std, ext = engine.run()
    check_records(std, ext, rec)
    engine.set_geno_raw(raw, ignore_miss_data=True)
    assert engine.pair_kernel() == "multi"


def test_kernels_with_vanishing_weights_take_a_second_opinion(engine):
    """Individuals 0, 64, 128, 192 -- lane 0's first four slots of wavefront 0 at 6,000 individuals on eight wavefronts in the a/b
    form, where eight slots share ONE reciprocal -- are certain alt/alt homozygotes at every site while the caller's maf says
    1e-45: their s is f3^2 = 1e-180 in the first iteration.  Each has a reciprocal; their PRODUCT has none: the step has to be
    redone with one reciprocal per individual instead of ending as the reference's all-NaN step.  (The streaming kernel, which
    takes one reciprocal per individual throughout, is held to the same records.)"""
    n_sites, n_ind = 4, 6000
    raw = synth.make_gl_numpy(n_sites, n_ind, 321, depth=8.0)
    raw[:, [0, 64, 128, 192, 512, 1024, 1536], :] = [0.0, 0.0, 1.0]
    o = orc.Oracle(raw, None, n_threads=4)
    o.maf[:] = 1e-45
    rec = o.run()
    assert np.isfinite(rec["hap"]).all() and (rec["n_iter"] > 1).all()
    bres = _forced("bres")
    try:
        for eng, name in ((bres, "stream"), (engine, "multi-ab")):   # (the default at 6,000 individuals shares reciprocals: its second opinion)
            eng.set_geno_lkl(o.gl, o.maf)
            assert eng.pair_kernel() == name
            eng.set_pos_dist(None)
            assert eng.plan(0, 0, 0.0, False, True, 1.0, 0) == len(rec)
            s1, s2, std, ext = eng.run()
            assert np.array_equal(s1, rec["s1"]) and np.array_equal(s2, rec["s2"])
            check_records(std, ext, rec)
    finally:
        bres.close()


def test_api_error_paths(engine):
    from ngsld_amd import capi
    raw = synth.make_gl_numpy(8, 10, 401, depth=4.0)
    fresh = capi.Engine(0)
    try:
        with pytest.raises(capi.NgsldError) as e:
            fresh.plan()
        assert e.value.code == capi.ERR_INVALID
        bad = raw.copy()
        bad[2, 3, :] = -1.0
        with pytest.raises(capi.NgsldError) as e:
            fresh.set_geno_raw(bad)
        assert e.value.code == capi.ERR_NAN and "NaN found" in e.value.msg
        fresh.set_geno_raw(raw)
        with pytest.raises(capi.NgsldError) as e:
            fresh.plan(max_kb_dist=5)                       # distance filter without positions (parse_args.cpp:174)
        assert "position file necessary" in e.value.msg
        with pytest.raises(capi.NgsldError) as e:
            fresh.plan(min_maf=1.5)
        assert "minimum allele frequency" in e.value.msg
        with pytest.raises(capi.NgsldError) as e:
            fresh.set_geno_raw(raw, call_geno=(0.9, 0.1))   # gen_func.cpp:887
        assert "missing data threshold" in e.value.msg
        fresh.set_geno_raw(raw)
        fresh.set_pos_dist(None)
        assert fresh.plan() == 28
        with pytest.raises(capi.NgsldError) as e:
            fresh.run(0, 99)
        assert e.value.code == capi.ERR_INVALID
    finally:
        fresh.close()


def test_chunked_host_ingestion(monkeypatch):
    """Host matrices enter the device in chunks of sites through two staging buffers; forced here to 3 sites per
    chunk (NGSLD_TEST_STAGE_BYTES) so that 50 sites take 17 chunks -- results must equal the one-chunk run bit for bit."""
    from ngsld_amd import capi
    raw = synth.make_gl_numpy(50, 70, 501, depth=5.0)
    outs = []
    for stage in (None, str(3 * 70 * 24), "1"):
        if stage is None:
            monkeypatch.delenv("NGSLD_TEST_STAGE_BYTES", raising=False)
        else:
            monkeypatch.setenv("NGSLD_TEST_STAGE_BYTES", stage)
        eng = capi.Engine(0)
        try:
            eng.set_geno_raw(raw)
            maf = eng.maf()
            eng.set_pos_dist(None)
            eng.plan(extend_out=True)
            outs.append((maf,) + eng.run())
        finally:
            eng.close()
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert np.array_equal(x, y)
    o = orc.Oracle(raw)
    assert np.all(close(outs[0][0], o.maf, MAF_TOL))
