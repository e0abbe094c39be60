"""GPU: the run kernels (one wavefront per pair: n_ind 129..640 in the P form, 641..960 in the a/b form; one workgroup per
run of up to 16 items of a row) against the oracle.

What is specific to it and therefore tested here: rows longer than one run (claims that cross item and run
boundaries), items whose mask drops candidates (maf[s2] skip, --rnd_sample), the row's short last item, records
written from the wave-private result rings (every 32 pairs and at the end of the run), site scalars that travel with
the site copy."""
import numpy as np
import pytest

from ngsld_amd import capi, shard, synth
from oracle import orc
from util import MAF_TOL, check_records, close

pytestmark = pytest.mark.gpu


def _check(engine, raw, pd, **kw):
    ignore = kw.get("ignore_miss_data", False)
    o = orc.Oracle(raw, pd, n_threads=16, **kw)
    rec = o.run()
    engine.set_geno_raw(raw, ignore_miss_data=ignore)
    engine.set_pos_dist(pd)
    assert np.all(close(engine.maf(), o.maf, MAF_TOL))
    n = engine.plan(kw.get("max_kb_dist", 0), kw.get("max_snp_dist", 0), kw.get("min_maf", 0.0), ignore, True,
                    kw.get("rnd_sample", 1.0), kw.get("seed", 0))
    assert n == len(rec)
    s1, s2, std, ext = engine.run()
    assert np.array_equal(s1, rec["s1"]) and np.array_equal(s2, rec["s2"])
    check_records(std, ext, rec)
    return rec


def test_rows_longer_than_a_run(engine):
    """All pairs of 700 sites x 300 ind: row 0 has 699 candidates = 11 items = 2 runs; rows shrink to one short item."""
    raw = synth.make_gl_numpy(700, 300, 901, depth=4.0)
    rec = _check(engine, raw, None)
    assert len(rec) == 700 * 699 // 2


def test_filters_inside_runs(engine):
    """maf[s2] skips and a Tausworthe sub-sample punch holes into the items' masks; some rows lose every pair."""
    raw = synth.make_gl_numpy(640, 260, 902, depth=6.0)
    o0 = orc.Oracle(raw, None)
    min_maf = float(np.round(np.nanquantile(o0.maf, 0.25), 3))
    rec = _check(engine, raw, None, min_maf=min_maf, rnd_sample=0.6, seed=12345)
    assert 0 < len(rec) < 640 * 639 // 2


def test_ignore_miss_data_in_runs(engine):
    """--ignore_miss_data (the masked variant of the run kernel): sample_size varies per pair and stays bit-exact."""
    raw = synth.make_gl_numpy(600, 400, 903, depth=5.0)
    miss = np.random.default_rng(903).random((600, 400)) < 0.1
    raw[miss] = 1.0 / 3.0
    rec = _check(engine, raw, None, ignore_miss_data=True)
    assert rec["n_ind_data"].min() < rec["n_ind_data"].max() <= 400


def test_windowed_rows_split_into_equal_runs(engine):
    """A 60 kb window over 3,000 sites x 512 ind (no padding lane): ~600 candidates per row = 10 items = 2 runs of 5."""
    n_sites = 3000
    raw = synth.make_gl_numpy(n_sites, 512, 904, depth=10.0)
    chrs, pos = synth.make_positions(n_sites, 904, max_gap=200, n_chr=2)
    pd = shard.pos_dist_from_positions(chrs, pos)
    rec = _check(engine, raw[:900], pd[:900], max_kb_dist=60)       # oracle-sized head: full comparison
    assert len(rec) > 200_000


@pytest.mark.parametrize("n_ind,ignore_miss", [(513, False), (640, True), (777, False), (896, False), (900, True), (960, False), (1000, False), (1024, True)])
def test_ab_form_kernel_matches_the_oracle(n_ind, ignore_miss):
    """NGSLD_PAIR_KERNEL=ab: one wavefront per pair for 513..1024 individuals, EM step in its a/b form (ld_pair_ab.hip: the
    default for 641..960).  Held to the same bars as every kernel."""
    import os
    from oracle import orc
    from util import check_records
    raw = synth.make_gl_numpy(30, n_ind, 905 + n_ind, depth=3.0)
    raw[5] = [1.0, 0.0, 0.0]
    raw[9, ::3] = 1.0 / 3.0
    want = orc.Oracle(raw, ignore_miss_data=ignore_miss, n_threads=4).run()
    os.environ["NGSLD_PAIR_KERNEL"] = "ab"
    try:
        eng = capi.Engine(0)
    finally:
        del os.environ["NGSLD_PAIR_KERNEL"]
    try:
        eng.set_geno_raw(raw, ignore_miss_data=ignore_miss)
        assert eng.pair_kernel() == "ab"
        eng.set_pos_dist(None)
        assert eng.plan(ignore_miss_data=ignore_miss) == len(want)
        s1, s2, std, ext = eng.run()
        check_records(std, ext, want)
    finally:
        eng.close()


@pytest.mark.parametrize("n_ind,ignore_miss", [(1281, False), (1400, True), (1536, False), (1664, True), (2561, False), (2700, True),
                                               (3072, False), (3328, False), (3300, True), (5121, False), (5200, True), (6000, False),
                                               (6656, True), (6600, False),
                                               # nine / ten per lane, and fourteen / fifteen -- under the flag too, where the default
                                               # keeps to the P form but a matrix set without the flag may still be planned with it
                                               (1100, True), (1153, False), (2100, False), (2100, True), (2400, True), (4300, False),
                                               (4700, True), (1700, False), (1700, True), (1920, True), (3500, False), (3800, True),
                                               (7000, False), (7000, True), (7680, False)])
def test_multi_wavefront_ab_form_kernel_matches_the_oracle(n_ind, ignore_miss):
    """NGSLD_PAIR_KERNEL=abm: two / four / eight wavefronts per pair in the a/b form, 11..13 individuals per lane with the row
    slice in registers (pair_ld_abm_kernel, ld_pair_ab.hip).  Held to the same bars as every kernel."""
    import os
    from oracle import orc
    from util import check_records
    n_sites = 14 if n_ind < 4000 else 7
    raw = synth.make_gl_numpy(n_sites, n_ind, 1905 + n_ind, depth=3.0)
    raw[3] = [1.0, 0.0, 0.0]
    raw[5, ::3] = 1.0 / 3.0
    want = orc.Oracle(raw, ignore_miss_data=ignore_miss, n_threads=4).run()
    os.environ["NGSLD_PAIR_KERNEL"] = "abm"
    try:
        eng = capi.Engine(0)
    finally:
        del os.environ["NGSLD_PAIR_KERNEL"]
    try:
        eng.set_geno_raw(raw, ignore_miss_data=ignore_miss)
        assert eng.pair_kernel() == "multi-ab"
        eng.set_pos_dist(None)
        assert eng.plan(ignore_miss_data=ignore_miss) == len(want)
        s1, s2, std, ext = eng.run()
        check_records(std, ext, want)
    finally:
        eng.close()


@pytest.mark.parametrize("n_ind", [100, 500])
def test_runs_recut_between_text_and_record_runs(n_ind):
    """ngsld_run cuts the run list to its batch size (text batches: shorter runs), ngsld_run_device back to whole rows:
    one context alternating between them keeps producing the same record bits, and its text is the text of a fresh context."""
    import torch
    n_sites = 1500
    raw = synth.make_gl_numpy(n_sites, n_ind, 906 + n_ind, depth=6.0)
    chrs, pos = synth.make_positions(n_sites, 906, max_gap=200)
    pd = shard.pos_dist_from_positions(chrs, pos)
    labels = [f"{c}:{p}" for c, p in zip(chrs, pos)]

    def fresh():
        e = capi.Engine(0)
        e.set_geno_raw(raw)
        e.set_pos_dist(pd)
        n = e.plan(30, 0, 0.0, False, True)
        return e, n

    ref_eng, n = fresh()
    try:
        assert ref_eng.pair_kernel() in ("group", "run") and n > 50_000
        ref_rec = ref_eng.run()
    finally:
        ref_eng.close()
    txt_eng, _ = fresh()
    try:
        txt_eng.set_text_output(labels)
        ref_text, _ = txt_eng.run_text()
    finally:
        txt_eng.close()

    eng, _ = fresh()
    try:
        dev = torch.device("cuda", 0)
        d_std = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
        d_ext = torch.zeros(n * 40, dtype=torch.uint8, device=dev)

        def device_records():
            d_std.zero_(); d_ext.zero_()
            torch.cuda.synchronize()
            eng.run_device(0, n_sites, d_std.data_ptr(), d_ext.data_ptr())
            torch.cuda.synchronize()
            return d_std.cpu().numpy().tobytes(), d_ext.cpu().numpy().tobytes()

        # (ngsld_run also replays the pairs whose PRINTED digits rounding could change, ngsld_run_device only the
        # ill-conditioned ones: the two record sets may differ in last bits there, so each is compared with its own kind)
        want = device_records()                                # whole-row runs
        dev_std = np.frombuffer(want[0], dtype=ref_rec[2].dtype)
        assert np.allclose(dev_std["r2"], ref_rec[2]["r2"], rtol=0, atol=1e-12, equal_nan=True)
        eng.set_tuning(batch_pairs=20_000)                     # small batches: the run list is cut finer
        eng.set_text_output(labels)
        text, _ = eng.run_text()
        assert text == ref_text
        assert device_records() == want                       # ... and back to whole rows
        eng.set_text_output(None, enable=False)
        got = eng.run()                                        # record batches of 20,000 pairs
        for a, b in zip(got, ref_rec):
            assert a.tobytes() == b.tobytes()
        assert device_records() == want
    finally:
        eng.close()


# what pair_config picks by cohort size (profiles/r03/sweep_513_1024.txt), and that every one of those shapes agrees with the oracle
SHAPES = [(512, False, "run"), (513, False, "run"), (513, True, "run"), (576, False, "run"), (576, True, "run"),
          (577, False, "run"), (577, True, "run"), (640, False, "run"), (640, True, "run"), (641, False, "ab"),
          (704, True, "ab"), (768, False, "ab"), (832, False, "ab"), (832, True, "ab"), (833, False, "ab"), (896, True, "ab"),
          (897, False, "ab"), (960, True, "ab"), (961, False, "multi"), (961, True, "multi")]


@pytest.mark.parametrize("n_ind,ignore_miss,family", SHAPES)
def test_cohort_sizes_around_the_kernel_boundaries(n_ind, ignore_miss, family):
    """513..640 individuals stay on ONE wavefront per pair (nine / ten individuals per lane), 641..960 take the a/b form,
    beyond that two wavefronts share a pair: the kernel reported is the one expected, and every one of them meets the
    oracle -- a monomorphic site, a site without data for a third of the cohort, a row longer than one item."""
    n_sites = 70
    raw = synth.make_gl_numpy(n_sites, n_ind, 1300 + n_ind, depth=3.0)
    raw[5] = [1.0, 0.0, 0.0]
    raw[9, ::3] = 1.0 / 3.0
    want = orc.Oracle(raw, ignore_miss_data=ignore_miss, n_threads=8).run()
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw, ignore_miss_data=ignore_miss)
        eng.set_pos_dist(None)
        assert eng.plan(ignore_miss_data=ignore_miss) == len(want)
        assert eng.pair_kernel() == family
        assert capi.describe_dispatch(n_ind, ignore_miss).split()[0] == family   # the table of tests/golden/dispatch_table.txt is what runs
        s1, s2, std, ext = eng.run()
        check_records(std, ext, want)
    finally:
        eng.close()


def test_multi_wavefront_kernel_can_be_forced_from_513_on():
    """NGSLD_PAIR_KERNEL=multi (tests, A/B): the several-wavefronts-per-pair kernel for a cohort the one-wavefront kernels
    would take -- its five-slot shapes with padding inside the last wavefront keep their coverage."""
    import os
    raw = synth.make_gl_numpy(40, 600, 1400, depth=3.0)
    want = orc.Oracle(raw, n_threads=8).run()
    os.environ["NGSLD_PAIR_KERNEL"] = "multi"
    try:
        eng = capi.Engine(0)
    finally:
        del os.environ["NGSLD_PAIR_KERNEL"]
    try:
        eng.set_geno_raw(raw)
        assert eng.pair_kernel() == "multi"
        eng.set_pos_dist(None)
        assert eng.plan() == len(want)
        s1, s2, std, ext = eng.run()
        check_records(std, ext, want)
    finally:
        eng.close()
