"""GPU: the device-side exact-order replay of LIKELIHOOD matrices (ld_replay_lkl.hip, include/ngsld.h: ngsld_set_exact_store).

Matrices that are not SNP-called (the reference's README.md:73) flag a third of their pairs: every pair with a (nearly)
monomorphic site.  Those are replayed on the device, a wavefront per pair, in the reference's operation order, on an exact
store built through the host's libm.  Held here: the device replay's records are the HOST replay's records bit for bit
(hap, D, D', r2, nIter, sample_size of every pair of the run, flagged or not), on every wavefront
shape of the kernel, with and without --ignore_miss_data, through every path a record can take; and both are the oracle's
bits on the degenerate pairs."""
import hashlib

import numpy as np
import pytest

from ngsld_amd import capi, synth
from oracle import orc
from util import check_records, close, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _every_pairs_em_in_the_pair_kernel(request, monkeypatch):
    # (the tests of this file that create their contexts inside the library -- streamed slabs, parts of one process -- compare
    # host-replay and device-replay runs bit for bit too: see `eng`)
    if "eng_skip" not in request.fixturenames:
        monkeypatch.setenv("NGSLD_REPLAY_SKIP", "0")


@pytest.fixture()
def eng(monkeypatch):
    # Device replay against HOST replay, every record of a run bit for bit -- un-flagged pairs included: the pair kernels run
    # the EM of every pair here.  (By default the pairs of a degenerate site skip theirs and take the replay's value, flagged
    # or not -- the reference's bits, where the host-replay run keeps the kernel's for pairs it does not flag: the same number
    # to 1e-9, not the same bits.  test_degenerate_sites_* below hold that path.)  Read at ngsld_create.
    monkeypatch.setenv("NGSLD_REPLAY_SKIP", "0")
    e = capi.Engine(0)
    yield e
    e.close()


@pytest.fixture()
def eng_skip(monkeypatch):
    monkeypatch.delenv("NGSLD_REPLAY_SKIP", raising=False)
    e = capi.Engine(0)
    yield e
    e.close()


def uncalled(n_sites, n_ind, seed, depth=6.0, missing=False, **kw):
    raw = synth.make_gl_numpy(n_sites, n_ind, seed=seed, depth=depth, **kw)
    if missing:
        rng = np.random.default_rng(seed + 1)
        raw[rng.random((n_sites, n_ind)) < 0.08] = 1.0      # individuals without reads: three equal likelihoods
    return raw


def run_records(e, raw, mode, ign=False, **plan):
    e.set_exact_store(mode)
    e.set_geno_raw(raw, ignore_miss_data=ign)
    e.set_pos_dist(None)
    n = e.plan(ignore_miss_data=ign, **plan)
    s1, s2, std, ext = e.run()
    assert len(s1) == n
    return s1, s2, std.copy(), ext.copy(), e.replay_info()


def assert_same_records(a, b):
    """(r2_ExpG of a pair replayed on the device stays the pair kernel's value -- GSL's long double recurrence has no device twin
    --, the host's replay writes the recurrence's: the same number to 1e-9, not the same bits)"""
    assert close(a[2]["r2_ExpG"], b[2]["r2_ExpG"]).all()
    for col in ("D", "Dp", "r2"):
        sb = same_bits(a[2][col], b[2][col])
        assert sb.all(), (col, int((~sb).sum()), a[2][col][~sb][:3], b[2][col][~sb][:3])
    assert same_bits(a[3]["hap"], b[3]["hap"]).all()
    assert np.array_equal(a[3]["n_iter"], b[3]["n_iter"]) and np.array_equal(a[3]["n_ind_data"], b[3]["n_ind_data"])


# one wavefront per pair with 1..8 individuals per lane, then 2, 4 and 8 wavefronts per pair
@pytest.mark.parametrize("n_ind", [24, 100, 130, 200, 260, 330, 390, 450, 500, 512, 700, 1000, 1500, 2000, 2600])
@pytest.mark.parametrize("ign", [False, True])
def test_device_replay_is_the_hosts_on_every_shape(eng, n_ind, ign):
    n_sites = 60 if n_ind <= 512 else (36 if n_ind <= 1024 else 24)
    raw = uncalled(n_sites, n_ind, seed=100 + n_ind, depth=6.0 if n_ind <= 512 else 14.0, mono_frac=0.3, missing=ign)
    host = run_records(eng, raw, 0, ign)
    dev = run_records(eng, raw, 2, ign)
    assert host[4]["pairs_flagged"] == dev[4]["pairs_flagged"] > len(host[0]) // (10 if n_ind <= 512 else 50)
    assert host[4]["pairs_on_device"] == 0 and host[4]["pairs_on_host"] == host[4]["pairs_replayed"]
    assert dev[4]["exact_store"] == 2 and dev[4]["pairs_on_device"] > 0
    assert dev[4]["pairs_on_device"] + dev[4]["pairs_on_host"] == dev[4]["pairs_replayed"] == host[4]["pairs_replayed"]
    assert_same_records(host, dev)
    want = orc.Oracle(raw, ignore_miss_data=ign, n_threads=4).run()
    check_records(dev[2], dev[3], want)          # 1e-9 everywhere, the degenerate pairs bit for bit


def test_auto_mode_builds_the_store_only_when_it_pays(eng):
    """Default policy: a few flagged pairs are the host's; more than half as many as the matrix has sites (and 4,096) build the store."""
    raw = synth.make_gl_numpy(200, 100, seed=5, depth=8.0)
    few = run_records(eng, raw, 1)
    assert few[4]["exact_store"] == 0 and few[4]["pairs_on_device"] == 0
    raw = uncalled(400, 100, seed=6, mono_frac=0.3)
    many = run_records(eng, raw, 1)
    assert many[4]["pairs_flagged"] > 4096
    assert many[4]["exact_store"] == 2 and many[4]["pairs_on_device"] > 0 and many[4]["exact_store_build_s"] > 0
    host = run_records(eng, raw, 0)
    assert_same_records(host, many)
    again = run_records(eng, raw, 1)             # (a new matrix call: the store is built again, the records are the same)
    assert_same_records(host, again)


@pytest.mark.parametrize("n_ind", [100, 500, 1000])
def test_normalised_input_is_its_own_store(eng, n_ind):
    """ngsld_set_geno_lkl: the caller's normal-space values ARE the reference's bits -- nothing is built, the flagged pairs are
    replayed on the device from the first one on, and the records are the oracle's on the degenerate pairs."""
    raw = uncalled(40, n_ind, seed=31 + n_ind, mono_frac=0.3)
    o = orc.Oracle(raw, n_threads=4)
    want = o.run()
    lkl, maf = o.gl, o.maf
    eng.set_exact_store(1)
    eng.set_geno_lkl(lkl, maf)
    eng.set_pos_dist(None)
    eng.plan()
    s1, s2, std, ext = eng.run()
    info = eng.replay_info()
    assert info["exact_store"] == 1 and info["pairs_on_device"] > 0 and info["exact_store_build_s"] == 0
    check_records(std, ext, want)
    eng.set_exact_store(0)
    eng.set_geno_lkl(lkl, maf)
    eng.set_pos_dist(None)
    eng.plan()
    host = eng.run()
    assert_same_records((s1, s2, std, ext), host)


@pytest.mark.parametrize("batch_pairs", [0, 3000])
def test_text_and_device_records_take_the_same_replay(eng, batch_pairs, monkeypatch):
    """The TSV of a run with the device replay is the host replay's byte for byte (rows formatted on the device, before and
    after the store is there: small batches make the first ones go out without it); so are records left on the device."""
    import torch
    raw = uncalled(300, 150, seed=77, mono_frac=0.25)
    chrs, pos = synth.make_positions(300, 9, max_gap=300)
    labels = [f"{c}:{p}" for c, p in zip(chrs, pos)]
    from ngsld_amd import shard
    pd = shard.pos_dist_from_positions(chrs, pos)
    texts = {}
    for mode in (0, 1, 2):
        eng.set_exact_store(mode)
        eng.set_geno_raw(raw)
        eng.set_pos_dist(pd)
        if batch_pairs:
            eng.set_tuning(batch_pairs=batch_pairs)
        eng.plan(max_kb_dist=20, extend_out=True)
        eng.set_text_output(labels)
        t, fallbacks = eng.run_text()
        assert fallbacks == 0
        texts[mode] = hashlib.md5(t).hexdigest()
        info = eng.replay_info()
        assert (info["pairs_on_device"] > 0) == (mode != 0), (mode, info)
        eng.set_text_output(None, enable=False)
    assert texts[0] == texts[1] == texts[2]
    # records in caller-owned device memory (ngsld_run_device + ngsld_finish_device)
    recs = {}
    for mode in (0, 2):
        eng.set_exact_store(mode)
        eng.set_geno_raw(raw)
        eng.set_pos_dist(pd)
        n = eng.plan(max_kb_dist=20, extend_out=True)
        d_std = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
        d_ext = torch.zeros(n * 40, dtype=torch.uint8, device="cuda")
        eng.run_device(0, 300, d_std.data_ptr(), d_ext.data_ptr(), torch.cuda.current_stream().cuda_stream)
        eng.finish_device()
        recs[mode] = (d_std.cpu().numpy().tobytes(), d_ext.cpu().numpy().tobytes())
    for k in (0, 1):
        a, b = np.frombuffer(recs[0][k], dtype=np.uint64), np.frombuffer(recs[2][k], dtype=np.uint64)
        if k == 0:                               # (word 0 of a standard record is r2_ExpG: see assert_same_records)
            a, b = a.reshape(-1, 4)[:, 1:], b.reshape(-1, 4)[:, 1:]
        assert np.array_equal(a, b)


def test_log_scale_and_filters(eng):
    """--log_scale input (the store is built from logs), --min_maf (rows of monomorphic sites are dropped: what is left flags
    little) and --rnd_sample (records are a subset: the cursor steps over the dropped candidates)."""
    raw = uncalled(120, 80, seed=41, mono_frac=0.3)
    with np.errstate(divide="ignore"):
        logs = np.log(raw)
    logs[np.isneginf(logs)] = -1e15
    for kw in (dict(), dict(min_maf=0.02), dict(rnd_sample=0.3, seed=7)):
        out = {}
        for mode in (0, 2):
            eng.set_exact_store(mode)
            eng.set_geno_raw(logs, log_scale=True)
            eng.set_pos_dist(None)
            n = eng.plan(**kw)
            out[mode] = eng.run()
            assert len(out[mode][0]) == n
        assert_same_records(out[0], out[2])


@pytest.mark.parametrize("n_ind", [100, 500, 1000])
def test_ill_conditioned_pearson_moments_are_settled_on_the_device(eng, n_ind):
    """Deep data: the expected genotypes of a monomorphic site are constant to 1e-5, and every pair of two such sites has a
    cross moment the pair kernels cannot form (1 / (std1 std2) > 2^13: rounds 2-4 left those to the host -- 4 % of the pairs at
    20 % monomorphic sites).  The device-side replay takes two passes over the exact values instead; what is left for the host
    are sites whose mean / std is beyond what GSL's own long double recurrence resolves to 1e-9."""
    raw = uncalled(50, n_ind, seed=300 + n_ind, depth=30.0, mono_frac=0.4)
    o = orc.Oracle(raw, n_threads=4)
    want = o.run()
    std = o.expg.std(axis=1)
    with np.errstate(divide="ignore"):
        cond = 1.0 / (std[want["s1"]] * std[want["s2"]])
    bad = int(np.count_nonzero(np.isfinite(cond) & (cond > 2.0 ** 13)))
    assert bad > len(want) // 20
    host = run_records(eng, raw, 0)
    dev = run_records(eng, raw, 2)
    assert dev[4]["pairs_flagged"] == host[4]["pairs_flagged"] >= bad
    assert dev[4]["pairs_on_host"] <= bad // 20, dev[4]
    assert_same_records(host, dev)               # (r2_ExpG: within 1e-9 of the recurrence the host ran)
    check_records(dev[2], dev[3], want)
    # the same as text: the device's r2_ExpG prints the host's digits (values on a rounding point go to the host)
    labels = [f"s:{k}" for k in range(50)]
    texts = {}
    for mode in (0, 2):
        eng.set_exact_store(mode)
        eng.set_geno_raw(raw)
        eng.set_pos_dist(None)
        eng.plan(extend_out=True)
        eng.set_text_output(labels)
        t, fallbacks = eng.run_text()
        assert fallbacks == 0
        texts[mode] = t
        eng.set_text_output(None, enable=False)
    assert texts[0] == texts[2]


def _bits_but_r2expg(a_std, a_ext, b_std, b_ext):
    assert close(a_std["r2_ExpG"], b_std["r2_ExpG"]).all()
    for col in ("D", "Dp", "r2"):
        assert same_bits(a_std[col], b_std[col]).all(), col
    assert a_ext.tobytes() == b_ext.tobytes()


@pytest.mark.parametrize("slab", [96, 300])
def test_streamed_slabs_build_their_own_store(eng, slab, monkeypatch):
    """A streamed run (row slabs, two contexts alternating: BASELINE configs[4]'s mode) of an un-called matrix: every slab's
    context builds the store of ITS sites from its host buffer and replays on the device -- the resident run's records
    (r2_ExpG of a replayed pair: to 1e-9, see assert_same_records)."""
    from ngsld_amd import shard
    n_sites, n_ind = 700, 60
    raw = uncalled(n_sites, n_ind, seed=55, mono_frac=0.3)
    chrs, pos = synth.make_positions(n_sites, 55, n_chr=2)
    pd = shard.pos_dist_from_positions(chrs, pos)
    kw = dict(max_kb_dist=3)
    monkeypatch.setenv("NGSLD_EXACT_STORE", "0")
    host = capi.run_streamed(lambda b, m: raw[b:b + m], n_sites, n_ind, pd, slab, **kw)
    monkeypatch.setenv("NGSLD_EXACT_STORE", "2")
    dev = capi.run_streamed(lambda b, m: raw[b:b + m], n_sites, n_ind, pd, slab, **kw)
    assert np.array_equal(host[0], dev[0]) and np.array_equal(host[1], dev[1])
    _bits_but_r2expg(host[2], host[3], dev[2], dev[3])
    eng.set_exact_store(2)
    eng.set_geno_raw(raw)
    eng.set_pos_dist(pd)
    eng.plan(**kw)
    s1, s2, std, ext = eng.run()
    assert eng.replay_info()["pairs_on_device"] > len(s1) // 10
    assert np.array_equal(s1, dev[0]) and np.array_equal(s2, dev[1])
    _bits_but_r2expg(std, ext, dev[2], dev[3])


@pytest.mark.parametrize("n_parts", [2, 3])
def test_parts_of_one_process_build_their_own_stores(n_parts, monkeypatch):
    """ngsld_run_multi (`ngsLD --devices`): every part's context replays its flagged pairs on its device."""
    from ngsld_amd import shard
    n_sites, n_ind = 500, 80
    raw = uncalled(n_sites, n_ind, seed=66, mono_frac=0.3)
    chrs, pos = synth.make_positions(n_sites, 66, max_gap=300)
    pd = shard.pos_dist_from_positions(chrs, pos)
    kw = dict(extend_out=True, max_kb_dist=10)
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("NGSLD_EXACT_STORE", mode)
        parts, maf, per = capi.run_multi(raw, pd, [0] * n_parts, **kw)
        out[mode] = [np.concatenate([p[k] for p in parts]) for k in range(4)]
    assert np.array_equal(out["0"][0], out["2"][0]) and np.array_equal(out["0"][1], out["2"][1])
    _bits_but_r2expg(out["0"][2], out["0"][3], out["2"][2], out["2"][3])


def test_a_device_without_room_for_the_store_leaves_the_pairs_to_the_host(eng, monkeypatch):
    """The store is the matrix once more in device memory; where that cannot be had the run does not fail -- its flagged pairs
    are replayed on host threads as before (NGSLD_TEST_EXACT_STORE_NO_ROOM pretends)."""
    raw = uncalled(300, 60, seed=88, mono_frac=0.3)
    want = run_records(eng, raw, 0)
    monkeypatch.setenv("NGSLD_TEST_EXACT_STORE_NO_ROOM", "1")
    got = run_records(eng, raw, 2)
    assert got[4]["exact_store"] == 0 and got[4]["pairs_on_device"] == 0 and got[4]["pairs_on_host"] == want[4]["pairs_on_host"] > 0
    for k in (2, 3):
        assert got[k].tobytes() == want[k].tobytes()


@pytest.mark.parametrize("callback", [False, True])
def test_store_built_beside_the_run_in_site_order(eng, monkeypatch, callback):
    """The exact store is built by a thread of its own while the run that asked for it goes on (engine_replay.hip: exact_builder):
    a batch's device-side replay waits only until the builder has passed the batch's last window.  Here the builder is slowed
    and works in chunks of 16 sites, so that every one of the run's many batches -- text rows and records -- catches up with the
    frontier and waits: the records are the host replay's, bit for bit, and the text is the same bytes."""
    n_sites, n_ind = 900, 60
    raw = uncalled(n_sites, n_ind, 811, mono_frac=0.25)
    raw *= 1.0 + 0.01 * np.random.default_rng(5).random(raw.shape)      # no triple repeats: the builder's memo never hits
    if callback:
        monkeypatch.setenv("NGSLD_PY_REPLAY_CALLBACK", "1")             # ngsld_set_replay_source instead of ngsld_set_replay_matrix
    eng.set_tuning(batch_pairs=2500)
    want = run_records(eng, raw, 0, max_snp_dist=60)
    assert want[4]["pairs_on_host"] > 5000
    monkeypatch.setenv("NGSLD_TEST_EXACT_CHUNK_SITES", "16")
    monkeypatch.setenv("NGSLD_TEST_EXACT_SLOW_US", "1500")
    got = run_records(eng, raw, 2, max_snp_dist=60)
    assert got[4]["exact_store"] == 2 and got[4]["pairs_on_device"] > 5000 and got[4]["pairs_on_host"] * 20 < got[4]["pairs_on_device"]
    assert got[4]["exact_store_build_s"] > 0.05
    assert_same_records(got, want)
    labels = [f"s{k}" for k in range(n_sites)]
    texts = []
    for mode in (0, 2):
        eng.set_exact_store(mode)
        eng.set_geno_raw(raw)                                           # (a new matrix: the store is built again, beside this run)
        eng.set_pos_dist(None)
        eng.plan(max_snp_dist=60)
        eng.set_text_output(labels)
        try:
            text, fallbacks = eng.run_text()
        finally:
            eng.set_text_output(None, enable=False)
        assert fallbacks == 0
        texts.append(hashlib.md5(text).hexdigest())
    assert texts[0] == texts[1]
    eng.set_exact_store(1)


def test_a_failing_replay_source_ends_the_run_that_built_the_store(monkeypatch):
    """A callback source that fails while the builder reads it: the run that started the build returns the error (the builder has
    ended by then), the context stays usable -- the next matrix runs."""
    import ctypes as C
    n_sites, n_ind = 400, 40
    raw = uncalled(n_sites, n_ind, 33, mono_frac=0.3)
    e = capi.Engine(0)
    try:
        e.set_exact_store(2)
        e.set_geno_raw(raw, replay_source=False)
        calls = [0]

        def reader(_user, site_begin, n, dst):
            calls[0] += 1
            if site_begin >= 200:
                return 1
            C.memmove(dst, raw.ctypes.data + int(site_begin) * n_ind * 24, int(n) * n_ind * 24)
            return 0

        cb = capi.READ_FN(reader)
        e._check(e._L.ngsld_set_replay_source(e._h, cb, None))
        monkeypatch.setenv("NGSLD_TEST_EXACT_CHUNK_SITES", "50")
        e.set_pos_dist(None)
        e.plan(max_snp_dist=40)
        with pytest.raises(capi.NgsldError) as err:
            e.run()
        assert "replay source" in str(err.value) and calls[0] >= 5
        monkeypatch.delenv("NGSLD_TEST_EXACT_CHUNK_SITES")
        e.set_geno_raw(raw)                                             # the array itself as the source: all is well
        e.set_pos_dist(None)
        n = e.plan(max_snp_dist=40)
        s1, _, _, _ = e.run()
        assert len(s1) == n and e.replay_info()["pairs_on_device"] > 100
    finally:
        e.close()


@pytest.mark.parametrize("n_ind,cap", [(700, "2"), (1500, "1"), (2000, "12")])
def test_large_cohorts_take_the_lanes_on_short_launches_with_a_cap(eng, monkeypatch, n_ind, cap):
    """Beyond 512 individuals several wavefronts share a pair in the wavefront-per-pair kernel and wait for its four chain lanes;
    such cohorts go to the lane-per-pair kernel on launches of any size, and a lane gives a pair that has not converged after
    `cap` EM steps back to the wavefront kernel (its bit set again), which starts it over.  Whatever the cap, the records are the
    host replay's, bit for bit."""
    raw = uncalled(260, n_ind, 4242 + n_ind, mono_frac=0.25, missing=True)
    want = run_records(eng, raw, 0, max_snp_dist=30)
    monkeypatch.setenv("NGSLD_TEST_LANE_ITER_CAP", cap)
    got = run_records(eng, raw, 2, max_snp_dist=30)
    assert got[4]["pairs_on_device"] > 500 and got[4]["pairs_on_host"] * 20 < got[4]["pairs_on_device"]
    assert_same_records(got, want)
    eng.set_exact_store(1)


@pytest.mark.parametrize("n_ind,n_sites", [(4200, 70), (8000, 40)])
def test_cohorts_beyond_the_wavefront_kernel_are_replayed_by_the_lanes(eng, n_ind, n_sites):
    """Beyond 4,096 individuals the wavefront-per-pair replay kernel has no shape (up to round 5 such cohorts kept the host's
    replay: a third of the pairs of an un-called matrix at ~1e4 pairs/s).  The lane-per-pair kernel has no such limit: it takes
    them on launches of any size, and what it leaves in the bitmap -- ill-conditioned Pearson moments -- is handed to the host
    (replay_leftover_kernel).  Same records as the host's replay, bit for bit."""
    raw = uncalled(n_sites, n_ind, 99 + n_ind, mono_frac=0.3)
    want = run_records(eng, raw, 0)
    got = run_records(eng, raw, 2)
    assert want[4]["pairs_on_device"] == 0 and want[4]["pairs_on_host"] > 100, want[4]
    assert got[4]["exact_store"] == 2 and got[4]["pairs_on_device"] > 100 and got[4]["pairs_on_host"] * 10 < got[4]["pairs_on_device"]
    assert_same_records(got, want)
    eng.set_exact_store(1)


@pytest.mark.parametrize("n_ind,ign", [(100, False), (500, False), (500, True), (700, False)])
def test_quotients_outside_the_shared_reciprocals_range_take_the_plain_divisions(eng, n_ind, ign):
    """The replay kernels form an individual's four quotients tmp_k / sum with ONE refined reciprocal where the hardware's
    division sequence would hand its operands through unscaled (ld_replay_lkl.hip: div_operand_plain), and with the compiler's
    own four divisions where it would not: numerators below 2^-600, products that underflowed to a denormal.  Likelihoods
    of 1e-100 .. 1e-170 beside ordinary ones put individuals of both kinds -- and exact zeros -- into every pair; the records
    are the host replay's (IEEE divisions on the CPU), bit for bit, on the wavefront-per-pair kernel (100, 500 individuals)
    and on the lane-per-pair kernel (700)."""
    # (large cohorts: the matrix of test_large_cohorts_take_the_lanes_on_short_launches_with_a_cap, whose 260 sites in windows of 30
    # flag hundreds of pairs -- 30 sites of 700 individuals at this depth flag two dozen)
    n_sites = 60 if n_ind <= 512 else 260
    plan = {} if n_ind <= 512 else {"max_snp_dist": 30}
    raw = uncalled(n_sites, n_ind, seed=777 + n_ind, depth=6.0, mono_frac=0.3, missing=ign)
    rng = np.random.default_rng(4321 + n_ind)
    tiny = rng.random((n_sites, n_ind)) < 0.15
    scale = np.where(rng.random((n_sites, n_ind)) < 0.5, 1e-100, 1e-170)
    major = np.argmax(raw, axis=2)
    for g in range(3):                               # the two likelihoods that are not the individual's largest, scaled down
        hit = tiny & (major != g) & ~np.all(raw == raw[:, :, :1], axis=2)
        raw[:, :, g] = np.where(hit, raw[:, :, g] * scale, raw[:, :, g])
    host = run_records(eng, raw, 0, ign, **plan)
    dev = run_records(eng, raw, 2, ign, **plan)
    assert host[4]["pairs_flagged"] == dev[4]["pairs_flagged"] > 50, (host[4], dev[4])
    assert dev[4]["exact_store"] == 2 and dev[4]["pairs_on_device"] > 50, dev[4]
    assert dev[4]["pairs_on_device"] + dev[4]["pairs_on_host"] == dev[4]["pairs_replayed"] == host[4]["pairs_replayed"]
    assert_same_records(host, dev)
    eng.set_exact_store(1)


# ---- degenerate sites: the pair kernels leave the EM of their pairs to the replay (ld_prep.hip site_skip_kernel, PairArgs::skip_degenerate) ----
# one wavefront per pair (1 .. 10 individuals per lane) and two (the shapes whose launches take the skip)
@pytest.mark.parametrize("n_ind", [250, 330, 500, 512, 640, 1000])
@pytest.mark.parametrize("ign", [False, True])
def test_degenerate_sites_skip_their_em_and_the_records_are_the_oracles(eng, eng_skip, n_ind, ign):
    n_sites = 80 if n_ind <= 640 else 40
    raw = uncalled(n_sites, n_ind, seed=300 + n_ind, depth=8.0, mono_frac=0.3, missing=ign)
    full = run_records(eng, raw, 2, ign)          # every pair's EM in the pair kernel
    skip = run_records(eng_skip, raw, 2, ign)     # the marked sites' pairs: the replay only
    assert full[4]["sites_degenerate"] == 0 and skip[4]["sites_degenerate"] >= n_sites // 8, (full[4], skip[4])
    assert skip[4]["pairs_flagged"] >= full[4]["pairs_flagged"] > len(full[0]) // 10
    assert skip[4]["pairs_on_device"] + skip[4]["pairs_on_host"] == skip[4]["pairs_replayed"] == skip[4]["pairs_flagged"]
    want = orc.Oracle(raw, ignore_miss_data=ign, n_threads=4).run()
    check_records(skip[2], skip[3], want)         # 1e-9 everywhere, the degenerate pairs bit for bit
    # against the run without the skip: the same bits wherever both replayed or neither did; the pairs only the skip sent to the
    # replay (marked site, outcome not decided by rounding) carry the reference's bits instead of the kernel's -- 1e-9, few
    differ = np.zeros(len(full[0]), dtype=bool)
    for col in ("D", "Dp", "r2"):
        differ |= ~same_bits(full[2][col], skip[2][col])
        assert close(full[2][col], skip[2][col]).all()
    differ |= ~same_bits(full[3]["hap"], skip[3]["hap"]).all(axis=1)
    assert close(full[3]["hap"], skip[3]["hap"]).all()
    assert np.array_equal(full[3]["n_iter"], skip[3]["n_iter"]) and np.array_equal(full[3]["n_ind_data"], skip[3]["n_ind_data"])
    assert same_bits(full[2]["r2_ExpG"], skip[2]["r2_ExpG"]).all()       # (the Pearson moment is the pair kernel's either way)
    assert differ.sum() <= skip[4]["pairs_flagged"] - full[4]["pairs_flagged"]
    assert differ.mean() < 0.15


def test_snp_called_input_marks_no_site(eng_skip):
    raw = synth.make_gl_numpy(300, 500, seed=12, depth=10.0)
    r = run_records(eng_skip, raw, 1)
    assert r[4]["sites_degenerate"] == 0


@pytest.mark.parametrize("n_ind", [100, 2000])
def test_shapes_that_do_not_take_the_skip_mark_no_site(eng_skip, n_ind):
    """The lockstep kernel (a wavefront is spared only what ALL its groups skip) and four or more wavefronts per pair (a replayed
    pair costs 7x a computed one there) measured slower with the skip: engine.hip, look_for_skip."""
    raw = uncalled(40, n_ind, seed=17, mono_frac=0.3)
    r = run_records(eng_skip, raw, 2)
    # (the sites are marked -- the marks also tell ngsld_run that the matrix is un-called -- but these kernels compute every pair)
    assert r[4]["sites_degenerate"] > 0 and r[4]["pairs_on_device"] > 0
    e = capi.Engine(0)
    try:
        e.set_replay(True)
        import os
        os.environ["NGSLD_REPLAY_SKIP"] = "0"
        e2 = capi.Engine(0)
        try:
            full = run_records(e2, raw, 2)
        finally:
            e2.close()
            del os.environ["NGSLD_REPLAY_SKIP"]
    finally:
        e.close()
    assert full[4]["pairs_flagged"] == r[4]["pairs_flagged"]
    for k in (2, 3):
        assert full[k].tobytes() == r[k].tobytes()


def test_degenerate_sites_text_is_the_host_replays(eng, eng_skip):
    """The TSV with the skip on is the TSV without it (and the host replay's) byte for byte: a pair only the skip sends to the
    replay is one whose printed digits no rounding decides."""
    raw = uncalled(300, 500, seed=91, depth=8.0, mono_frac=0.25)
    chrs, pos = synth.make_positions(300, 9, max_gap=300)
    labels = [f"{c}:{p}" for c, p in zip(chrs, pos)]
    from ngsld_amd import shard
    pd = shard.pos_dist_from_positions(chrs, pos)
    md5 = {}
    for name, e, mode in (("host", eng, 0), ("skip", eng_skip, 2), ("skip_auto", eng_skip, 1)):
        e.set_exact_store(mode)
        e.set_geno_raw(raw)
        e.set_pos_dist(pd)
        e.plan(max_kb_dist=20, extend_out=True)
        e.set_text_output(labels)
        t, fallbacks = e.run_text()
        assert fallbacks == 0
        md5[name] = hashlib.md5(t).hexdigest()
        if name != "host":
            assert e.replay_info()["sites_degenerate"] > 0
        e.set_text_output(None, enable=False)
    assert md5["host"] == md5["skip"] == md5["skip_auto"]


# ---- un-called input as TEXT goes in groups: one launch of pair kernels + one lane replay per group, the text of a group made of its
# final records while the next group is computed (engine_run.hip, run_grouped) ----
@pytest.mark.parametrize("group_pairs", ["7000", "1000000000"])
@pytest.mark.parametrize("extend", [True, False])
def test_text_in_groups_is_the_text_batch_by_batch(eng, eng_skip, monkeypatch, group_pairs, extend):
    n_sites, n_ind = 500, 300
    raw = uncalled(n_sites, n_ind, seed=123, depth=8.0, mono_frac=0.25)
    chrs, pos = synth.make_positions(n_sites, 9, max_gap=300)
    labels = [f"{c}:{p}" for c, p in zip(chrs, pos)]
    from ngsld_amd import shard
    pd = shard.pos_dist_from_positions(chrs, pos)

    def text_of(e, mode):
        e.set_exact_store(mode)
        e.set_geno_raw(raw)
        e.set_pos_dist(pd)
        e.set_tuning(batch_pairs=3000)            # several text batches inside a group
        e.plan(max_kb_dist=15, extend_out=extend)
        e.set_text_output(labels)
        t, fallbacks = e.run_text()
        assert fallbacks == 0
        info = e.replay_info()
        e.set_text_output(None, enable=False)
        return hashlib.md5(t).hexdigest(), len(t), info

    monkeypatch.setenv("NGSLD_TEST_TEXT_GROUPS", "0")
    host = text_of(eng, 0)                         # host replay, batch by batch, every pair's EM in the pair kernel
    batchwise = text_of(eng_skip, 2)               # device replay, batch by batch
    monkeypatch.delenv("NGSLD_TEST_TEXT_GROUPS")
    monkeypatch.setenv("NGSLD_TEST_TEXT_GROUP_PAIRS", group_pairs)
    grouped = text_of(eng_skip, 2)
    again = text_of(eng_skip, 1)                   # (the default policy: the groups' own finish builds the store)
    assert host[0] == batchwise[0] == grouped[0] == again[0] and host[1] == grouped[1]
    assert grouped[2]["sites_degenerate"] > 0 and grouped[2]["pairs_on_device"] > grouped[2]["pairs_flagged"] * 0.9
    assert grouped[2]["pairs_on_device"] + grouped[2]["pairs_on_host"] == grouped[2]["pairs_replayed"]


@pytest.mark.parametrize("group_pairs", ["1", "3", "40"])
def test_text_in_groups_with_rows_that_have_no_pair(eng_skip, monkeypatch, group_pairs):
    """All pairs of a short matrix in groups of a few pairs: the last row has no partner and ends up in a group (and a text batch) of
    its own with NO record -- whose text starts where the group's text ends.  Same bytes as batch by batch."""
    n_sites, n_ind = 60, 70
    raw = uncalled(n_sites, n_ind, seed=77, depth=8.0, mono_frac=0.25)
    labels = [f"chr1:{100 + 13 * s}" for s in range(n_sites)]

    def text_of():
        eng_skip.set_exact_store(2)
        eng_skip.set_geno_raw(raw)
        eng_skip.set_pos_dist(None)
        eng_skip.set_tuning(batch_pairs=25)
        n = eng_skip.plan(extend_out=True)
        assert n == n_sites * (n_sites - 1) // 2
        eng_skip.set_text_output(labels)
        t, fallbacks = eng_skip.run_text()
        assert fallbacks == 0
        info = eng_skip.replay_info()
        eng_skip.set_text_output(None, enable=False)
        return t, info

    monkeypatch.setenv("NGSLD_TEST_TEXT_GROUPS", "0")
    plain, _ = text_of()
    monkeypatch.delenv("NGSLD_TEST_TEXT_GROUPS")
    monkeypatch.setenv("NGSLD_TEST_TEXT_GROUP_PAIRS", group_pairs)
    grouped, info = text_of()
    assert info["sites_degenerate"] > 0
    assert grouped.count(b"\n") == n_sites * (n_sites - 1) // 2 and grouped == plain


def test_text_in_groups_of_a_job_without_a_pair(eng_skip, monkeypatch):
    """A job whose window leaves NO pair, in groups: one group without a record.  Its one text batch starts at offs[first record] --
    an entry behind the (empty) scan, never written: round 6's last soak (tools/text_soak.py, cases 42,789 / 42,954 / ... under
    NGSLD_TEST_TEXT_GROUP_PAIRS) got 'h_text.resize(len + len / 8): out of memory' for a length of 0 - garbage.  The bounds kernel now
    takes the text's end for a batch that starts behind the group's last record.  (Only reachable with the test knob: the library
    takes groups from 2**21 pairs.)"""
    n_sites, n_ind = 30, 50
    raw = uncalled(n_sites, n_ind, seed=5, depth=8.0, mono_frac=0.3)
    labels = [f"chr1:{1000 + 5000 * s}" for s in range(n_sites)]
    pd = np.full(n_sites, 5000.0)
    monkeypatch.setenv("NGSLD_TEST_TEXT_GROUP_PAIRS", "2500")
    for _ in range(3):  # (fresh buffers each time: set_geno_raw releases the groups')
        eng_skip.set_exact_store(2)
        eng_skip.set_geno_raw(raw)
        eng_skip.set_pos_dist(pd)
        assert eng_skip.plan(max_kb_dist=1, extend_out=True) == 0
        eng_skip.set_text_output(labels)
        t, fallbacks = eng_skip.run_text()
        eng_skip.set_text_output(None, enable=False)
        assert t == b""          # (the one batch has no row: it may arrive without a text buffer, which run_text counts as records)


def test_text_in_groups_through_the_host_formatter_fallback(eng_skip, monkeypatch, tmp_path):
    """Every third text batch of a grouped run goes out as records (the device formatter's fallback) and is formatted on the host:
    the records come from the group's buffers, the table is the same bytes."""
    import os
    n_sites, n_ind = 400, 260
    raw = uncalled(n_sites, n_ind, seed=321, depth=8.0, mono_frac=0.25)

    def table(path):
        eng_skip.set_exact_store(2)
        eng_skip.set_geno_raw(raw)
        eng_skip.set_pos_dist(None)
        eng_skip.set_tuning(batch_pairs=2500)
        n = eng_skip.plan(extend_out=True)
        eng_skip.set_text_output(None)
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        try:
            assert eng_skip.run_to_fd(0, n_sites, fd, None, None, eng_skip.maf(), 2) == n
        finally:
            os.close(fd)
        eng_skip.set_text_output(None, enable=False)
        return hashlib.md5(open(path, "rb").read()).hexdigest()

    monkeypatch.setenv("NGSLD_TEST_TEXT_GROUPS", "0")
    plain = table(str(tmp_path / "plain.tsv"))
    monkeypatch.delenv("NGSLD_TEST_TEXT_GROUPS")
    monkeypatch.setenv("NGSLD_TEST_TEXT_GROUP_PAIRS", "9000")
    monkeypatch.setenv("NGSLD_TEST_TEXT_FALLBACK_EVERY", "3")
    assert table(str(tmp_path / "grouped.tsv")) == plain
