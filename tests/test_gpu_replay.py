"""GPU: the exact-order replay (include/ngsld.h).  Pairs whose outcome the reference's own rounding decides -- a site
monomorphic in the estimated haplotypes (D' and r2 are 0/0-type quotients: -nan, 0 or inf), a frequency that ties
--min_maf, a site whose expected genotypes are constant up to rounding -- come out as the REFERENCE'S BITS on every path
records can take: the sink (host records), the device formatter (text) and caller-owned device memory."""
import numpy as np
import pytest

from ngsld_amd import capi, synth
from oracle import orc
from util import check_records, same_bits

pytestmark = pytest.mark.gpu


def degenerate_matrix(n_ind=37, n_sites=14, seed=7, all_called=False):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 3, size=(n_sites, n_ind))
    raw = np.zeros((n_sites, n_ind, 3))
    for s in range(n_sites):
        raw[s, np.arange(n_ind), g[s]] = 1.0
    raw[3] = 0.0
    raw[3, :, 0] = 1.0                      # monomorphic
    raw[5] = 1.0 / 3.0                      # no data at all
    if all_called:
        raw[7] = 0.0
        raw[7, :, 2] = 1.0                  # monomorphic for the other allele
    else:
        raw[7, :, :] = [0.2, 0.3, 0.5]      # the same uninformative triple everywhere
    raw[9, ::2] = 1.0 / 3.0                 # half missing
    return raw


def mixed_matrix(n_ind, n_sites=40, seed=11):
    """Likelihood data with a few degenerate sites mixed in (keeps the per-individual kernels in play)."""
    raw = synth.make_gl_numpy(n_sites, n_ind, seed=seed, depth=4.0)
    raw[4] = [1.0, 0.0, 0.0]
    raw[11] = 1.0 / 3.0
    raw[17] = [0.0, 0.0, 1.0]
    raw[23, :, :] = [0.25, 0.5, 0.25]
    return raw


def all_bits_equal(std, ext, want):
    for mine, theirs in (("D", "D"), ("Dp", "Dp"), ("r2", "r2"), ("r2_ExpG", "r2pear")):
        sb = same_bits(std[mine], want[theirs])
        assert sb.all(), (mine, np.flatnonzero(~sb)[:5], std[mine][~sb][:3], want[theirs][~sb][:3])
    assert same_bits(ext["hap"], want["hap"]).all()
    assert np.array_equal(ext["n_iter"], want["n_iter"]) and np.array_equal(ext["n_ind_data"], want["n_ind_data"])


@pytest.mark.parametrize("source", ["matrix", "callback"])
@pytest.mark.parametrize("ign", [False, True])
@pytest.mark.parametrize("hard_kernel", [True, False])
def test_degenerate_pairs_are_the_reference_bits(engine, ign, hard_kernel, source, monkeypatch):
    """Both forms of the replay source: the caller's array read in place (ngsld_set_replay_matrix, what the binding uses) and
    the reader callback (ngsld_set_replay_source)."""
    if source == "callback":
        monkeypatch.setenv("NGSLD_PY_REPLAY_CALLBACK", "1")
    raw = degenerate_matrix(all_called=hard_kernel)
    o = orc.Oracle(raw, ignore_miss_data=ign)
    want = o.run()
    engine.set_geno_raw(raw, ignore_miss_data=ign)
    assert (engine.pair_kernel() == "hard") == hard_kernel
    engine.set_pos_dist(None)
    assert engine.plan(ignore_miss_data=ign) == len(want)
    s1, s2, std, ext = engine.run()
    n_rep, _ = engine.replay_stats()
    assert n_rep > 0
    check_records(std, ext, want)
    flagged = np.isin(s1, [3, 5, 7]) | np.isin(s2, [3, 5, 7])      # every pair with a degenerate site was replayed
    all_bits_equal(std[flagged], ext[flagged], want[flagged])


@pytest.mark.parametrize("n_ind", [24, 100, 300, 500, 1000, 2100, 4500])
def test_every_kernel_family_flags_its_degenerate_pairs(engine, n_ind):
    raw = mixed_matrix(n_ind, n_sites=40 if n_ind <= 1000 else 26)
    o = orc.Oracle(raw, n_threads=4)
    want = o.run()
    engine.set_geno_raw(raw)
    engine.set_pos_dist(None)
    assert engine.plan() == len(want)
    s1, s2, std, ext = engine.run()
    assert engine.replay_stats()[0] > 0
    check_records(std, ext, want)
    flagged = np.isin(s1, [4, 11, 17, 23]) | np.isin(s2, [4, 11, 17, 23])
    all_bits_equal(std[flagged], ext[flagged], want[flagged])


def test_replay_without_a_source_reads_the_device_planes(engine):
    """No source registered: same operation order on the device's own prepped values.  Hard calls are exact either way
    (0, 1 and exp(log(1/3))-type values aside), so the called sites still give the reference's bits."""
    raw = degenerate_matrix()
    raw[5] = raw[4]                          # (no 1/3 triples: their exp/log differs by an ulp between host and device)
    raw[9] = raw[8]
    raw[7] = raw[6]
    o = orc.Oracle(raw)
    want = o.run()
    engine.set_geno_raw(raw, replay_source=False)
    engine.set_pos_dist(None)
    engine.plan()
    s1, s2, std, ext = engine.run()
    assert engine.replay_stats()[0] > 0
    check_records(std, ext, want)


def test_replay_off_keeps_the_kernels_values(engine):
    raw = degenerate_matrix()
    engine.set_geno_raw(raw)
    engine.set_pos_dist(None)
    engine.set_replay(False)
    try:
        engine.plan()
        engine.run()
        assert engine.replay_stats()[0] == 0
    finally:
        engine.set_replay(True)


def test_device_records_are_patched(engine):
    """ngsld_run_device: on the ctx's own stream the records are final on return; on a caller's stream after
    ngsld_finish_device."""
    import torch
    raw = mixed_matrix(300)
    o = orc.Oracle(raw, n_threads=4)
    want = o.run()
    engine.set_geno_raw(raw)
    engine.set_pos_dist(None)
    n = engine.plan()
    dev = torch.device("cuda", 0)
    for own_stream in (True, False):
        d_std = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
        d_ext = torch.zeros(n * 40, dtype=torch.uint8, device=dev)
        if own_stream:
            engine.run_device(0, engine.n_sites, d_std.data_ptr(), d_ext.data_ptr(), None)
        else:
            st = torch.cuda.Stream()
            engine.run_device(0, engine.n_sites, d_std.data_ptr(), d_ext.data_ptr(), st.cuda_stream)
            engine.finish_device()
        assert engine.replay_stats()[0] > 0
        std = d_std.cpu().numpy().view(capi.REC_STD)
        ext = d_ext.cpu().numpy().view(capi.REC_EXT)
        check_records(std, ext, want)
        # (records that stay on the device are replayed where the NUMBERS are ill-conditioned -- the monomorphic sites --
        # not where only a printed digit or the sign of a rounded zero is at stake)
        flagged = np.isin(want["s1"], [4, 17]) | np.isin(want["s2"], [4, 17])
        all_bits_equal(std[flagged], ext[flagged], want[flagged])


def test_a_second_device_run_finishes_the_pending_one_first(engine):
    """One pending run per context.  Two ngsld_run_device calls on a caller's stream with NO ngsld_finish_device in between --
    then a re-plan on top: the earlier run's flagged records must still come out replayed (round 2 cleared the flag buffer
    under the first run's kernels and dropped its pending state, leaving the kernels' own values in those records)."""
    import torch
    raw = mixed_matrix(300)
    want = orc.Oracle(raw, n_threads=4).run()
    engine.set_geno_raw(raw)
    engine.set_pos_dist(None)
    n = engine.plan()
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream()
    bufs = [(torch.zeros(n * 32, dtype=torch.uint8, device=dev), torch.zeros(n * 40, dtype=torch.uint8, device=dev))
            for _ in range(3)]
    engine.run_device(0, engine.n_sites, bufs[0][0].data_ptr(), bufs[0][1].data_ptr(), st.cuda_stream)
    engine.run_device(0, engine.n_sites, bufs[1][0].data_ptr(), bufs[1][1].data_ptr(), st.cuda_stream)   # settles run 1
    engine.run_device(0, engine.n_sites, bufs[2][0].data_ptr(), bufs[2][1].data_ptr(), st.cuda_stream)   # settles run 2
    assert engine.plan() == n                                                                            # settles run 3
    torch.cuda.synchronize()
    flagged = np.isin(want["s1"], [4, 17]) | np.isin(want["s2"], [4, 17])
    for d_std, d_ext in bufs:
        std = d_std.cpu().numpy().view(capi.REC_STD)
        ext = d_ext.cpu().numpy().view(capi.REC_EXT)
        check_records(std, ext, want)
        all_bits_equal(std[flagged], ext[flagged], want[flagged])
    engine.finish_device()                                                                               # nothing pending: a no-op


def test_device_text_of_replayed_pairs_equals_the_host_text(engine):
    """Text batches: the flagged records are patched on the device before the rows are formatted."""
    raw = mixed_matrix(100)
    chrs, pos = synth.make_positions(raw.shape[0], 3, max_gap=50, n_chr=1)
    from ngsld_amd import shard
    pd = shard.pos_dist_from_positions(chrs, pos)
    labels = [f"chr{c}:{p}" for c, p in zip(chrs, pos)]
    engine.set_geno_raw(raw)
    engine.set_pos_dist(pd)
    engine.plan(extend_out=True)
    s1, s2, std, ext = engine.run()
    maf = engine.maf()
    cum = np.cumsum(np.where(np.isinf(pd), 0.0, pd))
    host = "".join(capi.format_pair(labels[a], labels[b], float(cum[b] - cum[a]), std[k:k + 1], ext[k:k + 1], maf[a], maf[b])
                   for k, (a, b) in enumerate(zip(s1.astype(int), s2.astype(int))))
    engine.set_text_output(labels)
    try:
        text, fallbacks = engine.run_text()
    finally:
        engine.set_text_output(None, enable=False)
    assert engine.replay_stats()[0] > 0 and fallbacks == 0
    assert text.decode() == host


@pytest.mark.parametrize("n_ind", [60, 500])
def test_min_maf_equal_to_a_frequency(engine, n_ind):
    """--min_maf set to a site's own est_maf (as the reference computes it): the site is kept, as in the reference
    (`maf < min_maf` is false for equal values), whatever the last bits of the device's block-reduced est_maf are."""
    raw = synth.make_gl_numpy(50, n_ind, seed=21, depth=3.0)
    o0 = orc.Oracle(raw)
    for pick in (7, 19, 33):
        m = float(o0.maf[pick])
        o = orc.Oracle(raw, min_maf=m, n_threads=4)
        want = o.run()
        engine.set_geno_raw(raw)
        engine.set_pos_dist(None)
        assert engine.plan(min_maf=m) == len(want)
        assert engine.replay_stats()[1] >= 1
        assert engine.maf()[pick] == m                       # the tied site carries the reference's own est_maf now
        s1, s2, std, ext = engine.run()
        assert np.array_equal(s1, want["s1"]) and np.array_equal(s2, want["s2"])
        check_records(std, ext, want)
