"""Streamed (out-of-core) runs, BASELINE configs[4]'s mode: the matrix is cut into row slabs with halos and two
contexts alternate on the device.  The records must be those of the resident run, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from ngsld_amd import capi, shard, synth

pytestmark = pytest.mark.gpu


def _resident(engine, raw, pd, **kw):
    engine.set_geno_raw(raw, ignore_miss_data=kw.get("ignore_miss_data", False))
    engine.set_pos_dist(pd)
    engine.plan(**kw)
    return engine.run() + (engine.maf(),)


def _same(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2].tobytes() == b[2].tobytes()
    if a[3] is None:
        assert b[3] is None
    else:
        assert a[3].tobytes() == b[3].tobytes()


@pytest.mark.parametrize("kw", [
    dict(max_kb_dist=2),
    dict(max_kb_dist=2, extend_out=False),
    dict(max_kb_dist=3, max_snp_dist=9, min_maf=0.15),
    dict(max_snp_dist=25),
    dict(max_kb_dist=2, rnd_sample=0.4, seed=77),
    dict(max_kb_dist=2, ignore_miss_data=True),
])
@pytest.mark.parametrize("slab", [64, 150, 100000])
def test_streamed_equals_resident(engine, kw, slab):
    n_sites, n_ind = 900, 30
    raw = synth.make_gl_numpy(n_sites, n_ind, 41, depth=4.0)
    if kw.get("ignore_miss_data"):
        rng = np.random.default_rng(8)
        raw[rng.random((n_sites, n_ind)) < 0.2] = 1.0
    # degenerate sites (monomorphic, no data) on both sides of slab borders: their pairs go through the exact-order
    # replay, whose source is the slab's host buffer in a streamed run and the whole matrix in the resident one
    for s in (5, 63, 64, 149, 150, 151, 400, 899):
        raw[s] = [1.0, 0.0, 0.0] if s % 2 else 1.0 / 3.0
    chrs, pos = synth.make_positions(n_sites, 41, n_chr=2)
    pd = shard.pos_dist_from_positions(chrs, pos)
    want = _resident(engine, raw, pd, **kw)
    calls = []

    def read(b, m):
        calls.append((b, m))
        return raw[b:b + m]

    assert engine.replay_stats()[0] > 0
    got = capi.run_streamed(read, n_sites, n_ind, pd, slab, **kw)
    _same(got, want)
    assert np.array_equal(got[4], want[4], equal_nan=True)      # est_maf of every site, slab-independent (NaN: nobody has data)
    slabs = capi.plan_slabs(pd, n_sites, slab, **{k: v for k, v in kw.items() if k in ("max_kb_dist", "max_snp_dist")})
    assert got[5] == len(slabs) == len(calls)
    assert calls == [(int(s["row_begin"]), int(s["site_end"] - s["row_begin"])) for s in slabs]
    assert (len(slabs) == 1) == (slab >= n_sites)


def test_streamed_multi_wavefront_cohort(engine):
    """n_ind = 2000 (configs[4]'s cohort, 4 wavefronts per pair), 500 kb window over ~1 kb gaps."""
    torch = pytest.importorskip("torch")
    n_sites, n_ind = 6000, 2000
    raw = synth.make_gl_torch(n_sites, n_ind, 5, torch.device("cuda:0")).cpu().numpy()
    chrs, pos = synth.make_positions(n_sites, 5, max_gap=2000)
    pd = shard.pos_dist_from_positions(chrs, pos)
    want = _resident(engine, raw, pd, max_kb_dist=500)
    got = capi.run_streamed(lambda b, m: raw[b:b + m], n_sites, n_ind, pd, 1500, max_kb_dist=500)
    assert got[5] >= 5 and len(got[0]) > 2_000_000
    _same(got, want)


@pytest.mark.parametrize("thresholds,kernel", [((0.0, 0.0), "hard"), ((0.5, 0.5), "hard"), ((0.4, 0.9), None)])
def test_streamed_called_genotypes_keep_the_resident_kernel(engine, thresholds, kernel):
    """--call_geno with N_thresh == call_thresh leaves only called genotypes and "no data" (gen_func.cpp:886-914), which
    is known before the first slab is read: the streamed run takes the genotype-combination kernel like the resident one
    and gives the same bits.  With a gap between the thresholds some triples stay likelihoods and both keep to the
    per-individual kernels."""
    n_sites, n_ind = 1200, 90
    raw = synth.make_gl_numpy(n_sites, n_ind, 53, depth=3.0)
    raw[np.random.default_rng(3).random((n_sites, n_ind)) < 0.1] = 1.0 / 3.0
    chrs, pos = synth.make_positions(n_sites, 53, n_chr=2)
    pd = shard.pos_dist_from_positions(chrs, pos)
    engine.set_geno_raw(raw, call_geno=thresholds)
    if kernel is not None:
        assert engine.pair_kernel() == kernel
    else:
        assert engine.pair_kernel() != "hard"
    engine.set_pos_dist(pd)
    engine.plan(max_kb_dist=3)
    want = engine.run() + (engine.maf(),)
    got = capi.run_streamed(lambda b, m: raw[b:b + m], n_sites, n_ind, pd, 200, call_geno=thresholds, max_kb_dist=3)
    assert got[5] >= 5 and len(got[0]) > 10000
    _same(got, want)


def test_a_later_slab_that_is_not_all_called_goes_on_per_individual(engine):
    """A --call_geno job whose first slabs ran on the genotype-combination kernel and whose LATER slab holds a triple that is
    neither called nor missing (a NaN from a text file sets no NaN status): that slab and the ones after it are loaded again for
    the per-individual kernels and the job completes (up to round 5 it ended with an error behind a truncated table).  The
    resident run holds that triple and runs per individual as a whole: the same pairs, the same nIter / sample_size, every value
    within 1e-9 -- bit for bit from the disagreeing slab on."""
    n_sites, n_ind = 1200, 40
    raw = synth.make_gl_numpy(n_sites, n_ind, 61, depth=3.0)
    raw[900, 7, :] = np.nan
    chrs, pos = synth.make_positions(n_sites, 61)
    pd = shard.pos_dist_from_positions(chrs, pos)
    engine.set_geno_raw(raw, text=True, call_geno=(0.0, 0.0))
    assert engine.pair_kernel() != "hard"
    engine.set_pos_dist(pd)
    engine.plan(max_kb_dist=3)
    want = engine.run()
    got = capi.run_streamed(lambda b, m: raw[b:b + m], n_sites, n_ind, pd, 200, call_geno=(0.0, 0.0), text=True, max_kb_dist=3)
    assert got[5] >= 5 and len(got[0]) == len(want[0]) > 10000
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[3]["n_iter"], want[3]["n_iter"]) and np.array_equal(got[3]["n_ind_data"], want[3]["n_ind_data"])
    for f in ("r2_ExpG", "D", "Dp", "r2"):
        a, b = got[2][f], want[2][f]
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.all((np.abs(a - b) <= 1e-9) | (a == b) | np.isnan(a)), f
    late = got[0] >= 1000                                     # (rows of the slabs behind the disagreeing one)
    assert late.sum() > 1000 and got[2][late].tobytes() == want[2][late].tobytes()


def test_streamed_errors(engine):
    n_sites, n_ind = 300, 12
    raw = synth.make_gl_numpy(n_sites, n_ind, 43, depth=4.0)
    chrs, pos = synth.make_positions(n_sites, 43)
    pd = shard.pos_dist_from_positions(chrs, pos)
    with pytest.raises(capi.NgsldError) as e:                  # the window does not fit the slab
        capi.run_streamed(lambda b, m: raw[b:b + m], n_sites, n_ind, pd, 8, max_kb_dist=2)
    assert e.value.code == capi.ERR_NOMEM and "does not fit" in str(e.value)

    def failing(b, m):
        if b > 0:
            raise IOError("disk gone")
        return raw[b:b + m]

    with pytest.raises(capi.NgsldError) as e:                  # a reader failure on the second slab stops the job
        capi.run_streamed(failing, n_sites, n_ind, pd, 64, max_kb_dist=2)
    assert "cannot read" in str(e.value)
    bad = raw.copy()
    bad[250, 3, :] = -1.0                                       # log(-1): the NaN check of read_geno (read_data.cpp:42-45)
    with pytest.raises(capi.NgsldError) as e:
        capi.run_streamed(lambda b, m: bad[b:b + m], n_sites, n_ind, pd, 64, max_kb_dist=2)
    assert e.value.code == capi.ERR_NAN


@pytest.mark.parametrize("flags", [["--max_kb_dist", "3", "--extend_out"],
                                   ["--max_kb_dist", "3", "--rnd_sample", "0.5", "--seed", "9", "--min_maf", "0.1"],
                                   ["--max_kb_dist", "3", "--probs", "--call_geno"]])
def test_cli_streamed_output_is_identical(tmp_path, flags):
    """The drop-in binary forced to stream (NGSLD_TEST_SLAB_SITES) writes the same bytes as the resident run."""
    n_sites, n_ind = 1500, 40
    raw = synth.make_gl_numpy(n_sites, n_ind, 47, depth=5.0)
    chrs, pos = synth.make_positions(n_sites, 47, n_chr=2)
    g, p = str(tmp_path / "in.glf"), str(tmp_path / "in.pos")
    raw.tofile(g)
    synth.write_pos(p, chrs, pos)
    cmd = [capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p, "--verbose", "0",
           "--n_threads", "4"] + flags
    a = subprocess.run(cmd, capture_output=True)
    b = subprocess.run(cmd, capture_output=True, env=dict(os.environ, NGSLD_TEST_SLAB_SITES="100"))
    assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
    assert a.stdout == b.stdout and a.stdout.count(b"\n") > 1000
    # all pairs cannot be cut into slabs: a job that fits the device falls back to the resident run
    ap = cmd[:-len(flags)] + ["--max_kb_dist", "0", "--max_snp_dist", "30"]
    c = subprocess.run(ap[:-1] + ["0"], capture_output=True, env=dict(os.environ, NGSLD_TEST_SLAB_SITES="100"))
    d = subprocess.run(ap[:-1] + ["0"], capture_output=True)
    assert c.returncode == 0 and d.returncode == 0 and c.stdout == d.stdout
    assert c.stdout.count(b"\n") == 1 + n_sites * (n_sites - 1) // 2
