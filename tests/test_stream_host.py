"""Host side of the streamed (out-of-core) run: window ends, slab planning, ranged binary reads.  No device."""
import gzip
import os

import numpy as np
import pytest

from ngsld_amd import capi, shard, synth


def _pos_dist(n, seed, two_chr=False, max_gap=200):
    rng = np.random.default_rng(seed)
    pd = rng.integers(1, max_gap + 1, size=n).astype(np.float64)
    if two_chr:
        pd[n // 3] = np.inf
        pd[n // 2] = np.inf
    return pd


@pytest.mark.parametrize("kb,snp,two_chr", [(1, 0, False), (2, 0, True), (0, 7, False), (3, 11, True), (0, 0, False)])
def test_window_ends_match_the_literal_walk(kb, snp, two_chr):
    """ngsld_window_ends == the running-sum walk of ngsLD.cpp:240-262 (restated here literally)."""
    n = 700
    pd = _pos_dist(n, 3, two_chr)
    got = capi.window_ends(pd, n, max_kb_dist=kb, max_snp_dist=snp)
    want = np.empty(n, dtype=np.uint32)
    for s1 in range(n):
        dist, e = 0.0, s1 + 1
        while e < n:
            dist += pd[e]
            if kb > 0 and dist > kb * 1000:
                break
            if snp > 0 and e - s1 > snp:
                break
            e += 1
        want[s1] = e
    assert np.array_equal(got, want)
    assert np.array_equal(got, shard.row_ends(pd, max_kb_dist=kb, max_snp_dist=snp))
    assert np.array_equal(capi.window_ends(None, 5, max_kb_dist=0), np.full(5, 5, dtype=np.uint32))


@pytest.mark.parametrize("cap", [40, 64, 200, 5000])
def test_slabs_cover_every_row_once_and_hold_its_window(cap):
    n = 3000
    pd = _pos_dist(n, 5, two_chr=True)
    ends = capi.window_ends(pd, n, max_kb_dist=2)
    assert int((ends - np.arange(n)).max()) <= 40          # the widest window fits the smallest cap tried
    slabs = capi.plan_slabs(pd, n, cap, max_kb_dist=2)
    assert slabs["row_begin"][0] == 0 and slabs["row_end"][-1] == n
    assert np.array_equal(slabs["row_begin"][1:], slabs["row_end"][:-1])
    for sl in slabs:
        r0, r1, hi = int(sl["row_begin"]), int(sl["row_end"]), int(sl["site_end"])
        assert r1 > r0 and hi - r0 <= cap and hi <= n
        assert int(ends[r0:r1].max()) <= hi
        # greedy: one more row would not have fitted
        assert r1 == n or max(hi, int(ends[r1])) - r0 > cap
    if cap >= n:
        assert len(slabs) == 1


def test_slab_planning_errors():
    n = 500
    pd = _pos_dist(n, 6)
    with pytest.raises(capi.NgsldError) as e:
        capi.plan_slabs(pd, n, 5, max_kb_dist=2)             # windows of ~20 sites cannot fit 5
    assert e.value.code == capi.ERR_NOMEM
    with pytest.raises(capi.NgsldError) as e:
        capi.plan_slabs(None, n, n - 1, max_kb_dist=0)       # all pairs: row 0 needs every site
    assert e.value.code == capi.ERR_NOMEM
    assert len(capi.plan_slabs(None, n, n, max_kb_dist=0)) == 1
    assert len(capi.plan_slabs(None, n, 60, max_snp_dist=10)) > 1


def test_slab_budget_is_monotone_and_zero_when_too_small():
    assert capi.slab_sites_for_budget(500, 1 << 30) == 0
    a, b = capi.slab_sites_for_budget(500, 16 << 30), capi.slab_sites_for_budget(500, 64 << 30)
    assert 0 < a < b
    assert capi.slab_sites_for_budget(2000, 64 << 30) < b
    # two contexts, each planes of 24 * np bytes per site: a 64 GiB budget holds < 32 GiB of planes per slab
    assert b * 24 * 512 < 32 << 30


@pytest.mark.parametrize("compressed", [False, True])
def test_ranged_binary_read(tmp_path, compressed):
    raw = synth.make_gl_numpy(50, 7, 11, depth=4.0)
    path = str(tmp_path / ("g.glf" if not compressed else "g.glf.bgz"))
    if compressed:
        with gzip.open(path, "wb") as fh:
            fh.write(raw.tobytes())
    else:
        raw.tofile(path)
    for b, m in [(0, 50), (0, 1), (13, 20), (49, 1)]:
        assert np.array_equal(capi.read_geno_bin_range(path, 7, b, m), raw[b:b + m])
    with pytest.raises(capi.NgsldError):
        capi.read_geno_bin_range(path, 7, 40, 11)             # runs past the end
    with pytest.raises(capi.NgsldError):
        capi.read_geno_bin_range(os.path.join(str(tmp_path), "missing"), 7, 0, 1)


# n_ind -> padded individuals per genotype plane of the kernel shape pair_config picks: lane groups of 8 / 16 /
# 32 up to 128 (and the odd 32-multiples up to 224), one wavefront per pair up to 960 (64 per slot), several wavefronts
# beyond -- eight slots per lane, nine or ten just past a doubling -- and whole 64-blocks for the streaming kernel
PLANE_SHAPES = [(24, 24), (64, 64), (65, 80), (100, 112), (128, 128), (160, 160), (200, 224), (250, 256), (500, 512), (512, 512),
                (513, 576), (576, 576), (577, 640), (640, 640), (641, 704), (832, 832), (833, 896), (897, 960), (960, 960), (961, 1024), (1000, 1024), (1024, 1024),
                (1025, 1152), (1152, 1152), (1153, 1280), (1280, 1280), (1281, 1408), (1664, 1664), (1700, 1792), (1900, 2048),
                (1921, 2048), (2000, 2048), (2049, 2304), (2305, 2560), (2561, 2816), (3328, 3328), (3500, 3584), (3800, 4096),
                (3841, 4096), (4096, 4096), (4097, 4608), (4609, 5120), (5120, 5120), (5121, 5632), (6000, 6144), (6656, 6656),
                (7000, 7168), (7680, 7680), (7681, 7744), (10000, 10048)]
# (several wavefronts per pair in the a/b form, round 3: 2 / 4 / 8 wavefronts x 10..15 slots -- with --ignore_miss_data up to 13; the
# helper prices the wider of the two layouts: 1,900 individuals are 2 x 15 x 64 = 1,920 slots, or 4 x 8 x 64 = 2,048 under the flag)


@pytest.mark.parametrize("n_ind,np_want", PLANE_SHAPES)
def test_budget_helper_sizes_slabs_with_the_engines_own_plane_layout(n_ind, np_want):
    """ngsld_slab_sites_for_budget must count 3 x 24 * np + 64 bytes per site -- the planes and, since round 5, the exact store of
    the device-side replay with its individual-major copy, which input that is not SNP-called has built -- with the np the engine will really allocate
    (round 2's helper took the a/b kernel's layout for 513..1024 while the engine ran the two-wavefront one: slabs up to
    11 % too large).  np is read back from the helper at two budgets -- and pins pair_config's shape boundaries on the CPU."""
    lo, hi = capi.slab_sites_for_budget(n_ind, 64 << 30), capi.slab_sites_for_budget(n_ind, 192 << 30)
    assert 0 < lo < hi
    per_site = (64 << 30) / (hi - lo)                      # d(budget / 2) / d(sites)
    assert abs(per_site - (72 * np_want + 64)) < 1e-3 * per_site, (n_ind, per_site, (per_site - 64) / 72)


def test_budget_helper_prices_the_planes_alone_on_request():
    """ngsld_sites_for_budget(..., 1): what a run NEEDS -- the planes; 3 is ngsld_slab_sites_for_budget.  The command line asks with
    1 before it refuses a matrix it cannot stream (text input, no window): the exact store is optional (round 5 priced it into
    the resident decision and such jobs aborted at a third of the memory they used to run in)."""
    for n_ind in (100, 500, 2000):
        three = capi.sites_for_budget(n_ind, 64 << 30, 3)
        assert three == capi.slab_sites_for_budget(n_ind, 64 << 30) > 0
        one, two = capi.sites_for_budget(n_ind, 64 << 30, 1), capi.sites_for_budget(n_ind, 64 << 30, 2)
        assert three < two < one and 2.9 * three < one < 3.1 * three
