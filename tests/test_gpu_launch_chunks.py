"""GPU: long item / run lists go out as several launches (HIP addresses a launch's threads with 32 bits per dimension;
a grid beyond 2^32 threads silently wraps -- 50,000 x 1,000 all pairs on one device once computed 14 % of its pairs)."""
import os

import numpy as np
import pytest

from ngsld_amd import capi, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("n_sites,n_ind,seed", [(300, 20, 1), (260, 100, 2), (200, 300, 3), (90, 600, 4), (24, 4200, 5)])
def test_chunked_launches_are_bit_identical(engine, n_sites, n_ind, seed):
    """Every kernel family with the cap forced down to 5 workgroups per launch: same records, bit for bit."""
    raw = synth.make_gl_numpy(n_sites, n_ind, seed, depth=5.0)
    out = []
    for cap in (None, "5"):
        if cap:
            os.environ["NGSLD_TEST_MAX_BLOCKS"] = cap
        try:
            engine.set_geno_raw(raw)
            engine.set_pos_dist(None)
            engine.plan(extend_out=True, rnd_sample=0.7, seed=99)
            out.append(engine.run())
        finally:
            os.environ.pop("NGSLD_TEST_MAX_BLOCKS", None)
    assert len(out[0][0]) > 0
    for a, b in zip(*out):
        assert a.tobytes() == b.tobytes()


def test_more_than_2_to_32_threads_in_one_plan():
    """40,000 x 513 all pairs = 8e8 candidates = 5e7 items of the two-wavefront kernel (6.4e9 threads), thinned to ~1.6e6
    computed pairs by --rnd_sample: every record of the plan must have been written."""
    dev = torch.device("cuda", 0)
    n_sites, n_ind = 40_000, 513
    raw = synth.make_gl_torch(n_sites, n_ind, 11, dev, depth=8.0)
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
        del raw
        eng.set_pos_dist(None)
        n = eng.plan(extend_out=True, rnd_sample=0.002, seed=7)
        assert 1.2e6 < n < 2.0e6
        d_std = torch.full((n * 4,), float("nan"), dtype=torch.float64, device=dev)
        d_ext = torch.full((n * 10,), -1, dtype=torch.int32, device=dev)          # n_iter = -1: "never written"
        eng.run_device(0, n_sites, d_std.data_ptr(), d_ext.data_ptr(), None)
        meta = d_ext.view(-1, 10)[:, 8:10]
        assert bool(torch.all(meta[:, 0] == n_ind)) and bool(torch.all((meta[:, 1] >= 0) & (meta[:, 1] <= 100)))
        hap = d_ext.view(torch.float64).view(-1, 5)[:, :4]
        assert float((hap.sum(dim=1) - 1).abs().max()) < 1e-12
        assert bool(torch.all(torch.isfinite(d_std.view(-1, 4)[:, 1])))           # D of every pair
    finally:
        eng.close()


@pytest.mark.parametrize("n_ind,n_sites", [(520, 2300), (1100, 2200), (2100, 2150)])
def test_tiled_workgroup_order_of_the_multi_wavefront_kernel(n_ind, n_sites):
    """Rows of 32 and more items (2,048+ candidates) of the multi-wavefront kernel are worked through in tiles of 64 rows x 8
    items, ids without an item are empty workgroups (launch_pair_kernel): same records, bit for bit, as the plain item order
    (NGSLD_TEST_TILES=0), through record batches and through ngsld_run_device; the first rows against the oracle."""
    import os
    import torch
    from oracle import orc
    from util import check_records
    raw = synth.make_gl_numpy(n_sites, n_ind, 77 + n_ind, depth=4.0)
    os.environ["NGSLD_TEST_TILE_MIN_MB"] = "0"                       # (tiles are for matrices beyond the 256 MB Infinity Cache)
    os.environ["NGSLD_PAIR_KERNEL"] = "multi"                   # (n_ind 520 would run on one wavefront per pair)
    try:
        eng = capi.Engine(0)
    finally:
        del os.environ["NGSLD_PAIR_KERNEL"]
    try:
        eng.set_geno_raw(raw)
        assert eng.pair_kernel() == "multi"
        eng.set_pos_dist(None)
        n = eng.plan(0, 0, 0.0, False, True)
        assert n == n_sites * (n_sites - 1) // 2
        eng.set_tuning(batch_pairs=300_000)                    # batches of ~140 rows: two whole row blocks and a partial one
        tiled = eng.run()
        dev = torch.device("cuda", 0)
        d_std = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
        d_ext = torch.zeros(n * 40, dtype=torch.uint8, device=dev)

        def device_records():
            d_std.zero_(); d_ext.zero_()
            torch.cuda.synchronize()
            eng.run_device(0, n_sites, d_std.data_ptr(), d_ext.data_ptr())
            torch.cuda.synchronize()
            return d_std.cpu().numpy().tobytes(), d_ext.cpu().numpy().tobytes()

        tiled_dev = device_records()
        os.environ["NGSLD_TEST_TILES"] = "0"
        try:
            plain = eng.run()
            plain_dev = device_records()
        finally:
            del os.environ["NGSLD_TEST_TILES"]
        for a, b in zip(tiled, plain):
            assert a.tobytes() == b.tobytes()
        assert tiled_dev == plain_dev
        os.environ["NGSLD_TEST_MAX_BLOCKS"] = "300"                 # a tile grid beyond the launch cap: that row group in plain order, chunked
        try:
            assert device_records() == plain_dev
        finally:
            del os.environ["NGSLD_TEST_MAX_BLOCKS"]
        rows = 12
        want = orc.Oracle(raw, n_threads=8).run(0, rows)
        m = tiled[0] < rows
        assert m.sum() == len(want)
        check_records(tiled[2][m], tiled[3][m], want)
    finally:
        del os.environ["NGSLD_TEST_TILE_MIN_MB"]
        eng.close()
