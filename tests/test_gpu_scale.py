"""GPU: BASELINE.json's full-size configurations through size-independent properties, plus sampled pairs
checked against the oracle (the oracle cannot run 1e7..1e8 pairs in a test, it can run a few hundred)."""
import numpy as np
import pytest

from ngsld_amd import capi, shard, synth
from oracle import orc
from util import TOL, close

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _run_device(eng, n_rows, n_pairs, dev):
    d_std = torch.empty(max(n_pairs, 1) * 32, dtype=torch.uint8, device=dev)
    d_ext = torch.empty(max(n_pairs, 1) * 40, dtype=torch.uint8, device=dev)
    eng.run_device(0, n_rows, d_std.data_ptr(), d_ext.data_ptr(), None)
    std = d_std.view(torch.float64).view(-1, 4)[:n_pairs]
    ext = d_ext.view(torch.float64).view(-1, 5)[:n_pairs]
    meta = d_ext.view(torch.int32).view(-1, 10)[:n_pairs, 8:10]
    return std, ext[:, :4], meta[:, 0], meta[:, 1]


def _sample_check(raw_t, pos_dist, row_off, row_end, std, hap, n_data, n_iter, n_sample, seed, max_kb, n_rows=None):
    rng = np.random.default_rng(seed)
    n_sites = raw_t.shape[0]
    rows = rng.integers(0, (n_rows or n_sites) - 1, size=n_sample)
    for s1 in rows:
        span = int(row_end[s1]) - (s1 + 1)
        if span <= 0:
            continue
        s2 = s1 + 1 + int(rng.integers(0, span))
        k = int(row_off[s1]) + (s2 - s1 - 1)
        two = raw_t[[s1, s2]].cpu().numpy()
        o = orc.Oracle(two, None)
        r = o.run()[0]
        assert int(n_iter[k]) == r["n_iter"] and int(n_data[k]) == r["n_ind_data"], (s1, s2)
        assert np.all(close(hap[k].cpu().numpy(), r["hap"])) and np.all(
            close(std[k].cpu().numpy(), [r["r2pear"], r["D"], r["Dp"], r["r2"]])), (s1, s2)


def test_c2_all_pairs_5000x100():
    """configs[1]: 5,000 sites x 100 ind, all 12,497,500 pairs."""
    dev = torch.device("cuda", 0)
    n_sites, n_ind = 5000, 100
    raw = synth.make_gl_torch(n_sites, n_ind, 2, dev, depth=10.0)
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
        eng.set_pos_dist(None)
        n = eng.plan(extend_out=True)
        assert n == n_sites * (n_sites - 1) // 2
        row_off, row_end = eng.plan_rows()
        std, hap, n_data, n_iter = _run_device(eng, n_sites, n, dev)
        # properties that hold for every pair
        assert bool(torch.all(n_data == n_ind)) and bool(torch.all((n_iter >= 0) & (n_iter <= 100)))
        assert float((hap.sum(dim=1) - 1).abs().max()) < 1e-12 and float(hap.min()) >= 0.0
        r2 = std[:, 3]
        assert float(r2.min()) >= 0.0 and float(r2.max()) <= 1.0 + 1e-9
        assert float(std[:, 2].abs().max()) <= 1.0 + 1e-9                     # |D'| <= 1
        assert float(std[:, 0].min()) >= 0.0 and float(std[:, 0].max()) <= 1.0 + 1e-12
        # determinism: a second pass is bit-identical (fixed per-pair reduction order)
        std2, hap2, _, it2 = _run_device(eng, n_sites, n, dev)
        assert torch.equal(std.view(torch.int64), std2.view(torch.int64)) and torch.equal(n_iter, it2)
        # allele-flip invariance: swapping genotype 0 <-> 2 at every site leaves r2 and |D| alone, hap00 <-> hap11
        eng.set_geno_raw(raw.flip(2).contiguous().data_ptr(), n_sites=n_sites, n_ind=n_ind)
        eng.set_pos_dist(None)
        eng.plan(extend_out=True)
        std3, hap3, _, it3 = _run_device(eng, n_sites, n, dev)
        same_iter = it3 == n_iter
        assert float(same_iter.double().mean()) > 0.9999                      # threshold flips are ~1e-10 events
        assert float((std3[:, 3] - r2)[same_iter].abs().max()) < TOL
        assert float((std3[:, 1] - std[:, 1])[same_iter].abs().max()) < TOL   # D -> (-)(-)D = D under a double flip
        assert float((hap3[:, 3] - hap[:, 0])[same_iter].abs().max()) < TOL
        eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
        _sample_check(raw, None, row_off, row_end, std, hap, n_data, n_iter, 300, 5, 0)
    finally:
        eng.close()


def test_c3_windowed_100000x500():
    """configs[2]: 100,000 sites x 500 ind, --max_kb_dist 100 (~1e8 pairs), --extend_out records."""
    dev = torch.device("cuda", 0)
    n_sites, n_ind, max_kb = 100_000, 500, 100
    chrs, pos = synth.make_positions(n_sites, 3)
    pd = shard.pos_dist_from_positions(chrs, pos)
    raw = synth.make_gl_torch(n_sites, n_ind, 3, dev, depth=10.0)
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
        eng.set_pos_dist(pd)
        n = eng.plan(max_kb_dist=max_kb, extend_out=True)
        row_off, row_end = eng.plan_rows()
        assert np.array_equal(row_end.astype(np.int64), shard.row_ends(pd, max_kb, 0))   # host mirror == engine
        assert n == int(row_off[-1]) and 9.0e7 < n < 1.1e8
        std, hap, n_data, n_iter = _run_device(eng, n_sites, n, dev)
        assert bool(torch.all(n_data == n_ind))
        assert float((hap.sum(dim=1) - 1).abs().max()) < 1e-12 and float(hap.min()) >= 0.0
        assert float(std[:, 3].min()) >= 0.0 and float(std[:, 3].max()) <= 1.0 + 1e-9
        # sharding invariance: rows [lo, hi) computed from a slab (local indices) give the same records
        lo, hi = 40_000, 40_400
        slab_lo, slab_hi = shard.slab_for_rows(shard.row_ends(pd, max_kb, 0), lo, hi)
        eng2 = capi.Engine(0)
        try:
            eng2.set_geno_raw(raw[slab_lo:slab_hi].data_ptr(), n_sites=slab_hi - slab_lo, n_ind=n_ind)
            eng2.set_pos_dist(pd[slab_lo:slab_hi].copy())
            eng2.plan(max_kb_dist=max_kb, extend_out=True)
            ro2, _ = eng2.plan_rows()
            m = int(ro2[hi - lo])
            std_s, hap_s, _, it_s = _run_device(eng2, hi - lo, m, dev)
            a, b = int(row_off[lo]), int(row_off[hi])
            assert b - a == m
            assert torch.equal(std_s.view(torch.int64), std[a:b].view(torch.int64)) and torch.equal(it_s, n_iter[a:b])
        finally:
            eng2.close()
        _sample_check(raw, pd, row_off, row_end, std, hap, n_data, n_iter, 200, 7, max_kb)
    finally:
        eng.close()


def test_c4_all_pairs_50000x1000_in_rank_shards():
    """configs[3]: 50,000 sites x 1,000 ind, all 1,249,975,000 pairs, computed as the 8 row shards of an 8-GPU run
    (shard.split_rows: equal pair counts), one after the other on this GPU; 2 wavefronts per pair."""
    dev = torch.device("cuda", 0)
    n_sites, n_ind, world = 50_000, 1000, 8
    raw = synth.make_gl_torch(n_sites, n_ind, 4, dev, depth=10.0)
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
        eng.set_pos_dist(None)
        n = eng.plan(extend_out=True)
        assert n == n_sites * (n_sites - 1) // 2 == 1_249_975_000
        row_off, row_end = eng.plan_rows()
        bounds = shard.split_rows(np.diff(row_off.astype(np.int64)), world)
        assert bounds[0][0] == 0 and bounds[-1][1] == n_sites
        total, it_sum = 0, 0
        for rank, (lo, hi) in enumerate(bounds):
            m = int(row_off[hi] - row_off[lo])
            assert abs(m - n / world) < 2 * n_sites                           # balanced to within a row or two
            d_std = torch.empty(m * 32, dtype=torch.uint8, device=dev)
            d_ext = torch.empty(m * 40, dtype=torch.uint8, device=dev)
            eng.run_device(lo, hi, d_std.data_ptr(), d_ext.data_ptr(), None)
            std = d_std.view(torch.float64).view(-1, 4)
            hap = d_ext.view(torch.float64).view(-1, 5)[:, :4]
            meta = d_ext.view(torch.int32).view(-1, 10)[:, 8:10]
            assert bool(torch.all(meta[:, 0] == n_ind)) and bool(torch.all((meta[:, 1] >= 0) & (meta[:, 1] <= 100)))
            assert float((hap.sum(dim=1) - 1).abs().max()) < 1e-12 and float(hap.min()) >= 0.0
            assert float(std[:, 3].min()) >= 0.0 and float(std[:, 3].max()) <= 1.0 + 1e-9
            assert float(std[:, 2].abs().max()) <= 1.0 + 1e-9
            total += m
            it_sum += int(meta[:, 1].sum(dtype=torch.int64))
            if rank in (0, 5):                                                 # sampled pairs of this shard vs the oracle
                rng = np.random.default_rng(rank)
                for s1 in rng.integers(lo, hi, size=40):
                    s2 = int(rng.integers(s1 + 1, n_sites)) if s1 + 1 < n_sites else None
                    if s2 is None:
                        continue
                    k = int(row_off[s1] - row_off[lo]) + (s2 - int(s1) - 1)
                    r = orc.Oracle(raw[[int(s1), s2]].cpu().numpy(), None).run()[0]
                    assert int(meta[k, 1]) == r["n_iter"], (s1, s2)
                    assert np.all(close(hap[k].cpu().numpy(), r["hap"])) and np.all(
                        close(std[k].cpu().numpy(), [r["r2pear"], r["D"], r["Dp"], r["r2"]])), (s1, s2)
            del std, hap, meta, d_std, d_ext
        assert total == n
        assert 8.0 < it_sum / n < 12.0                                         # depth-10 data: ~9.5 iterations per pair
    finally:
        eng.close()


def test_c5_rank_slab_125000x2000_windowed():
    """configs[4]: 1,000,000 sites x 2,000 ind, --max_kb_dist 500 over ~1 kb gaps, 8 ranks: one rank's slab
    (125,000 rows + halo, 6 GB of GLs, 4 wavefronts per pair) at full size."""
    dev = torch.device("cuda", 0)
    n_sites, n_ind, max_kb = 125_600, 2000, 500
    chrs, pos = synth.make_positions(n_sites, 5, max_gap=2000)
    pd = shard.pos_dist_from_positions(chrs, pos)
    ends = shard.row_ends(pd, max_kb, 0)
    n_rows = 125_000
    assert int(ends[:n_rows].max()) <= n_sites                                # the halo covers the last row's window
    raw = synth.make_gl_torch(n_sites, n_ind, 5, dev, depth=10.0)
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
        eng.set_pos_dist(pd)
        eng.plan(max_kb_dist=max_kb, extend_out=True)
        row_off, row_end = eng.plan_rows()
        assert np.array_equal(row_end.astype(np.int64), ends)
        m = int(row_off[n_rows])
        assert 5.5e7 < m < 7.0e7                                               # ~500 partners per site
        std, hap, n_data, n_iter = _run_device(eng, n_rows, m, dev)
        assert bool(torch.all(n_data == n_ind)) and bool(torch.all((n_iter >= 0) & (n_iter <= 100)))
        assert float((hap.sum(dim=1) - 1).abs().max()) < 1e-12 and float(hap.min()) >= 0.0
        assert float(std[:, 3].min()) >= 0.0 and float(std[:, 3].max()) <= 1.0 + 1e-9
        assert float(std[:, 0].min()) >= 0.0 and float(std[:, 0].max()) <= 1.0 + 1e-12
        _sample_check(raw, pd, row_off, row_end, std, hap, n_data, n_iter, 60, 9, max_kb, n_rows=n_rows)
    finally:
        eng.close()


@pytest.mark.timeout(1500)
def test_c5_full_size_on_one_gpu_through_the_bench():
    """configs[4] at FULL size -- 1,000,000 sites x 2,000 individuals, 500 kb window, the 48 GB matrix resident in HBM -- as
    one `bench.py --config c4` step on this box's GPU (what rank 0 of an 8-GPU run computes is an eighth of it).  The bench's
    own checks are the test: its host mirror of the window walk and the engine's plan agree on the pair count (asserted
    inside), and the cpu_baseline leg re-computes the matrix's first rows with the oracle and compares them with the GPU's
    records of the same rows -- pairs and executed EM iterations EQUAL, sum of r2 to 1e-9 relative."""
    import json
    import subprocess
    import sys
    free, total = capi.device_memory(0) if hasattr(capi, "device_memory") else (None, None)
    if free is not None and free < 150 * 2 ** 30:
        pytest.skip(f"{free / 2 ** 30:.0f} GiB of device memory free: the full-size run needs ~135 GiB")
    cmd = [sys.executable, "bench.py", "--config", "c4", "--steps", "1", "--warmup", "0", "--no-traffic", "--no-e2e", "--no-sink",
           "--cpu-seconds", "4"]
    r = subprocess.run(cmd, cwd=capi.REPO_DIR, capture_output=True, text=True, timeout=1400)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = d["config"]
    assert c["n_sites_total"] == 1_000_000 and "2000 ind" in c["workload"] and "BASELINE configs[4]" in c["workload"]
    assert 4.5e8 < c["pairs_per_step"] < 5.5e8                                 # ~500 partners per site
    assert 8.5 < c["mean_executed_em_iterations"] < 10.5
    assert d["roofline"]["kernel"].startswith("pair_ld_kernel") and d["value"] > 2.5e7
    rec = c["rank_records"][0]
    assert rec["pairs"] == c["pairs_per_step"] and rec["sites_held"] == [0, 1_000_000]
    par = d["cpu_baseline"]["parity_on_sample"]
    assert par["pairs_equal"] and par["executed_iterations_equal"] and par["pairs"] > 20_000, par
    assert par["abs_diff_sum_r2"] <= 1e-9 * max(1.0, abs(par["sum_r2_cpu"])), par
    print(f"\n[c5 full size] {c['pairs_per_step']} pairs, {d['value']:.4g} pairs/s, frac {d['roofline']['frac']:.3f}; "
          f"parity sample {par['pairs']} pairs: iterations equal, |d sum r2| {par['abs_diff_sum_r2']:.2e}")
