"""Multi-GPU row sharding on the CPU: the host mirror of the pair-space plan, the pair-count
balanced split, the slab/halo rule, and a world_size-2 gloo run in which each rank computes its own shard
(with the CPU oracle standing in for the device) and the union must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest

from ngsld_amd import shard, synth
from oracle import orc


def _pd(n, seed, n_chr=1, max_gap=200):
    chrs, pos = synth.make_positions(n, seed, max_gap=max_gap, n_chr=n_chr)
    return shard.pos_dist_from_positions(chrs, pos)


@pytest.mark.parametrize("max_kb,max_snp,n_chr", [(0, 0, 1), (2, 0, 1), (3, 0, 3), (0, 9, 2), (5, 15, 2), (1, 0, 1)])
def test_row_end_mirror_matches_oracle_walk(max_kb, max_snp, n_chr):
    n = 400
    pd = _pd(n, 17, n_chr=n_chr)
    raw = synth.make_gl_numpy(n, 4, 17, depth=3.0)
    o = orc.Oracle(raw, pd, max_kb_dist=max_kb, max_snp_dist=max_snp)
    assert np.array_equal(shard.row_ends(pd, max_kb, max_snp), o.row_ends().astype(np.int64))


def test_window_boundary_is_inclusive():
    # dist == max_kb_dist*1000 is kept (ngsLD.cpp:252 breaks only when limit < dist)
    pd = shard.pos_dist_from_positions(["c"] * 4, np.array([1, 1001, 2001, 3002]))
    assert list(shard.row_ends(pd, 1, 0)) == [2, 3, 3, 4]
    assert list(shard.row_ends(pd, 2, 0)) == [3, 3, 4, 4]


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_split_rows_covers_and_balances(world):
    n = 5000
    counts = shard.row_pair_counts(_pd(n, 23), 20, 0)
    parts = shard.split_rows(counts, world)
    assert parts[0][0] == 0 and parts[-1][1] == n and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    per = [int(counts[lo:hi].sum()) for lo, hi in parts]
    assert sum(per) == int(counts.sum())
    assert max(per) - min(per) <= 2 * int(counts.max()) + 1          # within two rows of perfect balance
    counts = shard.row_pair_counts(_pd(300, 29), 0, 0)              # all-pairs: triangular row lengths
    per = [int(counts[lo:hi].sum()) for lo, hi in shard.split_rows(counts, world)]
    assert max(per) - min(per) <= 2 * 300


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_split_rows_weighted_balances_the_weight_not_the_count(world):
    """bench.py --balance work: rows cut by estimated WORK (pairs x executed iterations per pair).  With a cost per pair
    that grows along the rows the cut moves towards the front; the parts cover the rows, and their weights are within two
    rows of equal."""
    n = 4000
    counts = shard.row_pair_counts(_pd(n, 41), 20, 0).astype(np.float64)
    cost = counts * np.linspace(9.0, 14.0, n)                        # later rows iterate longer
    parts = shard.split_rows_weighted(cost, world)
    assert parts[0][0] == 0 and parts[-1][1] == n and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    per = [float(cost[lo:hi].sum()) for lo, hi in parts]
    assert abs(sum(per) - float(cost.sum())) <= 1e-6 * cost.sum()
    assert max(per) - min(per) <= 2 * float(cost.max()) + 1e-9
    if world > 1:
        by_count = shard.split_rows(counts.astype(np.int64), world)
        assert parts[0][1] > by_count[0][1]                         # the first part takes MORE rows: its pairs are cheaper
    assert shard.split_rows_weighted(counts, world) == shard.split_rows(counts.astype(np.int64), world)   # equal weights: the same cut


def test_bench_flags_of_round_4_parse():
    """--balance / --native-multi are part of the driver-facing script: they must parse without a GPU."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--help"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--balance" in r.stdout and "--native-multi" in r.stdout


def _worker(rank, world, port, n_sites, n_ind, max_kb, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pd = _pd(n_sites, 31, n_chr=2)
        raw = torch.from_numpy(synth.make_gl_numpy(n_sites, n_ind, 31, depth=5.0)) if rank == 0 else \
            torch.empty((n_sites, n_ind, 3), dtype=torch.float64)
        shard.broadcast_matrix(raw, src=0)                            # the one collective of the design
        row_end = shard.row_ends(pd, max_kb, 0)
        counts = row_end - (np.arange(n_sites) + 1)
        lo, hi = shard.split_rows(counts, world)[rank]
        slab_lo, slab_hi = shard.slab_for_rows(row_end, lo, hi)
        # rank-local problem: its slab only, local site indices, local pos_dist
        o = orc.Oracle(raw[slab_lo:slab_hi].numpy(), pd[slab_lo:slab_hi].copy(), max_kb_dist=max_kb, n_threads=1)
        rec = o.run(0, hi - lo)
        rec["s1"] += slab_lo
        rec["s2"] += slab_lo
        q.put((rank, lo, hi, rec))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("max_kb", [0, 3])
def test_two_rank_gloo_shards_equal_single_process(max_kb):
    import torch.multiprocessing as mp
    n_sites, n_ind, world = 90, 20, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_sites, n_ind, max_kb, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pd = _pd(n_sites, 31, n_chr=2)
    want = orc.Oracle(synth.make_gl_numpy(n_sites, n_ind, 31, depth=5.0), pd, max_kb_dist=max_kb).run()
    rec = np.concatenate([g[3] for g in got])
    assert got[0][1] == 0 and got[-1][2] == n_sites and got[0][2] == got[1][1]
    assert len(rec) == len(want)
    for col in ("s1", "s2", "dist", "hap", "n_iter", "n_ind_data", "r2", "D", "Dp", "r2pear"):
        assert np.array_equal(rec[col], want[col], equal_nan=True), col


def test_plan_parts_covers_the_rows_and_balances_the_pairs():
    """ngsld_plan_parts (host only): contiguous parts, every row once, halo = the furthest site a row pairs with,
    candidate pairs within a few rows of equal."""
    from ngsld_amd import capi
    n_sites = 5000
    chrs, pos = synth.make_positions(n_sites, 9, max_gap=200, n_chr=3)
    pd = shard.pos_dist_from_positions(chrs, pos)
    for kw in (dict(max_kb_dist=20), dict(max_kb_dist=0), dict(max_snp_dist=50)):
        ends = capi.window_ends(pd if kw.get("max_kb_dist", 0) else pd, n_sites, **kw).astype(np.int64)
        counts = np.maximum(ends - (np.arange(n_sites) + 1), 0)
        for n_parts in (1, 2, 8):
            parts = capi.plan_parts(pd, n_sites, n_parts, **kw)
            assert parts["row_begin"][0] == 0 and parts["row_end"][-1] == n_sites
            assert np.array_equal(parts["row_begin"][1:], parts["row_end"][:-1])
            per = [int(counts[a:b].sum()) for a, b in zip(parts["row_begin"], parts["row_end"])]
            assert sum(per) == int(counts.sum())
            assert max(per) - min(per) <= 2 * int(counts.max()) + 1
            for a, b, e in zip(parts["row_begin"], parts["row_end"], parts["site_end"]):
                assert e == (max(int(ends[a:b].max()), b) if b > a else b)


def test_run_multi_fails_loudly_and_does_not_hang_without_a_device():
    """ngsld_run_multi where ngsld_create fails in every part (no GPU here / a device index that does not exist): the
    parts must all get through their barriers and the call must come back with the library's message."""
    import os
    from ngsld_amd import capi
    raw = synth.make_gl_numpy(40, 6, seed=1)
    devices = [0, 1, 2] if not os.path.exists("/dev/kfd") else [97, 98, 99]
    with pytest.raises(capi.NgsldError) as e:
        capi.run_multi(raw, None, devices, extend_out=True)
    assert e.value.code in (capi.ERR_DEVICE, capi.ERR_INVALID) and "part 0" in e.value.msg
