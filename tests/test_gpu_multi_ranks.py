"""GPU: the N > 1 paths, exercised the way the driver launches them.

(a) `bench.py --gpus 2` under `python -m torch.distributed.run --nproc-per-node 2` -- the driver's own command line.  A
    one-GPU box runs it with NGSLD_BENCH_ONE_DEVICE=1 (both ranks on GPU 0, gloo instead of RCCL: the rank bookkeeping,
    the row shards, the per-rank slabs and the aggregation are the same code); a box with >= 2 GPUs runs it as the driver
    does, one rank per GPU over RCCL.  The two ranks' records must add up to the 1-rank run of the same matrix: pair
    counts, executed-iteration totals and the wrap-around checksum of every record word are partition-invariant, so they
    are compared for EQUALITY -- sharding changes who computes a pair, never its result (ngsLD.cpp:153-198: the
    reference's per-s1 jobs are independent in the same way).
(b) `ngsld_run_multi` over DISTINCT devices: the all-pairs matrix must travel by ONE ncclBroadcast over RCCL / xGMI
    (ngsld_multi_last_distribution) and the parts must concatenate to the single-device run byte for byte.  Skipped, with
    the reason, where the box has one GPU (tests/test_gpu_multi_native.py covers the same code on one device).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from ngsld_amd import capi, shard, synth

pytestmark = pytest.mark.gpu

REPO = capi.REPO_DIR
SITES_PER_RANK, N_IND = 6000, 500


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(n_ranks: int, sites: int, one_device: bool, extra: tuple = ()) -> dict:
    """One bench.py job exactly as the driver starts it (N = 1: plain python; N > 1: torch.distributed.run)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.pop("NGSLD_BENCH_ONE_DEVICE", None)
    if one_device:
        env["NGSLD_BENCH_ONE_DEVICE"] = "1"
    args = ["bench.py", "--gpus", str(n_ranks), "--config", "c2", "--sites", str(sites), "--ind", str(N_IND), "--steps", "2",
            "--warmup", "1", "--no-e2e", "--no-cpu", "--no-traffic", *extra]
    if n_ranks == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"rank 0 must print exactly ONE JSON line, got {len(lines)}:\n{r.stdout[-2000:]}"
    return json.loads(lines[0])


@pytest.mark.timeout(1800)
def test_bench_two_ranks_add_up_to_the_one_rank_run():
    two_gpus = capi.device_count() >= 2
    one = _bench(1, 2 * SITES_PER_RANK, one_device=False)
    two = _bench(2, SITES_PER_RANK, one_device=not two_gpus)

    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "weak" and two["steps"] == 2 and two["warmup"] == 1
    assert two["config"]["n_sites_total"] == one["config"]["n_sites_total"] == 2 * SITES_PER_RANK
    c1, c2 = one["config"], two["config"]
    # the pair space is split, not duplicated: per-rank counts add up to the single-rank plan
    lo, hi = c2["pairs_per_rank_min_max"]
    assert lo + hi == c2["pairs_per_step"] == c1["pairs_per_step"]
    assert hi - lo <= 0.02 * hi, "ranks are balanced by pair count"
    tmin, tmax = c2["rank_seconds_min_max"]
    assert 0 < tmin <= tmax
    assert abs(two["ms_per_step"] - tmax / two["steps"] * 1e3) < 1e-6 * tmax * 1e3 + 1e-9, "the step time is the MAX over ranks"
    assert abs(two["value"] - c2["pairs_per_step"] * two["steps"] / tmax) <= 1e-9 * two["value"]

    recs = sorted(c2["rank_records"], key=lambda r: r["rank"])
    assert [r["rank"] for r in recs] == [0, 1]
    assert recs[0]["rows"][0] == 0 and recs[0]["rows"][1] == recs[1]["rows"][0] and recs[1]["rows"][1] == 2 * SITES_PER_RANK
    assert sorted(r["pairs"] for r in recs) == [lo, hi]
    # a rank holds its rows plus the halo their windows reach into -- not the whole matrix
    assert recs[0]["sites_held"][1] < 2 * SITES_PER_RANK and recs[1]["sites_held"][0] == recs[1]["rows"][0]
    whole = c1["rank_records"][0]
    assert sum(r["executed_iterations"] for r in recs) == whole["executed_iterations"]
    assert sum(r["records_checksum_u64"] for r in recs) % (1 << 64) == whole["records_checksum_u64"], \
        "the two ranks' records are not the 1-rank run's records"
    assert abs(sum(r["sum_r2_finite"] for r in recs) - whole["sum_r2_finite"]) <= 1e-9 * abs(whole["sum_r2_finite"])
    if two_gpus:
        assert {r["device_index"] for r in recs} == {0, 1} and "nccl" in c2["backend"]
    else:
        assert "gloo" in c2["backend"]
    # the record checks itself: an all_reduce of ones over the job's backend saw both ranks; every rank reports its broadcast
    assert c2["rccl_ranks_seen"] == 2 and c1["rccl_ranks_seen"] == 1
    assert all(r["gl_broadcast_s"] >= 0 for r in recs)
    print(f"\n[multi-ranks] {'RCCL, 2 GPUs' if two_gpus else 'one device, gloo dry run'}: {c2['pairs_per_step']} pairs, "
          f"ranks {lo} / {hi}, {tmin:.3f} / {tmax:.3f} s, checksum {whole['records_checksum_u64']:#018x}")


@pytest.mark.timeout(1800)
def test_bench_two_ranks_balanced_by_estimated_work_add_up_too():
    """--balance work: rank 0 measures every ~100th row, the rows are cut by estimated work (pairs x executed iterations)
    instead of pair count -- another partition of the same pair space, so the ranks' records still SUM to the 1-rank run."""
    two_gpus = capi.device_count() >= 2
    one = _bench(1, 2 * SITES_PER_RANK, one_device=False)
    two = _bench(2, SITES_PER_RANK, one_device=not two_gpus, extra=("--balance", "work"))
    c1, c2 = one["config"], two["config"]
    assert c2["balance"] == "work" and c2["balance_estimate"]["sampled_rows"] >= 64 and "estimated work" in c2["parallelism"]
    recs = sorted(c2["rank_records"], key=lambda r: r["rank"])
    assert recs[0]["rows"][0] == 0 and recs[0]["rows"][1] == recs[1]["rows"][0] and recs[1]["rows"][1] == 2 * SITES_PER_RANK
    whole = c1["rank_records"][0]
    assert sum(r["pairs"] for r in recs) == whole["pairs"] == c1["pairs_per_step"]
    assert sum(r["executed_iterations"] for r in recs) == whole["executed_iterations"]
    assert sum(r["records_checksum_u64"] for r in recs) % (1 << 64) == whole["records_checksum_u64"]
    # the estimate does its job: the ranks' executed-iteration totals are within 2 % of each other
    it = [r["executed_iterations"] for r in recs]
    assert abs(it[0] - it[1]) <= 0.02 * max(it), it
    assert all("mean_executed_iterations" in r and r["kernel_ms_per_launch"] > 0 for r in recs)


@pytest.mark.timeout(1800)
def test_bench_native_multi_parts_add_up_to_one_part():
    """bench.py --native-multi: the product's own one-process multi-device path (ngsld_run_multi) through the same bench.
    Three parts (on the devices there are: all on GPU 0 on a one-GPU box) against one part of the same 12,000 sites: pair
    counts, executed-iteration totals and record checksums are EQUAL."""
    def run(parts, sites):
        cmd = [sys.executable, "bench.py", "--native-multi", "--gpus", str(parts), "--config", "c2", "--sites", str(sites),
               "--ind", str(N_IND), "--steps", "2", "--warmup", "1"]
        r = subprocess.run(cmd, cwd=REPO, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])
    one = run(1, 12_000)
    three = run(3, 4_000)
    a, b = one["config"], three["config"]
    assert three["n_gpus"] == 3 and len(b["pairs_per_part"]) == 3 and all(n > 0 for n in b["pairs_per_part"])
    assert "ngsld_run_multi" in b["parallelism"] and b["matrix_distribution"] in ("upload", "peer_copy", "rccl")
    assert a["n_sites_total"] == b["n_sites_total"] == 12_000
    assert sum(b["pairs_per_part"]) == b["pairs_per_step"] == a["pairs_per_step"]
    assert b["executed_iterations_total"] == a["executed_iterations_total"]
    assert b["records_checksum_u64"] == a["records_checksum_u64"]
    assert max(b["pairs_per_part"]) - min(b["pairs_per_part"]) <= 0.02 * max(b["pairs_per_part"])
    assert three["value"] > 0 and abs(three["value"] - b["pairs_per_step"] * 2 / sum(b["step_seconds"])) <= 2e-3 * three["value"]


def test_run_multi_on_distinct_devices_broadcasts_over_rccl():
    n_dev = capi.device_count()
    if n_dev < 2:
        pytest.skip(f"hipGetDeviceCount() = {n_dev}: the RCCL broadcast needs two distinct devices "
                    "(the same code on one device: tests/test_gpu_multi_native.py)")
    n_sites, n_ind = 2000, 300
    raw = synth.make_gl_numpy(n_sites, n_ind, seed=61, depth=5.0)
    raw[7] = [1.0, 0.0, 0.0]                                   # a degenerate site: its pairs are replayed on every device
    chrs, pos = synth.make_positions(n_sites, 61, max_gap=200, n_chr=2)
    pd = shard.pos_dist_from_positions(chrs, pos)
    devices = list(range(min(n_dev, 8)))
    for kw, expect in ((dict(extend_out=True), "rccl"), (dict(extend_out=True, max_kb_dist=20), "upload")):
        eng = capi.Engine(0)
        try:
            eng.set_geno_raw(raw)
            eng.set_pos_dist(pd)
            eng.plan(**kw)
            s1, s2, std, ext = eng.run()
        finally:
            eng.close()
        parts, _, per = capi.run_multi(raw, pd, devices, **kw)
        assert capi.multi_last_distribution() == expect
        got = [np.concatenate([p[k] for p in parts]) for k in range(4)]
        assert sum(per) == len(s1) and all(n > 0 for n in per)
        assert np.array_equal(got[0], s1) and np.array_equal(got[1], s2)
        assert got[2].tobytes() == std.tobytes() and got[3].tobytes() == ext.tobytes()


def test_rccl_calls_of_run_multi_on_a_one_device_communicator():
    """What one GPU can prove about the broadcast of ngsld_run_multi: librccl loads, ncclCommInitAll / ncclGroupStart /
    ncclBroadcast / ncclGroupEnd / ncclCommDestroy run on a communicator of one device, 64 MiB come back unchanged."""
    capi.rccl_selftest(0, 64 << 20)


RCCL_ONE_RANK = r"""
import os, sys
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from ngsld_amd import shard
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # the backend bench.py --gpus N uses (RCCL)
x = torch.arange(1 << 22, dtype=torch.float64, device=dev)
dist.broadcast(x, src=0)                                                  # bench.py: shard.broadcast_matrix
s = torch.tensor([3.0, -1.0], dtype=torch.float64, device=dev)
dist.all_reduce(s, op=dist.ReduceOp.MAX)                                  # bench.py: MAX / SUM over ranks
got = [None]
dist.all_gather_object(got, {"rank": 0})                                  # bench.py: rank_records
dist.barrier()
torch.cuda.synchronize()
assert float(x[12345]) == 12345.0 and float(s[0]) == 3.0 and got == [{"rank": 0}]
dist.destroy_process_group()
print("rccl one-rank ok")
"""


def test_torch_rccl_backend_runs_the_collectives_bench_uses():
    """bench.py --gpus N talks RCCL through torch.distributed's nccl backend: a one-rank group on this box runs the three
    collectives it uses (broadcast of the matrix, all_reduce of the timings, all_gather_object of the rank records)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", RCCL_ONE_RANK, REPO], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_bench_refuses_more_ranks_than_devices():
    """`--gpus N` with fewer visible devices than ranks is an error (a dry run on one device has to be asked for by name)."""
    if capi.device_count() >= 2:
        pytest.skip("two devices here: nothing to refuse")
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("NGSLD_BENCH_ONE_DEVICE", None)
    env["MASTER_ADDR"] = "127.0.0.1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--sites", "2000", "--no-cpu", "--no-sink", "--no-e2e", "--no-traffic"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert "has no device of its own" in (r.stderr + r.stdout)
