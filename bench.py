#!/usr/bin/env python
"""bench.py -- SNP-pair EM-LD computations per second on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2], the n_ind = 500 configuration the metric is quoted on):
synthetic binary GL, 100,000 sites x 500 individuals PER GPU, depth-10 generator, positions with gaps
~ UniformInt[1,200] on one chromosome, --max_kb_dist 100 windowed, --extend_out records.  A "step" is one
pass of the pair kernel over every pair of the rank's rows, inputs already resident in HBM, results
left in HBM (ngsld_run_device).  N GPUs: the site axis is N x 100,000 long (weak scaling), rank 0
generates the matrix and broadcasts it once over RCCL (outside the timed region), ranks take contiguous
row ranges balanced by pair count and never communicate on the compute path.

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
  roofline      HBM roofline of the pair kernel on ALGORITHMIC bytes: (48*n_ind + 72) B per pair
                (both sites' GL vectors + the 32 B standard and 40 B extended record) / kernel time from
                HIP events on the launch stream, against 8 TB/s.  Also carries the FP64-VALU view, which
                is the bound that actually binds this kernel.
  cpu_baseline  the CPU oracle (bit-checked restatement of the reference) on all host cores, on a
                bounded sample of the same workload (first rows of the same matrix).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X FP64 vector peak
STD_BYTES, EXT_BYTES = 32, 40


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sites", type=int, default=100_000, help="sites per GPU")
    ap.add_argument("--ind", type=int, default=500)
    ap.add_argument("--max-kb", type=int, default=100)
    ap.add_argument("--depth", type=float, default=10.0)
    ap.add_argument("--max-gap", type=int, default=200, help="site gaps ~ UniformInt[1, max-gap] (SURVEY 8d: 2000 for configs[4])")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--pairs-per-item", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the cpu_baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--ignore-miss", action="store_true", help="run the --ignore_miss_data kernels (not the headline config)")
    ap.add_argument("--rnd-sample", type=float, default=1.0, help="--rnd_sample of the plan (not the headline config)")
    ap.add_argument("--sink", action="store_true", help="also time ngsld_run (records copied to pinned host memory)")
    ap.add_argument("--hard-calls", action="store_true",
                    help="hard-call the synthetic likelihoods (argmax -> 1/0/0 triples): the genotype-combination kernel (not the headline config)")
    ap.add_argument("--traffic-json", default=os.path.join(REPO, "profiles", "hbm_traffic.json"))
    return ap.parse_args()


def cpu_baseline(raw_head: np.ndarray, pos_dist_head: np.ndarray, max_kb: int, target_s: float, gpu_rows=None) -> dict:
    """Oracle (kind 'port') on every host core, rows [0, R) of the bench matrix with R sized for ~target_s.
    gpu_rows(R) -> (#pairs, sum of finite r2, sum of executed EM iterations) of the GPU records of the same rows:
    a whole-sample parity check (the iteration totals must be EQUAL, i.e. no convergence-threshold flip)."""
    from oracle import orc
    cores = os.cpu_count() or 1
    n_have = raw_head.shape[0]
    o = orc.Oracle(raw_head, pos_dist_head, max_kb_dist=max_kb, n_threads=cores)
    ends = o.row_ends().astype(np.int64)
    halo = int((ends - np.arange(n_have)).max())
    usable = max(1, n_have - halo)                      # rows whose whole window lies inside the sample
    cal_rows = min(usable, max(cores, 64))
    t0 = time.perf_counter()
    n_cal, _, _ = o.bench(0, cal_rows)
    t_cal = max(time.perf_counter() - t0, 1e-6)
    rows = int(min(usable, max(cal_rows, cal_rows * target_s / t_cal)))
    t0 = time.perf_counter()
    n, chk, iters = o.bench(0, rows)
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": f"rows 0..{rows} of the same matrix ({n} pairs, {dt:.1f} s, mean executed EM iterations "
                     f"{iters / max(n, 1):.2f}); oracle/liborc.so, {cores} pthreads"}
    if gpu_rows is not None:
        gn, gsum, giters = gpu_rows(rows)
        out["parity_on_sample"] = {"pairs": int(gn), "pairs_equal": bool(gn == n),
                                   "executed_iterations_cpu": int(iters), "executed_iterations_gpu": int(giters),
                                   "executed_iterations_equal": bool(giters == iters),
                                   "abs_diff_sum_r2": abs(gsum - chk), "sum_r2_cpu": chk}
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    from ngsld_amd import capi, shard, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda is not available (no CPU fallback)")
    # NGSLD_BENCH_ONE_DEVICE=1 (debug only): put every rank on GPU 0 with the gloo backend, so the N > 1 code
    # path can be exercised on a 1-GPU box; the driver's runs use one GPU per rank over RCCL (backend nccl).
    one_dev = os.environ.get("NGSLD_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_dev else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    n_sites = args.sites * world
    n_ind = args.ind

    # ---- positions (host, identical on every rank) and the row shards ----
    chrs, pos = synth.make_positions(n_sites, args.seed, max_gap=args.max_gap, n_chr=1)
    pos_dist = shard.pos_dist_from_positions(chrs, pos)
    row_end = shard.row_ends(pos_dist, args.max_kb, 0)
    counts = row_end - (np.arange(n_sites, dtype=np.int64) + 1)
    lo, hi = shard.split_rows(counts, world)[rank]
    slab_lo, slab_hi = shard.slab_for_rows(row_end, lo, hi)

    # ---- the GL matrix: rank 0 generates, one RCCL broadcast distributes (not timed) ----
    t_gen = time.perf_counter()
    if rank == 0:
        raw = synth.make_gl_torch(n_sites, n_ind, args.seed, dev, depth=args.depth)
    else:
        raw = torch.empty((n_sites, n_ind, 3), dtype=torch.float64, device=dev)
    if args.hard_calls and rank == 0:
        raw = torch.nn.functional.one_hot(raw.argmax(dim=2), 3).to(torch.float64).contiguous()
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    t_bc = time.perf_counter()
    shard.broadcast_matrix(raw, src=0)
    torch.cuda.synchronize()
    t_bc = time.perf_counter() - t_bc

    # ---- engine: this rank's slab (its rows + halo) goes through the device prep kernel ----
    eng = capi.Engine(dev_index)
    slab = raw[slab_lo:slab_hi]
    torch.cuda.synchronize()
    t_prep = time.perf_counter()
    eng.set_geno_raw(slab.data_ptr(), n_sites=slab_hi - slab_lo, n_ind=n_ind,
                     ignore_miss_data=args.ignore_miss)                          # per-site prep kernel (one-off)
    t_prep = time.perf_counter() - t_prep
    local_pd = pos_dist[slab_lo:slab_hi].copy()
    eng.set_pos_dist(local_pd)
    if args.pairs_per_item:
        eng.set_tuning(pairs_per_item=args.pairs_per_item)
    t_plan = time.perf_counter()
    eng.plan(max_kb_dist=args.max_kb, extend_out=True, ignore_miss_data=args.ignore_miss,
             rnd_sample=args.rnd_sample, seed=12345)                                # pair-space plan (one-off)
    t_plan = time.perf_counter() - t_plan
    row_off, _ = eng.plan_rows()
    n_rows = hi - lo
    n_pairs = int(row_off[n_rows] - row_off[0])
    assert args.rnd_sample < 1.0 or n_pairs == int(counts[lo:hi].sum()), "engine plan and host mirror disagree on the pair count"

    raw_head = None
    family = eng.pair_kernel()
    if rank == 0 and world == 1 and not args.no_cpu and not args.ignore_miss and args.rnd_sample >= 1.0:
        head = min(n_sites, 12_000)
        raw_head = raw[:head].cpu().numpy()
    del slab, raw
    torch.cuda.empty_cache()

    d_std = torch.empty(max(n_pairs, 1) * STD_BYTES, dtype=torch.uint8, device=dev)
    d_ext = torch.empty(max(n_pairs, 1) * EXT_BYTES, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.run_device(0, n_rows, d_std.data_ptr(), d_ext.data_ptr(), stream)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    kernel_ms, launches = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        ms, nl, _ = eng.last_kernel_time()   # HIP events on the launch stream (waits for that step's kernels)
        kernel_ms += ms
        launches += nl
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    sink_rate = None
    if args.sink:                       # PCIe-inclusive hand-off; reported beside, never as `value`
        eng.run_discard(0, n_rows)
        t1 = time.perf_counter()
        got = eng.run_discard(0, n_rows)
        sink_rate = got / (time.perf_counter() - t1)
        assert got == n_pairs

    # ---- aggregate over ranks: MAX time, SUM pairs ----
    stats = torch.tensor([elapsed, kernel_ms / max(launches, 1), float(n_pairs)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed_max, total_pairs = float(mx[0]), int(sm[2])
    else:
        elapsed_max, total_pairs = elapsed, n_pairs

    # mean executed EM iterations (n_iter is the 0-based index of the converging iteration; 100 = cap)
    ext_i32 = d_ext.view(torch.int32).view(-1, EXT_BYTES // 4)
    n_iter = ext_i32[:n_pairs, 9].to(torch.float64)
    mean_exec = float(torch.clamp(n_iter + 1, max=100).mean()) if n_pairs else 0.0

    if rank == 0:
        value = total_pairs * args.steps / elapsed_max
        bytes_pair = 48 * n_ind + STD_BYTES + EXT_BYTES
        launch_s = (kernel_ms / max(launches, 1)) / 1e3
        pairs_per_launch = n_pairs * args.steps / max(launches, 1)
        achieved = bytes_pair * pairs_per_launch / launch_s / 1e9
        # FP64 view: per individual and executed iteration 9 FMA (s) + 8 FMA (R) + 3 of the shared-reciprocal tree
        dp_ops = pairs_per_launch * n_ind * mean_exec * 20.0
        fp64_tflops = 2.0 * dp_ops / launch_s / 1e12
        traffic = None
        try:
            with open(args.traffic_json) as fh:
                tj = json.load(fh)
            if tj.get("workload") == f"{args.sites}x{n_ind}@{args.max_kb}kb":
                traffic = tj.get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        out = {
            "metric": "SNP-pair EM-LD computations/sec @ n_ind=500", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic binary GL, {args.sites} sites/GPU x {n_ind} ind, depth {args.depth:g}, "
                                   f"--max_kb_dist {args.max_kb} {'windowed' if args.max_kb else 'all pairs'}, --extend_out"
                                   + (", hard-called" if args.hard_calls else "")
                                   + (" (BASELINE configs[2])" if (args.sites, n_ind, args.max_kb, args.max_gap, args.hard_calls) == (100_000, 500, 100, 200, False) else ""),
                       "n_sites_total": n_sites, "pairs_per_step": total_pairs,
                       "mean_executed_em_iterations": round(mean_exec, 3),
                       "parallelism": f"rows sharded by pair count over {world} GPU(s), no data-path collective",
                       "gl_generate_s": round(t_gen, 3), "gl_broadcast_s": round(t_bc, 3),
                       "one_off_prep_ms_rank0": round(t_prep * 1e3, 2), "one_off_plan_ms_rank0": round(t_plan * 1e3, 2),
                       "host_handoff_pairs_per_s_rank0": sink_rate},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": {"group": "pair_ld_group_kernel", "run": "pair_ld_run_kernel", "wave": "pair_ld_pf_kernel",
                                    "multi": "pair_ld_kernel (multi-wavefront)", "stream": "pair_ld_stream_kernel",
                                    "direct": "pair_ld_kernel (no prefetch)",
                                    "hard": "pair_ld_hard_kernel (genotype-combination counts)"}.get(family, family),
                         "kernel_ms_per_launch": launch_s * 1e3,
                         "algorithmic_bytes_per_pair": bytes_pair,
                         "fp64_valu": {"achieved": fp64_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": fp64_tflops / FP64_PEAK_TFLOPS,
                                       "note": "20 f64 VALU ops per individual per executed EM iteration "
                                               "(FMA counted as 2 flop); the binding roofline"}},
        }
        if raw_head is not None:
            def gpu_rows(rows):
                k = int(row_off[rows] - row_off[0])
                r2 = d_std.view(torch.float64).view(-1, 4)[:k, 3]
                it = ext_i32[:k, 9].to(torch.int64)
                fin = torch.isfinite(r2)
                return k, float(r2[fin].sum()), int(torch.clamp(it + 1, max=100).sum())
            out["cpu_baseline"] = cpu_baseline(raw_head, pos_dist[:raw_head.shape[0]].copy(), args.max_kb,
                                               args.cpu_seconds, gpu_rows)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
