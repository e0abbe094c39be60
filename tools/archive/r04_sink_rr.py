#!/usr/bin/env python
"""Dev measurement (GPU box): ONE process, several engines over the same matrix that differ in how record batches reach the
host (environment read at ngsld_create), round robin: device-resident step of the first engine, then the host-resident leg of
each -- box drift cancels.  python tools/r04_sink_rr.py [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ngsld_amd import capi, shard, synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_sites, n_ind = 100_000, 500
dev = torch.device("cuda", 0)
raw = synth.make_gl_torch(n_sites, n_ind, 3, dev)
host = raw.cpu().numpy()
chrs, pos = synth.make_positions(n_sites, 3)
pd = shard.pos_dist_from_positions(chrs, pos)
CONFIGS = [("2 streams, 2^22", dict(NGSLD_TEST_RUN_STREAMS="2", NGSLD_TEST_BATCH_PAIRS=str(1 << 22))),
           ("2 streams, 2^21", dict(NGSLD_TEST_RUN_STREAMS="2", NGSLD_TEST_BATCH_PAIRS=str(1 << 21))),
           ("2 streams, 2^22, run 16", dict(NGSLD_TEST_RUN_STREAMS="2", NGSLD_TEST_BATCH_PAIRS=str(1 << 22), NGSLD_RUN_LEN="16")),
           ("2 streams, 2^23, run 8", dict(NGSLD_TEST_RUN_STREAMS="2", NGSLD_RUN_LEN="8")),
           ("2 streams, 2^22, no tail", dict(NGSLD_TEST_RUN_STREAMS="2", NGSLD_TEST_BATCH_PAIRS=str(1 << 22), NGSLD_TEST_TAIL_LEN="0")),
           ("1 stream, 2^22", dict(NGSLD_TEST_RUN_STREAMS="1", NGSLD_TEST_BATCH_PAIRS=str(1 << 22)))]
if len(sys.argv) > 2:
    CONFIGS = eval(open(sys.argv[2]).read())
engines = []
for name, env in CONFIGS:
    for k in ("NGSLD_TEST_RUN_STREAMS", "NGSLD_TEST_BATCH_PAIRS", "NGSLD_TEST_TAIL_PAIRS", "NGSLD_TEST_RUN_DIRECT", "NGSLD_RUN_LEN", "NGSLD_TEST_TAIL_LEN", "NGSLD_HEAD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = capi.Engine(0)
    e.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
    e.set_replay_source(host)
    e.set_pos_dist(pd)
    n = e.plan(max_kb_dist=100, extend_out=True)
    e.run_discard(0, n_sites)
    engines.append((name, e))
d_std = torch.empty(n * 32, dtype=torch.uint8, device=dev)
d_ext = torch.empty(n * 40, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
ref = engines[0][1]
ratios = {name: [] for name, _ in engines}
for r in range(rounds):
    for name, e in engines:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ref.run_device(0, n_sites, d_std.data_ptr(), d_ext.data_ptr(), stream)
        ref.finish_device()
        torch.cuda.synchronize()
        dt_dev = time.perf_counter() - t0
        t0 = time.perf_counter()
        e.run_discard(0, n_sites)
        torch.cuda.synchronize()
        dt_host = time.perf_counter() - t0
        ratios[name].append(dt_dev / dt_host)
        print(f"round {r} {name:30s} device {dt_dev * 1e3:7.2f} ms host {dt_host * 1e3:7.2f} ms ratio {dt_dev / dt_host:.4f}", flush=True)
for name, v in ratios.items():
    print(f"{name:30s} mean ratio {np.mean(v):.4f} min {np.min(v):.4f} max {np.max(v):.4f}")
