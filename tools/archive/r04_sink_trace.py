#!/usr/bin/env python
"""Dev measurement (GPU box): in ONE process, alternate the device-resident step (ngsld_run_device: one launch) and the
host-resident leg (ngsld_run with a discarding sink: a launch per batch) on configs[2], so that box drift cancels.
python tools/r04_sink_trace.py [n_sites] [n_ind] [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ngsld_amd import capi, shard, synth  # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
n_ind = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
raw = synth.make_gl_torch(n_sites, n_ind, 3, dev)
chrs, pos = synth.make_positions(n_sites, 3)
pd = shard.pos_dist_from_positions(chrs, pos)
eng = capi.Engine(0)
eng.set_geno_raw(raw.data_ptr(), n_sites=n_sites, n_ind=n_ind)
host = raw.cpu().numpy()
eng.set_replay_source(host)
eng.set_pos_dist(pd)
n = eng.plan(max_kb_dist=100, extend_out=True)
d_std = torch.empty(n * 32, dtype=torch.uint8, device=dev)
d_ext = torch.empty(n * 40, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
eng.run_discard(0, n_sites)
for k in range(rounds):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_device(0, n_sites, d_std.data_ptr(), d_ext.data_ptr(), stream)
    eng.finish_device()
    torch.cuda.synchronize()
    dt_dev = time.perf_counter() - t0
    ms_dev, nl_dev, _ = eng.last_kernel_time()
    t0 = time.perf_counter()
    got = eng.run_discard(0, n_sites)
    torch.cuda.synchronize()
    dt_host = time.perf_counter() - t0
    ms_host, nl_host, _ = eng.last_kernel_time()
    print(f"round {k}: device-resident {dt_dev * 1e3:.2f} ms (kernel {ms_dev:.2f}, {nl_dev} launch) | host-resident {dt_host * 1e3:.2f} ms "
          f"(kernels {ms_host:.2f} in {nl_host} launches) | ratio {dt_dev / dt_host:.4f}", flush=True)
eng.close()
