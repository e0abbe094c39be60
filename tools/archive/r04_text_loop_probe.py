"""Dev probe (GPU box): the text path's loop (ngsld_run with device-side TSV, rows discarded) on configs[2], for a matrix
handed over from HOST memory (what the binary does: the upload runs on the context's copy stream) and from DEVICE memory
(on_device: no upload).  Does the loop's speed depend on what the streams were first used for?
    python tools/r04_text_loop_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ngsld_amd import capi, shard, synth

dev = torch.device("cuda", 0)
n_sites, n_ind = 100_000, 500
raw_t = synth.make_gl_torch(n_sites, n_ind, 3, dev)
raw_h = raw_t.cpu().numpy()
chrs, pos = synth.make_positions(n_sites, 3)
pd = shard.pos_dist_from_positions(chrs, pos)
labels = [f"{c}:{p}" for c, p in zip(chrs, pos)]


def loop(source: str) -> float:
    eng = capi.Engine(0)
    try:
        if source == "host":
            eng.set_geno_raw(raw_h)
        else:
            eng.set_geno_raw(raw_t.data_ptr(), n_sites=n_sites, n_ind=n_ind)
        eng.set_replay_source(raw_h)
        eng.set_pos_dist(pd)
        eng.plan(max_kb_dist=100, extend_out=True)
        eng.set_text_output(labels)
        got = [0]

        def sink(_u, bp):
            got[0] += bp.contents.text_len
            return 0

        cb = capi.SINK_FN(sink)
        ts = []
        for _ in range(3):  # (the first pass pins the text batches' host buffers and sizes the device ones; the binary has a library thread do that ahead: ngsld_reserve_text_buffers)
            got[0] = 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng._check(eng._L.ngsld_run(eng._h, 0, n_sites, cb, None))
            ts.append(time.perf_counter() - t0)
            assert got[0] > 9e9
        return ts
    finally:
        eng.close()


for src in ("host", "device", "host", "device"):
    print(f"matrix from {src:6s} memory: text loop, three passes of one context: " + " / ".join(f"{t:.3f}" for t in loop(src)) + " s", flush=True)
