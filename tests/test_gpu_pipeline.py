"""GPU: how record batches reach the host (ngsld_run without text output; engine.hip).  Whatever the route --
the pair kernels writing into the pinned host buffers themselves (default) or device buffers + a D2H copy per batch, one
compute stream or two half a batch out of phase, batches tapered or not, the launches' last rows cut into short runs or not,
many small batches or few large ones -- the sink must see the SAME records, byte for byte, in the same order, and they are
the records ngsld_run_device leaves on the device."""
import numpy as np
import pytest

from ngsld_amd import capi, shard, synth

pytestmark = pytest.mark.gpu

VARIANTS = [
    {},                                                        # default: direct writes, one stream, tails shaped
    {"NGSLD_TEST_RUN_DIRECT": "0"},                                 # device buffers + D2H, tapered batches
    {"NGSLD_TEST_RUN_DIRECT": "0", "NGSLD_TEST_RUN_TAPER": "0"},
    {"NGSLD_TEST_RUN_STREAMS": "2"},                                # two streams, first batch half a batch
    {"NGSLD_TEST_RUN_STREAMS": "2", "NGSLD_TEST_RUN_DIRECT": "0"},
    {"NGSLD_TEST_TAIL_LEN": "0"},                                   # no short runs at the launches' ends
    {"NGSLD_TEST_TAIL_LEN": "1", "NGSLD_TEST_TAIL_PAIRS": "100000"},
    {"NGSLD_TEST_BATCH_PAIRS": "40000"},                            # many small batches
    {"NGSLD_TEST_BATCH_PAIRS": "40000", "NGSLD_TEST_RUN_STREAMS": "2"},
    {"NGSLD_TEST_BATCH_PAIRS": "40000", "NGSLD_TEST_RUN_DIRECT": "0"},
    {"NGSLD_TEST_PIN_LIMIT_BYTES": "2500000"},                      # pinned memory is scarce: batches halve until two buffers fit
]


def _run(raw, pd, kw, env, monkeypatch, device_run=False):
    for k in ("NGSLD_TEST_RUN_DIRECT", "NGSLD_TEST_RUN_TAPER", "NGSLD_TEST_RUN_STREAMS", "NGSLD_TEST_TAIL_LEN", "NGSLD_TEST_TAIL_PAIRS", "NGSLD_TEST_BATCH_PAIRS",
              "NGSLD_REPLAY", "NGSLD_TEST_PIN_LIMIT_BYTES"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw, ignore_miss_data=kw.get("ignore_miss_data", False))
        eng.set_pos_dist(pd)
        n = eng.plan(**kw)
        s1, s2, std, ext = eng.run()
        assert len(s1) == n
        dev = None
        if device_run:
            import torch
            d_std = torch.zeros(max(n, 1) * 32, dtype=torch.uint8, device="cuda:0")
            d_ext = torch.zeros(max(n, 1) * 40, dtype=torch.uint8, device="cuda:0")
            eng.run_device(0, raw.shape[0], d_std.data_ptr(), d_ext.data_ptr() if ext is not None else None, None)
            dev = d_std.cpu().numpy().tobytes()
        return s1, s2, std.tobytes(), ext.tobytes() if ext is not None else b"", dev
    finally:
        eng.close()


@pytest.mark.parametrize("shape", ["run_500", "group_100", "multi_1000", "called_300"])
def test_every_route_delivers_the_same_records(shape, monkeypatch):
    if shape == "run_500":
        n_sites, n_ind, kw = 2500, 500, dict(max_kb_dist=15, extend_out=True)
    elif shape == "group_100":
        n_sites, n_ind, kw = 700, 100, dict(extend_out=True)
    elif shape == "multi_1000":
        n_sites, n_ind, kw = 1200, 1000, dict(max_kb_dist=10, extend_out=True, ignore_miss_data=True)
    else:
        n_sites, n_ind, kw = 900, 300, dict(max_kb_dist=30, extend_out=False)
    raw = synth.make_gl_numpy(n_sites, n_ind, 900 + n_ind, depth=6.0)
    if shape == "called_300":
        raw = np.eye(3)[raw.argmax(axis=2)]
    raw[11] = [1.0, 0.0, 0.0]                                   # a monomorphic site: its pairs are flagged and replayed
    chrs, pos = synth.make_positions(n_sites, 77, n_chr=2)
    pd = shard.pos_dist_from_positions(chrs, pos)
    base = _run(raw, pd, kw, {}, monkeypatch, device_run=True)
    assert len(base[0]) > 50_000
    # the records ngsld_run_device leaves on the device are the same, but for the few pairs only TEXT output has replayed
    # (a printed digit on a rounding point: flag_text) -- those may differ in their last bits
    std_h = np.frombuffer(base[2], dtype=np.uint64).reshape(-1, 4)
    std_d = np.frombuffer(base[4], dtype=np.uint64).reshape(-1, 4)[:len(std_h)]
    assert np.count_nonzero(np.any(std_h != std_d, axis=1)) <= max(8, len(std_h) // 500)
    for env in VARIANTS[1:]:
        got = _run(raw, pd, kw, env, monkeypatch)
        assert np.array_equal(got[0], base[0]) and np.array_equal(got[1], base[1]), env
        assert got[2] == base[2] and got[3] == base[3], f"records differ on route {env}"


@pytest.mark.parametrize("streams", ["1", "2"])
def test_reported_kernel_time_of_a_two_stream_run_is_a_span_not_a_sum(streams, monkeypatch):
    """ngsld_last_kernel_time after ngsld_run (round 4's advisor): launches on two compute streams share the device, so their time
    is first start .. last end -- the sum of their durations would exceed the wall time of the run.  Text batches (two streams by
    default; NGSLD_TEST_TEXT_STREAMS=1: one) of a run long enough to measure."""
    import time
    monkeypatch.setenv("NGSLD_TEST_TEXT_STREAMS", streams)
    n_sites, n_ind = 6000, 200
    raw = synth.make_gl_numpy(n_sites, n_ind, seed=9, depth=8.0)
    chrs, pos = synth.make_positions(n_sites, 9)
    pd = shard.pos_dist_from_positions(chrs, pos)
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw)
        eng.set_pos_dist(pd)
        eng.set_tuning(batch_pairs=1 << 17)
        n = eng.plan(max_kb_dist=100, extend_out=True)
        eng.set_text_output([f"c:{p}" for p in pos])
        eng.run_text()                                   # (buffers sized and pinned)
        t0 = time.perf_counter()
        eng.run_text()
        wall_ms = (time.perf_counter() - t0) * 1e3
        ms, launches, pairs = eng.last_kernel_time()
    finally:
        eng.close()
    assert pairs == n and launches >= 20
    assert 0.0 < ms <= 1.02 * wall_ms, (streams, ms, wall_ms, launches)   # (the wall time is mostly the Python sink's)
