"""GPU: several devices from ONE process (ngsld_run_multi, `ngsLD --devices`).  The test box has one GPU, so the parts
share device 0 -- what is tested is that the parts' records, in part order, are the single-device run byte for byte,
on every distribution path a single GPU can exercise (per-part upload, the broadcast buffer copied device to device)."""
import os
import subprocess

import numpy as np
import pytest

from ngsld_amd import capi, shard, synth

pytestmark = pytest.mark.gpu


def single(raw, pd, **kw):
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw, ignore_miss_data=kw.get("ignore_miss_data", False))
        eng.set_pos_dist(pd)
        eng.plan(**kw)
        return eng.run(), eng.maf()
    finally:
        eng.close()


def joined(parts):
    return [np.concatenate([p[k] for p in parts]) for k in range(4)]


@pytest.mark.parametrize("n_parts", [2, 3])
@pytest.mark.parametrize("case", ["windowed", "all_pairs", "all_pairs_upload", "filters"])
def test_parts_concatenate_to_the_single_device_run(n_parts, case, monkeypatch):
    n_sites, n_ind = (700, 120) if case != "all_pairs" else (260, 300)
    raw = synth.make_gl_numpy(n_sites, n_ind, seed=31, depth=4.0)
    chrs, pos = synth.make_positions(n_sites, 31, max_gap=300, n_chr=2)
    pd = shard.pos_dist_from_positions(chrs, pos)
    for s in (3, n_sites // 3, n_sites // 2, n_sites - 2):     # degenerate sites: their pairs are replayed in every layout
        raw[s] = [1.0, 0.0, 0.0] if s % 2 else 1.0 / 3.0
    kw = dict(extend_out=True)
    if case == "windowed":
        kw.update(max_kb_dist=8)
    elif case == "filters":
        kw.update(max_kb_dist=12, max_snp_dist=40, min_maf=0.12, rnd_sample=0.4, seed=77, ignore_miss_data=True)
        raw[::7, ::5] = 1.0 / 3.0
    if case == "all_pairs_upload":
        monkeypatch.setenv("NGSLD_TEST_MULTI_DIST", "upload")
    (s1, s2, std, ext), maf = single(raw, pd, **kw)
    parts, maf_m, per = capi.run_multi(raw, pd, [0] * n_parts, **kw)
    got = joined(parts)
    # one device listed several times: the broadcast buffer is copied device to device (RCCL needs distinct devices:
    # tests/test_gpu_multi_ranks.py); windowed runs and NGSLD_TEST_MULTI_DIST=upload send every part its own slab
    assert capi.multi_last_distribution() == ("peer_copy" if case == "all_pairs" else "upload")
    assert sum(per) == len(s1) and [len(p[0]) for p in parts] == per
    assert np.array_equal(got[0], s1) and np.array_equal(got[1], s2)
    assert got[2].tobytes() == std.tobytes() and got[3].tobytes() == ext.tobytes()
    assert maf_m.tobytes() == maf.tobytes()
    if case in ("windowed", "all_pairs"):            # balanced by candidate pairs
        assert max(per) <= 1.25 * (sum(per) / n_parts) + 2000


def test_one_kernel_family_for_all_parts():
    """Called genotypes everywhere except a few sites of the LAST part: no part may run the genotype-combination kernel
    while another runs the per-individual one (same values to 1e-12, not the same bits)."""
    rng = np.random.default_rng(5)
    n_sites, n_ind = 300, 90
    raw = np.eye(3)[rng.integers(0, 3, size=(n_sites, n_ind))]
    raw[280:284] = synth.make_gl_numpy(4, n_ind, seed=6, depth=3.0)
    (s1, s2, std, ext), _ = single(raw, None, extend_out=True)
    parts, _, per = capi.run_multi(raw, None, [0, 0, 0], extend_out=True)
    got = joined(parts)
    assert got[2].tobytes() == std.tobytes() and got[3].tobytes() == ext.tobytes()


def test_text_parts_are_the_single_device_text():
    n_sites, n_ind = 500, 64
    raw = synth.make_gl_numpy(n_sites, n_ind, seed=41, depth=5.0)
    chrs, pos = synth.make_positions(n_sites, 41, max_gap=100, n_chr=1)
    pd = shard.pos_dist_from_positions(chrs, pos)
    labels = [f"{c}:{p}" for c, p in zip(chrs, pos)]
    eng = capi.Engine(0)
    try:
        eng.set_geno_raw(raw)
        eng.set_pos_dist(pd)
        eng.plan(max_kb_dist=5, extend_out=True)
        eng.set_text_output(labels)
        want, _ = eng.run_text()
    finally:
        eng.close()
    parts, _, _ = capi.run_multi(raw, pd, [0, 0], labels=labels, text_output=True, max_kb_dist=5, extend_out=True)
    assert b"".join(parts) == want


@pytest.mark.parametrize("mode", ["windowed_bin", "all_pairs_bin", "text_called"])
def test_cli_devices_flag(tmp_path, mode):
    """`ngsLD --devices 0,0,0` writes what `ngsLD --device 0` writes."""
    n_sites, n_ind = 400, 50
    raw = synth.make_gl_numpy(n_sites, n_ind, seed=51, depth=4.0)
    chrs, pos = synth.make_positions(n_sites, 51, max_gap=200, n_chr=2)
    p = str(tmp_path / "in.pos")
    synth.write_pos(p, chrs, pos)
    if mode == "text_called":
        import gzip
        g = str(tmp_path / "in.geno.gz")
        calls = raw.argmax(axis=2)
        calls[::9, ::4] = -1
        with gzip.open(g, "wt") as fh:
            for s in range(n_sites):
                fh.write(f"{chrs[s]}\t{pos[s]}\t" + "\t".join(str(int(v)) for v in calls[s]) + "\n")
    else:
        g = str(tmp_path / "in.glf")
        raw.tofile(g)
    flags = ["--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p, "--extend_out", "--verbose", "0",
             "--max_kb_dist", "0" if mode == "all_pairs_bin" else "10", "--min_maf", "0.06"]
    outs = []
    for dev in (["--device", "0"], ["--devices", "0,0,0"], ["--devices", "0-0"]):
        o = str(tmp_path / ("out" + "_".join(dev).replace(",", "").replace("-", "")))
        r = subprocess.run([capi.CLI_PATH] + flags + dev + ["--out", o], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(open(o, "rb").read())
        assert not [f for f in os.listdir(tmp_path) if ".part" in f], "part files must be gone"
    assert outs[0] == outs[1] == outs[2] and outs[0].count(b"\n") > 1000
    # --keep_parts: parts 1.. stay in <out>.part<k>; the output followed by them, in order, is the same table
    o = str(tmp_path / "out_kept")
    r = subprocess.run([capi.CLI_PATH] + flags + ["--devices", "0,0,0", "--keep_parts", "--out", o], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    kept = sorted(f for f in os.listdir(tmp_path) if f.startswith("out_kept.part"))
    assert kept == ["out_kept.part1", "out_kept.part2"]
    assert open(o, "rb").read() + b"".join(open(str(tmp_path / f), "rb").read() for f in kept) == outs[0]
    # ... and a merge that cannot stay inside the kernel (the output is a pipe) takes the buffered copy
    r = subprocess.run([capi.CLI_PATH] + flags + ["--devices", "0,0,0"], capture_output=True)
    assert r.returncode == 0 and r.stdout == outs[0]


def test_a_failing_part_stops_the_whole_job():
    """One part cannot get its device (index 99), another meets a NaN: the call returns the error, nothing hangs, no sink call."""
    raw = synth.make_gl_numpy(120, 40, seed=61)
    with pytest.raises(capi.NgsldError) as e:
        capi.run_multi(raw, None, [0, 99], extend_out=True)
    assert "part 1" in e.value.msg
    bad = raw.copy()
    bad[100, 3, :] = -1.0                    # log(-1) = NaN, in the last part only
    with pytest.raises(capi.NgsldError) as e:
        capi.run_multi(bad, None, [0, 0, 0], extend_out=True)
    assert e.value.code == capi.ERR_NAN and "NaN found" in e.value.msg
