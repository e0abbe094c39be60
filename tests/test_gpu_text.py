"""GPU: TSV rows formatted on the device (ngsld_set_text_output) are byte for byte the host writer's
(ngsld_host_write_batch / ngsld_host_format_pair, themselves pinned to the oracle's text by the golden md5 tests)."""
import os
import subprocess

import numpy as np
import pytest

from ngsld_amd import capi, shard, synth

pytestmark = pytest.mark.gpu


def _cli(args, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([capi.CLI_PATH] + args, capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()
    return r.stdout


@pytest.mark.parametrize("case", ["std", "ext", "ext_filters", "nopos", "degenerate", "many_batches", "n1000", "slabs", "mixed"])
def test_cli_device_text_equals_host_text(tmp_path, case):
    n_sites, n_ind = (400, 60) if case != "n1000" else (120, 1000)
    raw = synth.make_gl_numpy(n_sites, n_ind, 77, depth=3.0)
    if case == "degenerate":
        raw[5] = np.array([1.0, 0.0, 0.0])                        # monomorphic: -nan columns
        raw[9] = 1.0 / 3.0                                          # no information at all
        raw[11] = np.eye(3)[np.random.default_rng(1).integers(0, 3, size=n_ind)]   # hard calls
    g = str(tmp_path / "in.glf")
    raw.tofile(g)
    chrs, pos = synth.make_positions(n_sites, 77, max_gap=300, n_chr=3)
    p = str(tmp_path / "in.pos")
    synth.write_pos(p, chrs, pos, extra_col=(case == "ext_filters"))
    args = ["--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--verbose", "0", "--n_threads", "3"]
    if case == "nopos":
        args += ["--max_kb_dist", "0", "--extend_out"]
    else:
        args += ["--pos", p, "--max_kb_dist", "0" if case in ("degenerate", "n1000") else "20"]
    if case in ("ext", "ext_filters", "degenerate", "many_batches", "n1000", "slabs", "mixed"):
        args += ["--extend_out"]
    if case == "ext_filters":
        args += ["--min_maf", "0.1", "--rnd_sample", "0.5", "--seed", "42", "--ignore_miss_data"]
    env = {"NGSLD_TEST_BATCH_PAIRS": "700"} if case == "many_batches" else {}
    if case == "mixed":                                             # every third batch falls back to records + host formatter
        env = {"NGSLD_TEST_BATCH_PAIRS": "500", "NGSLD_TEST_TEXT_FALLBACK_EVERY": "3"}
    if case == "slabs":                                             # the streamed path: slabs of 150 sites, text per slab
        env = {"NGSLD_TEST_SLAB_SITES": "150", "NGSLD_TEST_BATCH_PAIRS": "900"}
    host = _cli(args, dict(env, NGSLD_HOST_TEXT="1"))
    dev = _cli(args, env)
    assert len(host) > 1000 and host.count(b"\n") > 10
    assert dev == host


def test_api_text_rows(engine):
    """ngsld_run with text output against ngsld_host_format_pair row by row (labels with TABs, long labels)."""
    n_sites, n_ind = 150, 40
    raw = synth.make_gl_numpy(n_sites, n_ind, 5, depth=2.0)
    chrs, pos = synth.make_positions(n_sites, 5, n_chr=2)
    pd = shard.pos_dist_from_positions(chrs, pos)
    labels = [f"{c}:{q}" + ("\tid%d" % k if k % 7 == 0 else "") + ("x" * 40 if k == 13 else "") for k, (c, q) in
              enumerate(zip(chrs, pos))]
    engine.set_geno_raw(raw)
    engine.set_pos_dist(pd)
    engine.plan(max_kb_dist=10, extend_out=True)
    maf = engine.maf()
    s1, s2, std, ext = engine.run()
    engine.set_text_output(labels)
    try:
        text, fallbacks = engine.run_text()
    finally:
        engine.set_text_output(None, enable=False)
    assert fallbacks == 0
    rows = text.split(b"\n")
    assert rows[-1] == b"" and len(rows) - 1 == len(s1)
    for k in np.random.default_rng(3).choice(len(s1), size=min(400, len(s1)), replace=False):
        a, b = int(s1[k]), int(s2[k])
        dist = float(np.sum(pd[a + 1:b + 1]))
        want = capi.format_pair(labels[a], labels[b], dist, std[k], ext[k], maf[a], maf[b]).encode()
        assert rows[k] + b"\n" == want, (k, rows[k], want)


def test_cli_gz_output_is_the_plain_output_compressed(tmp_path):
    """--out name.gz: gzip members written by --n_threads deflate workers; decompressed = the plain output, byte for byte --
    resident, streamed (slabs) and multi-device runs."""
    import gzip
    import subprocess
    import numpy as np
    from ngsld_amd import capi, synth
    n_sites, n_ind = 3000, 40
    raw = synth.make_gl_numpy(n_sites, n_ind, seed=71, depth=4.0)
    chrs, pos = synth.make_positions(n_sites, 71, max_gap=150, n_chr=2)
    g, p = str(tmp_path / "in.glf"), str(tmp_path / "in.pos")
    raw.tofile(g)
    synth.write_pos(p, chrs, pos)
    base = [capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--pos", p, "--max_kb_dist", "20",
            "--extend_out", "--verbose", "0", "--n_threads", "4"]
    plain = str(tmp_path / "plain.ld")
    r = subprocess.run(base + ["--out", plain], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = open(plain, "rb").read()
    assert want.count(b"\n") > 100_000
    for tag, extra, env in (("resident", [], {}), ("slabs", [], {"NGSLD_TEST_SLAB_SITES": "700"}), ("devices", ["--devices", "0,0"], {})):
        out = str(tmp_path / f"{tag}.ld.gz")
        import os
        r = subprocess.run(base + extra + ["--out", out], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        with gzip.open(out, "rb") as fh:
            assert fh.read() == want, tag
        assert os.path.getsize(out) < len(want) // 2


def _text_run(eng, raw, pd, labels, **plan):
    eng.set_geno_raw(raw)
    eng.set_pos_dist(pd)
    eng.plan(**plan)
    eng.set_text_output(labels)
    try:
        text, fallbacks = eng.run_text()
    finally:
        eng.set_text_output(None, enable=False)
    return text, fallbacks, eng.replay_info()


@pytest.mark.parametrize("extend", [False, True])
def test_host_replayed_rows_are_overwritten_in_the_hosts_text(monkeypatch, extend):
    """The pairs a text batch leaves to the host's exact-order replay (engine_run.hip: send_flag_rows / apply_host_patch): their
    rows' value columns are overwritten in the text the host has received -- nothing goes back to the device.  The text is the
    one the device's way gives (NGSLD_TEST_TEXT_HOST_PATCH=0: records patched on the device, the batch written again), also when a
    patched batch falls back to that way half-way through (NGSLD_TEST_TEXT_HOST_PATCH_FAIL_EVERY), with labels that hold TABs."""
    n_sites, n_ind = 500, 50
    raw = synth.make_gl_numpy(n_sites, n_ind, 123, depth=2.0, mono_frac=0.1)
    chrs, pos = synth.make_positions(n_sites, 123, n_chr=2)
    pd = shard.pos_dist_from_positions(chrs, pos)
    labels = [f"{c}:{q}" + ("\tid\t%d" % k if k % 5 == 0 else "") for k, (c, q) in enumerate(zip(chrs, pos))]
    plan = dict(max_kb_dist=0, extend_out=extend)
    eng = capi.Engine(0)
    try:
        eng.set_exact_store(0)            # (every flagged pair is the host's)
        eng.set_tuning(batch_pairs=3000)
        monkeypatch.setenv("NGSLD_TEST_TEXT_HOST_PATCH", "0")
        want, fb0, info0 = _text_run(eng, raw, pd, labels, **plan)
        monkeypatch.delenv("NGSLD_TEST_TEXT_HOST_PATCH")
        got, fb1, info1 = _text_run(eng, raw, pd, labels, **plan)
        monkeypatch.setenv("NGSLD_TEST_TEXT_HOST_PATCH_FAIL_EVERY", "2")
        half, fb2, info2 = _text_run(eng, raw, pd, labels, **plan)
        monkeypatch.delenv("NGSLD_TEST_TEXT_HOST_PATCH_FAIL_EVERY")
        assert fb0 == fb1 == fb2 == 0
        assert info0["pairs_on_host"] > 50 and info0["text_rows_patched"] == 0
        # (a batch that leaves the host more than 1,024 pairs -- kFlagRowsCap -- goes the device's way)
        assert info1["pairs_on_host"] == info0["pairs_on_host"] and 1000 < info1["text_rows_patched"] <= info1["pairs_on_host"]
        assert 0 < info2["text_rows_patched"] < info1["text_rows_patched"] and info2["pairs_on_host"] == info0["pairs_on_host"]
        assert got == want and half == want
        # ... and the rows are the host formatter's over the replayed records
        maf = eng.maf()
        s1, s2, std, ext = eng.run()
    finally:
        eng.close()
    rows = got.split(b"\n")
    assert len(rows) - 1 == len(s1)
    for k in range(0, len(s1), 37):
        a, b = int(s1[k]), int(s2[k])
        dist = float(np.sum(pd[a + 1:b + 1]))
        assert rows[k] + b"\n" == capi.format_pair(labels[a], labels[b], dist, std[k], ext[k] if extend else None, maf[a], maf[b]).encode()
    print(f"text rows overwritten in the host's text: {info1['text_rows_patched']} of {len(s1)} pairs; with every second patched batch "
          f"falling back {info2['text_rows_patched']}")
