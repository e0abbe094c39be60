"""The oracle against the reference's OWN compiled functions (oracle/_ref/libngsld_ref.so), bit for bit, on
fresh random inputs.  Runs wherever oracle/_ref exists (it is built from /root/reference in the build
container and travels as a .so); the committed goldens carry the same evidence where it does not."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

from ngsld_amd import synth
from oracle import orc

ref = orc.ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("n_sites,n_ind,depth,seed,log_scale,ignore", [
    (40, 24, 1.0, 101, False, False), (30, 100, 5.0, 102, False, True), (20, 500, 10.0, 103, True, False),
    (12, 1000, 10.0, 104, False, False), (25, 37, 2.0, 105, True, True)])
def test_reader_maf_em_bit_exact(n_sites, n_ind, depth, seed, log_scale, ignore):
    raw = synth.make_gl_numpy(n_sites, n_ind, seed, depth=depth)
    rng = np.random.default_rng(seed)
    raw[rng.random((n_sites, n_ind)) < 0.1] = 1.0 / 3.0     # some missing triples
    if log_scale:
        with np.errstate(divide="ignore"):
            raw = np.log(raw)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.glf")
        raw.tofile(path)
        gl_ref = np.empty_like(raw)
        ref.ref_read_geno_bin(path.encode(), int(log_scale), n_ind, n_sites, orc.dp(gl_ref))
        gl_orc = np.empty_like(raw)
        err = C.create_string_buffer(256)
        assert orc.lib().orc_read_geno_bin(path.encode(), int(log_scale), n_ind, n_sites, orc.dp(gl_orc), err, 256) == 0
    assert np.array_equal(gl_ref, gl_orc)
    o = orc.Oracle(raw, log_scale=log_scale, ignore_miss_data=ignore)
    assert np.array_equal(o.gl_log, gl_ref)
    maf, expg = np.empty(n_sites), np.empty((n_sites, n_ind))
    ref.ref_preprocess(orc.dp(gl_ref), n_ind, n_sites, int(ignore), orc.dp(maf), orc.dp(expg))
    assert np.array_equal(maf, o.maf, equal_nan=True) and np.array_equal(expg, o.expg) and np.array_equal(gl_ref, o.gl)
    for r in o.run():
        hap, n = np.zeros(4), C.c_uint64()
        it = ref.ref_haplo_freq(orc.dp(hap), C.byref(n), orc.dp(o.gl[r["s1"]]), orc.dp(o.gl[r["s2"]]),
                                o.maf[r["s1"]], o.maf[r["s2"]], n_ind, int(ignore))
        assert np.array_equal(hap, r["hap"], equal_nan=True) and it == r["n_iter"] and n.value == r["n_ind_data"]


def test_single_em_step_bit_exact():
    rng = np.random.default_rng(7)
    for _ in range(50):
        n = int(rng.integers(1, 80))
        a = rng.dirichlet([1, 1, 1], size=n)
        b = rng.dirichlet([1, 1, 1], size=n)
        f = rng.dirichlet([1, 1, 1, 1])
        f1, f2, e = f.copy(), f.copy(), C.c_int(0)
        x1 = ref.ref_pair_freq_iter(orc.dp(f1), orc.dp(a), orc.dp(b), n, 0)
        x2 = orc.lib().orc_pair_freq_iter(orc.dp(f2), orc.dp(a), orc.dp(b), n, 0, C.byref(e))
        assert x1 == x2 == n and np.array_equal(f1, f2)


def test_pos_reader_matches_reference(tmp_path):
    chrs, pos = synth.make_positions(50, 9, n_chr=3)
    p = tmp_path / "a.pos"
    synth.write_pos(str(p), chrs, pos, header=True, extra_col=True)
    want = np.empty(50)
    ref.ref_read_dist(str(p).encode(), 1, 50, orc.dp(want))
    P = orc.OrcParams()
    P.in_pos, P.in_pos_header, P.n_sites = str(p).encode(), 1, 50
    err = C.create_string_buffer(256)
    assert orc.lib().orc_read_pos(C.byref(P), err, 256) == 0
    got = np.ctypeslib.as_array(P.pos_dist, shape=(50,)).copy()
    buf = C.create_string_buffer(50 * 128)
    assert ref.ref_read_labels(str(p).encode(), 1, buf, 128, 50) == 50
    for s in range(50):
        assert P.labels[s] == buf.raw[s * 128:(s + 1) * 128].split(b"\0")[0]
    orc.lib().orc_free_pos(C.byref(P))
    assert np.array_equal(got, want)


# ---- ngsLD.cpp's own in-tree arithmetic and formats (a1 walk, a4 D / D' / r2, a5 float chi2, a6 fprintf), compiled from
# the reference's text by oracle/build_ref.sh (line ranges cut out by anchor): the oracle's restatement against them ----
needs_cpp = pytest.mark.skipif(ref is None or not hasattr(ref, "ref_pair_stats"),
                               reason="oracle/_ref predates the ngsLD.cpp doors (rebuild with oracle/build_ref.sh)")


def _hap_vectors(n: int, seed: int) -> np.ndarray:
    """Random and degenerate haplotype-frequency vectors: simplex points at several concentrations, vectors with one or
    two exact zeros, a (nearly) monomorphic site, margins of 1e-16 .. 1e-6, un-normalised sums, NaN / inf entries."""
    rng = np.random.default_rng(seed)
    out = []
    for conc in (1.0, 0.1, 0.01, 10.0):
        out.append(rng.dirichlet([conc] * 4, size=n // 8))
    z = rng.dirichlet([1.0] * 4, size=n // 8)
    z[np.arange(len(z)), rng.integers(0, 4, len(z))] = 0.0
    out.append(z / z.sum(axis=1, keepdims=True))
    z = rng.dirichlet([1.0] * 4, size=n // 8)
    k = rng.integers(0, 4, len(z))
    z[np.arange(len(z)), k] = 0.0
    z[np.arange(len(z)), (k + rng.integers(1, 4, len(z))) % 4] = 0.0
    out.append(z / z.sum(axis=1, keepdims=True))
    tiny = 10.0 ** rng.uniform(-17, -6, size=(n // 8, 1))
    z = rng.dirichlet([1.0] * 4, size=n // 8)
    z[:, 2:] *= tiny                                     # site 1 nearly monomorphic
    z[:, 0] = 1.0 - z[:, 1] - z[:, 2] - z[:, 3]
    out.append(z)
    z = rng.dirichlet([1.0] * 4, size=n // 8) * (1.0 + rng.normal(0, 1e-12, size=(n // 8, 1)))   # sums off by rounding
    out.append(z)
    special = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [0.5, 0, 0, 0.5], [0, 0.5, 0.5, 0],
                        [0.25, 0.25, 0.25, 0.25], [np.nan] * 4, [np.nan, 0.3, 0.3, 0.4], [np.inf, 0, 0, 0],
                        [0.5, 0.5, 0, 0], [0.5, 0, 0.5, 0], [1 - 1e-16, 1e-16, 0, 0], [0, 0, 0, 0]], dtype=np.float64)
    out.append(special)
    return np.ascontiguousarray(np.concatenate(out))


def _bits(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


@needs_cpp
def test_pair_stats_bit_exact_against_the_reference_lines():
    """orc_pair_stats == ngsLD.cpp:296-306 + :328-333 compiled here, bit for bit (NaN payloads and signs included) on
    10^5 haplotype vectors."""
    L = orc.lib()
    haps = _hap_vectors(100_000, 11)
    got = np.empty((len(haps), 5))
    want = np.empty((len(haps), 5))
    gchi = np.empty(len(haps), dtype=np.float32)
    wchi = np.empty(len(haps), dtype=np.float32)
    for k, h in enumerate(haps):
        h = np.ascontiguousarray(h)
        c1, c2 = C.c_float(), C.c_float()
        L.orc_pair_stats(orc.dp(h), orc.dp(got[k, 0:1]), orc.dp(got[k, 1:2]), orc.dp(got[k, 2:3]), orc.dp(got[k, 3:5]),
                         C.byref(c1))
        ref.ref_pair_stats(orc.dp(h), orc.dp(want[k, 0:1]), orc.dp(want[k, 1:2]), orc.dp(want[k, 2:3]),
                           orc.dp(want[k, 3:5]), C.byref(c2))
        gchi[k], wchi[k] = c1.value, c2.value
    # NaN payload / sign: x86 produces the default quiet NaN for 0/0 and inf-inf in both builds; compared as bits
    assert np.array_equal(_bits(got), _bits(want))
    assert np.array_equal(_bits(gchi), _bits(wchi))
    assert np.isnan(want[:, 1]).sum() > 100 and np.isinf(want[:, 2]).sum() + np.isnan(want[:, 2]).sum() > 100   # the degenerate ones are in


@needs_cpp
def test_rows_and_header_byte_equal_to_the_reference_fprintf():
    """orc_print_pair / orc_print_header == the reference's own fprintf lines (ngsLD.cpp:77, :314-351), byte for byte,
    both column sets: labels with a TAB inside, inf distance, -nan / nan / inf columns, %lu counts."""
    L = orc.lib()
    for ext in (0, 1):
        a, b = C.create_string_buffer(1024), C.create_string_buffer(1024)
        na, nb = L.orc_format_header(a, 1024, ext), ref.ref_print_header(b, 1024, ext)
        assert na == nb > 0 and a.raw[:na] == b.raw[:nb]
    haps = _hap_vectors(4_000, 12)
    rng = np.random.default_rng(12)
    P = orc.OrcParams()
    maf = np.ascontiguousarray(rng.uniform(0, 0.5, 2))
    maf_nan = np.array([np.nan, 0.25])
    labels = (C.c_char_p * 2)(b"chr1:1234\tsnpA", b"chr22:99999999")
    P.labels = labels
    rec = np.zeros(1, dtype=orc.PAIR_DTYPE)
    for k, h in enumerate(haps):
        h = np.ascontiguousarray(h)
        mm = maf_nan if k % 97 == 0 else maf
        P.maf = orc.dp(mm)
        dist = float("inf") if k % 5 == 0 else float(rng.integers(1, 10**9))
        r2p = [float(rng.uniform()), float("nan"), -float("nan"), 0.0, 1.0, 0.9999995, 1e-7][k % 7]
        n_data, n_iter = int(rng.integers(0, 5000)), int(rng.integers(0, 101))
        c = C.c_float()
        r = rec[0]
        r["s1"], r["s2"], r["dist"], r["r2pear"], r["n_ind_data"], r["n_iter"] = 0, 1, dist, r2p, n_data, n_iter
        D, Dp, r2, hm = np.zeros(1), np.zeros(1), np.zeros(1), np.zeros(2)
        L.orc_pair_stats(orc.dp(h), orc.dp(D), orc.dp(Dp), orc.dp(r2), orc.dp(hm), C.byref(c))
        r["D"], r["Dp"], r["r2"], r["hap"], r["hap_maf"], r["chi2"] = D[0], Dp[0], r2[0], h, hm, c.value
        for ext in (0, 1):
            P.extend_out = ext
            a, b = C.create_string_buffer(2048), C.create_string_buffer(2048)
            na = L.orc_format_pair(a, 2048, C.byref(P), rec.ctypes.data_as(C.c_void_p))
            nb = ref.ref_format_row(b, 2048, labels[0], labels[1], dist, r2p, orc.dp(h), n_data, float(mm[0]), float(mm[1]),
                                    n_iter, ext)
            assert na == nb > 0 and a.raw[:na] == b.raw[:nb], (a.raw[:na], b.raw[:nb])


@needs_cpp
@pytest.mark.parametrize("max_kb,max_snp,min_maf,n_chr", [(0, 0, 0.0, 1), (5, 0, 0.0, 2), (0, 7, 0.0, 3), (3, 4, 0.12, 2),
                                                           (1, 0, 0.3, 1), (0, 0, 0.2, 4)])
def test_window_walk_equals_the_reference_loop(max_kb, max_snp, min_maf, n_chr):
    """The oracle's pair walk (s2 and the running dist of every pair that passes the distance / SNP-count / maf filters)
    == ngsLD.cpp:240-275 compiled here, row by row; chromosome breaks (inf gaps) and a NaN maf included."""
    n_sites, n_ind = 150, 12
    raw = synth.make_gl_numpy(n_sites, n_ind, seed=300 + max_kb + max_snp, depth=3.0)
    raw[17] = 1.0 / 3.0
    chrs, pos = synth.make_positions(n_sites, 31 + n_chr, n_chr=n_chr)
    from ngsld_amd import shard
    pd = shard.pos_dist_from_positions(chrs, pos)
    o = orc.Oracle(raw, pd, max_kb_dist=max_kb, max_snp_dist=max_snp, min_maf=min_maf)
    rec = o.run()
    maf = np.ascontiguousarray(o.maf.copy())
    pdc = np.ascontiguousarray(pd.copy())
    s2buf = np.empty(n_sites, dtype=np.uint64)
    dbuf = np.empty(n_sites)
    total = 0
    for s1 in range(n_sites):
        n = ref.ref_walk(n_sites, orc.dp(pdc), orc.dp(maf), max_kb, max_snp, min_maf, s1,
                         s2buf.ctypes.data_as(C.POINTER(C.c_uint64)), orc.dp(dbuf), n_sites)
        mine = rec[rec["s1"] == s1]
        assert n == len(mine)
        assert np.array_equal(s2buf[:n], mine["s2"]) and np.array_equal(_bits(dbuf[:n].copy()), _bits(mine["dist"].copy()))
        total += n
    assert total == len(rec) > 0


def test_binary_reader_on_values_at_the_edges_of_the_double_range():
    """Zeros, denormals, 1e300, negative numbers, -0.0, exact thirds ... in natural and log scale: the oracle's binary reader
    (normalisation, the NaN check) gives the reference reader's values or fails where it fails (read_data.cpp:28-47,106-116).
    The reference runs in a forked child: its error() ends the process."""
    rng = np.random.default_rng(5)
    special = np.array([0.0, 1.0, 1 / 3, 0.5, 1e-310, 5e-324, 1e-300, 1e300, 1.7e308, -1.0, -0.0, 1e-17, 1 - 1e-16, 2.0, 3.0])
    failed = 0
    with tempfile.TemporaryDirectory() as d:
        path, shared = os.path.join(d, "b.glf"), os.path.join(d, "ref.npy")
        for trial in range(200):
            n_sites, n_ind = int(rng.integers(1, 6)), int(rng.integers(1, 6))
            log_scale = bool(trial % 2)
            raw = rng.choice(special, size=(n_sites, n_ind, 3))
            m = rng.random((n_sites, n_ind)) < 0.3
            raw[m] = rng.dirichlet([1, 1, 1], size=int(m.sum()))
            if log_scale:
                with np.errstate(all="ignore"):
                    raw = np.log(np.abs(raw))
                if trial % 4 == 1:
                    raw[rng.random(raw.shape) < 0.1] = rng.choice([1.0, 700.0, -1e15, -745.0, 0.0])
            raw.tofile(path)
            pid = os.fork()
            if pid == 0:
                os.dup2(os.open(os.devnull, os.O_WRONLY), 2)
                out = np.empty_like(raw)
                ref.ref_read_geno_bin(path.encode(), int(log_scale), n_ind, n_sites, orc.dp(out))
                np.save(shared, out)
                os._exit(0)
            _, st = os.waitpid(pid, 0)
            ok_ref = os.WIFEXITED(st) and os.WEXITSTATUS(st) == 0
            gl = np.empty_like(raw)
            err = C.create_string_buffer(256)
            rc = orc.lib().orc_read_geno_bin(path.encode(), int(log_scale), n_ind, n_sites, orc.dp(gl), err, 256)
            assert ok_ref == (rc == 0), (trial, ok_ref, rc, err.value)
            if ok_ref:
                assert np.array_equal(np.load(shared), gl, equal_nan=True), trial
            else:
                failed += 1
                assert b"NaN found" in err.value
    assert 20 < failed < 180   # (both outcomes are exercised)
