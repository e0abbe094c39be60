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
