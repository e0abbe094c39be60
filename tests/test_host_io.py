"""Host-side logic of the product (include/ngsld_host.h) on the CPU: readers and the TSV writer, against the
golden vectors (reference-produced pos_dist/labels, oracle TSV text)."""
import gzip
import os

import numpy as np
import pytest

from ngsld_amd import capi, shard
from util import Fixture, fixtures

POS_FIXTURES = [n for n in fixtures() if Fixture(n).has_pos]


@pytest.mark.parametrize("name", POS_FIXTURES)
def test_read_pos_matches_reference(name, tmp_path):
    fx = Fixture(name)
    _, p = fx.write_inputs(str(tmp_path))
    pd, labels = capi.read_pos(p, fx.header, fx.n_sites)
    assert np.array_equal(pd, fx.pos_dist)          # produced by the reference's read_dist
    assert labels == fx.labels                      # reference's labels (first TAB -> ':')


def test_read_pos_gz_comments_and_last_line(tmp_path):
    p = tmp_path / "x.pos.gz"
    with gzip.open(p, "wt") as fh:
        fh.write("#comment\n\nchr1\t10\nchr1\t25\nchr2\t7\nchr2\t9")   # no trailing newline
    pd, labels = capi.read_pos(str(p), False, 4)
    assert np.array_equal(pd, [10.0, 15.0, np.inf, 2.0]) and labels == ["chr1:10", "chr1:25", "chr2:7", "chr2:9"]


@pytest.mark.parametrize("text,n,msg", [
    ("chr1\t10\nchr1\t5\n", 2, "invalid distance between adjacent sites!"),
    ("chr1\t10\nchr1\t10\n", 2, "invalid distance between adjacent sites!"),
    ("chr1\t10\n", 2, "wrong number of lines in POS file!"),
    ("chr1 10\nchr1 20\n", 2, "wrong POS file format!"),
    ("chr1\t10\nchr1\t20\t3\n", 2, "invalid number of fields in file!"),
    ("chr\tpos\nchr1\t20\n", 2, "header line found"),
])
def test_read_pos_errors(tmp_path, text, n, msg):
    p = tmp_path / "bad.pos"
    p.write_text(text)
    with pytest.raises(capi.NgsldError) as e:
        capi.read_pos(str(p), False, n)
    assert msg in e.value.msg


def test_read_pos_missing_file(tmp_path):
    with pytest.raises(capi.NgsldError) as e:
        capi.read_pos(str(tmp_path / "nope.pos"), False, 3)
    assert "cannot open file!" in e.value.msg


def test_pos_dist_mirror_matches_reference():
    for name in POS_FIXTURES:
        fx = Fixture(name)
        lines = [l.split("\t") for l in fx.pos_text.splitlines()][1 if fx.header else 0:]
        pd = shard.pos_dist_from_positions([l[0] for l in lines], np.array([int(l[1]) for l in lines]))
        assert np.array_equal(pd, fx.pos_dist)


def test_read_geno_bin_and_size_rule(tmp_path):
    fx = Fixture("f3_degenerate")
    g, _ = fx.write_inputs(str(tmp_path))
    raw = capi.read_geno_bin(g, fx.n_ind, fx.n_sites)
    assert np.array_equal(raw, fx.raw)
    gz = str(tmp_path / "z.glf")                      # gzread semantics: a gzip-compressed binary file works too
    with gzip.open(gz, "wb") as fh:
        fh.write(fx.raw.tobytes())
    assert np.array_equal(capi.read_geno_bin(gz, fx.n_ind, fx.n_sites), fx.raw)
    L = capi.lib()
    size = os.path.getsize(g)
    assert L.ngsld_host_geno_size_ok(size, fx.n_ind, fx.n_sites) == 1
    assert L.ngsld_host_geno_size_ok(size + 7, fx.n_ind, fx.n_sites) == 1    # integer division, ngsLD.cpp:55
    assert L.ngsld_host_geno_size_ok(size, fx.n_ind, fx.n_sites + 1) == 0
    with pytest.raises(capi.NgsldError) as e:
        capi.read_geno_bin(g, fx.n_ind, fx.n_sites + 1)
    assert "premature EOF" in e.value.msg
    with pytest.raises(capi.NgsldError) as e:
        capi.read_geno_bin(g, fx.n_ind, fx.n_sites - 1)
    assert "not at EOF" in e.value.msg


@pytest.mark.parametrize("name", [n for n in fixtures() if "orc_tsv_ext" in Fixture(n)])
@pytest.mark.parametrize("extend", [False, True])
def test_formatter_reproduces_oracle_text(name, extend):
    """Feed the golden records through the product's TSV writer: every line must equal the oracle's text,
    including -nan / inf and the float-chi2 column."""
    fx = Fixture(name)
    tag = "ext" if extend else "std"
    want = str(fx[f"orc_tsv_{tag}"]).splitlines(keepends=True)
    assert capi.format_header(extend) == want[0]
    n = len(fx["orc_s1"])
    std = np.zeros(n, dtype=capi.REC_STD)
    std["r2_ExpG"], std["D"], std["Dp"], std["r2"] = fx["orc_r2pear"], fx["orc_D"], fx["orc_Dp"], fx["orc_r2"]
    ext = np.zeros(n, dtype=capi.REC_EXT)
    ext["hap"], ext["n_ind_data"], ext["n_iter"] = fx["ref_hap"], fx["ref_n_ind_data"], fx["ref_n_iter"]
    maf = fx["ref_maf"]
    for k in range(n):
        s1, s2 = int(fx["orc_s1"][k]), int(fx["orc_s2"][k])
        line = capi.format_pair(fx.labels[s1], fx.labels[s2], float(fx["orc_dist"][k]), std[k:k + 1],
                                ext[k:k + 1] if extend else None, float(maf[s1]), float(maf[s2]))
        assert line == want[1 + k], f"pair {k}: {line!r} != {want[1 + k]!r}"


def test_formatter_nan_inf_and_null_labels():
    std = np.zeros(1, dtype=capi.REC_STD)
    std["r2_ExpG"], std["D"], std["Dp"], std["r2"] = np.nan, -0.0, -np.inf, np.inf
    line = capi.format_pair(None, None, np.inf, std, None, 0.1, 0.2)
    assert line == "(null)\t(null)\tinf\t-nan\t-0.000000\t-inf\tinf\n"


def test_format_double_is_printf_exact():
    """The fast %f / %.0f path against the C library's exact conversion (Python's % formatting is correctly
    rounded on the exact binary value, like glibc), over magnitudes, ties, subnormals and non-finite values."""
    import struct
    rng = np.random.default_rng(11)
    vals = [0.0, -0.0, 1.0, -1.0, 0.5, 0.0078125, 0.00390625, 2.5e-7, 5e-7, 4.9999999999999998e-7, 1.5e-6, 0.1, 0.7,
            1e-300, 5e-324, 123456789.987654321, 9.2e12, 9.3e12, 1e15, 2.0 ** 53, 2.0 ** 63, 1e22, 1e300,
            999999.9999995, 0.9999995, 0.9999994999999999, 1 - 2.0 ** -53]
    vals += list(rng.random(60000))                                        # [0,1): the bulk of what is printed
    vals += list((rng.random(60000) - 0.5) * 10.0 ** rng.integers(-12, 14, 60000))
    vals += [struct.unpack("<d", struct.pack("<Q", int(b)))[0] for b in rng.integers(0, 2 ** 63, 40000)]
    # exact ties at the 6th decimal: k / 2^j with 7+ decimals
    vals += [float(k) / 2.0 ** j for j in range(7, 20) for k in range(1, 200, 2)]
    for v in vals:
        if np.isnan(v):
            continue
        assert capi.format_double(v, 6) == "%f" % v, repr(v)
        assert capi.format_double(-v, 6) == "%f" % -v, repr(-v)
        assert capi.format_double(v, 0) == "%.0f" % v, repr(v)
    assert capi.format_double(np.nan) == "-nan" and capi.format_double(-np.nan) == "-nan"
    assert capi.format_double(np.inf) == "inf" and capi.format_double(-np.inf, 0) == "-inf"


@pytest.mark.parametrize("name", ["f2_twochr_kb5", "f3_degenerate_ignmiss", "f5_minmaf", "f6_n500"])
@pytest.mark.parametrize("threads", [1, 3, 16])
def test_write_batch_threads_reproduce_oracle_text(name, threads, tmp_path):
    """ngsld_host_write_batch (threads format, one ordered write) == the oracle's TSV body, for a batch
    assembled from the golden records (keep / row_end / row_off as the engine would deliver them)."""
    import ctypes as C
    fx = Fixture(name)
    want = str(fx["orc_tsv_ext"]).splitlines(keepends=True)[1:]
    n, ns = len(fx["orc_s1"]), fx.n_sites
    std = np.zeros(n, dtype=capi.REC_STD)
    std["r2_ExpG"], std["D"], std["Dp"], std["r2"] = fx["orc_r2pear"], fx["orc_D"], fx["orc_Dp"], fx["orc_r2"]
    ext = np.zeros(n, dtype=capi.REC_EXT)
    ext["hap"], ext["n_ind_data"], ext["n_iter"] = fx["ref_hap"], fx["ref_n_ind_data"], fx["ref_n_iter"]
    maf = np.ascontiguousarray(fx["ref_maf"])
    items = capi.items_from_pairs(fx["orc_s1"], fx["orc_s2"], span=7 if threads == 3 else 64)
    p = tmp_path / "in.pos"
    p.write_text(fx.pos_text)
    L = capi.lib()
    h = C.c_void_p()
    err = C.create_string_buffer(256)
    assert L.ngsld_host_read_pos(str(p).encode(), int(fx.header), ns, C.byref(h), err, 256) == 0
    pd = np.ascontiguousarray(fx.pos_dist)
    b = capi.Batch(0, ns, n, len(items), items.ctypes.data, std.ctypes.data, ext.ctypes.data)
    out = tmp_path / "out.tsv"
    with open(out, "wb") as fh:
        rc = L.ngsld_host_write_batch(C.byref(b), h, pd.ctypes.data, maf.ctypes.data, threads, fh.fileno())
    L.ngsld_host_free_pos(h)
    assert rc == 0
    assert out.read_text().splitlines(keepends=True) == want


@pytest.mark.parametrize("name", [n for n in fixtures() if Fixture(n).text_mode])
def test_read_geno_text_matches_reference_reader(name, tmp_path):
    """Product text reader -> raw values; pushed through the reference branch's arithmetic (plain log, post_prob)
    they must reproduce the reference's own text reader output (ref_gl_log in the fixture), bit for bit."""
    from oracle import orc
    fx = Fixture(name)
    g, _ = fx.write_inputs(str(tmp_path))
    raw, is_log = capi.read_geno_text(g, fx.text_mode == "probs", fx.log_scale, fx.n_ind, fx.n_sites)
    assert raw.shape == (fx.n_sites, fx.n_ind, 3)
    assert is_log == (fx.log_scale if fx.text_mode == "probs" else True)
    want = fx["ref_gl_log"].copy()
    if fx.call_geno is None:
        got = raw.copy()
        for t in got.reshape(-1, 3):
            if not is_log:
                with np.errstate(divide="ignore"):
                    t[:] = np.log(t)
            orc.lib().orc_post_prob(orc.dp(t), orc.dp(t.copy()), 3)
        assert np.array_equal(got, want, equal_nan=True)


def test_read_geno_text_errors(tmp_path):
    p = tmp_path / "bad.geno.gz"
    with gzip.open(p, "wt") as fh:
        fh.write("id\ta\tb\n" + "s1\t0.1\t0.2\t0.7\t0.3\t0.3\t0.4\n" + "s2\t0.1\t0.2\n")
    with pytest.raises(capi.NgsldError) as e:
        capi.read_geno_text(str(p), True, False, 2, 2)
    assert "Less fields than expected" in e.value.msg
    with pytest.raises(capi.NgsldError) as e:
        capi.read_geno_text(str(p), True, False, 2, 3)
    assert "Less fields than expected" in e.value.msg or "premature EOF" in e.value.msg
    with gzip.open(p, "wt") as fh:
        fh.write("0\t1\t3\n")
    with pytest.raises(capi.NgsldError) as e:
        capi.read_geno_text(str(p), False, False, 3, 1)
    assert "Genotypes must be coded as {-1,0,1,2}" in e.value.msg
    with gzip.open(p, "wt") as fh:
        fh.write("0\t1\t2\n1\t1\t1\n")
    with pytest.raises(capi.NgsldError) as e:
        capi.read_geno_text(str(p), False, False, 3, 1)
    assert "not at EOF" in e.value.msg


def test_read_geno_text_threads_agree(tmp_path):
    fx = Fixture("f8_text_probs")
    g, _ = fx.write_inputs(str(tmp_path))
    L = capi.lib()
    L.ngsld_host_set_threads(1)
    a, la = capi.read_geno_text(g, True, False, fx.n_ind, fx.n_sites)
    L.ngsld_host_set_threads(7)
    b, lb = capi.read_geno_text(g, True, False, fx.n_ind, fx.n_sites)
    L.ngsld_host_set_threads(1)
    assert la == lb and np.array_equal(a, b, equal_nan=True)


def test_gz_output_writer_round_trip(tmp_path):
    """ngsld_host_gz_open / _close: what goes into the pipe comes out of the .gz file, whatever the write sizes; the file is
    a sequence of gzip members that gzip.open reads through."""
    import ctypes as C
    import gzip
    import os
    L = capi.lib()
    rng = np.random.default_rng(3)
    rows = [("chr1:%d\tchr1:%d\t%d\t%.6f\t%.6f\n" % (a, a + b, b, x, y)).encode()
            for a, b, x, y in zip(rng.integers(1, 10 ** 8, 400_000), rng.integers(1, 10 ** 5, 400_000), rng.random(400_000),
                                  rng.random(400_000))]
    data = b"".join(rows)                                     # ~17 MB: several 4 MiB blocks and a partial one
    for n_threads, piece in ((1, 1 << 20), (4, 7919), (8, len(data))):
        path = str(tmp_path / f"out{n_threads}.gz")
        h, fd = C.c_void_p(), C.c_int(-1)
        assert L.ngsld_host_gz_open(path.encode(), n_threads, C.byref(h), C.byref(fd)) == capi.OK
        off = 0
        while off < len(data):
            off += os.write(fd.value, data[off:off + piece])
        os.close(fd.value)
        assert L.ngsld_host_gz_close(h) == capi.OK
        with gzip.open(path, "rb") as fh:
            assert fh.read() == data
        assert os.path.getsize(path) < len(data) // 2
    # nothing written at all: an empty, valid file
    path = str(tmp_path / "empty.gz")
    h, fd = C.c_void_p(), C.c_int(-1)
    assert L.ngsld_host_gz_open(path.encode(), 2, C.byref(h), C.byref(fd)) == capi.OK
    os.close(fd.value)                                        # the write end is the caller's to close
    assert L.ngsld_host_gz_close(h) == capi.OK
    assert os.path.getsize(path) == 0


@pytest.mark.timeout(120)
def test_gz_output_write_error_does_not_block_the_producer():
    """A write error on the compressed file (/dev/full: ENOSPC on every write) must not leave the producer blocked on a full
    pipe: the rest of the stream is read and dropped, every byte is accepted, and ngsld_host_gz_close reports the failure."""
    import ctypes as C
    import os
    if not os.path.exists("/dev/full"):
        pytest.skip("no /dev/full here")
    L = capi.lib()
    data = os.urandom(1 << 20) * 48                           # 48 MiB: far beyond the slots (2 threads: 6 x 4 MiB) + the pipe
    h, fd = C.c_void_p(), C.c_int(-1)
    assert L.ngsld_host_gz_open(b"/dev/full", 2, C.byref(h), C.byref(fd)) == capi.OK
    off = 0
    while off < len(data):
        off += os.write(fd.value, data[off:off + (1 << 20)])
    os.close(fd.value)
    assert L.ngsld_host_gz_close(h) != capi.OK


def test_readers_survive_mutated_input(tmp_path):
    """Garbage in, an error (or the values) out -- never a crash: 400 byte-level mutations of a text genotype file and of a
    position file (bytes replaced, runs deleted, lines duplicated / truncated, NULs, very long tokens) through
    ngsld_host_read_geno_text and ngsld_host_read_pos.  tests/run_asan.sh runs this file on the AddressSanitizer / UBSan build
    of the host code, where an out-of-bounds read is a failure, not luck."""
    import gzip
    rng = np.random.default_rng(77)
    n_sites, n_ind = 12, 5
    probs = rng.random((n_sites, n_ind, 3))
    geno = "marker\ta1\ta2\t" + "\t".join(f"I{i}" for i in range(n_ind)) + "\n" + "".join(
        f"c_{s}\tA\tC\t" + "\t".join(repr(float(x)) for x in probs[s].reshape(-1)) + "\n" for s in range(n_sites))
    pos = "".join(f"chr{1 + s // 6}\t{10 * (s % 6 + 1)}\n" for s in range(n_sites))

    def mutate(text: str) -> bytes:
        b = bytearray(text.encode())
        for _ in range(int(rng.integers(1, 6))):
            kind = int(rng.integers(0, 6))
            at = int(rng.integers(0, max(1, len(b))))
            if kind == 0 and b:
                b[at] = int(rng.integers(0, 256))
            elif kind == 1:
                del b[at:at + int(rng.integers(1, 40))]
            elif kind == 2:
                b[at:at] = b[at:at + int(rng.integers(1, 60))]
            elif kind == 3:
                b[at:at] = bytes([0])
            elif kind == 4:
                b[at:at] = b"9" * int(rng.integers(100, 5000))
            else:
                del b[at:]
        return bytes(b)

    outcomes = {"ok": 0, "error": 0}
    for k in range(200):
        g = tmp_path / f"m{k}.geno.gz"
        with gzip.open(g, "wb") as fh:
            fh.write(mutate(geno))
        try:
            raw, _ = capi.read_geno_text(str(g), True, bool(k % 2), n_ind, n_sites)
            assert raw.shape == (n_sites, n_ind, 3)
            outcomes["ok"] += 1
        except (capi.NgsldError, RuntimeError, ValueError):
            outcomes["error"] += 1
        p = tmp_path / f"m{k}.pos"
        p.write_bytes(mutate(pos))
        try:
            pd, labels = capi.read_pos(str(p), bool(k % 3 == 0), n_sites)
            assert len(pd) == n_sites and len(labels) == n_sites
            outcomes["ok"] += 1
        except (capi.NgsldError, RuntimeError, ValueError):
            outcomes["error"] += 1
    assert outcomes["ok"] + outcomes["error"] == 400 and outcomes["error"] > 0
    print("mutated inputs:", outcomes)


# ---- text genotype files that are not "headers, then rows": three readers, one outcome --------------------------------
_REF_TEXT_CHILD = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %r)
from oracle import orc
R = orc.ref()
path, in_probs, n_ind, n_sites, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
gl = np.full((n_sites, n_ind, 3), np.nan)
R.ref_read_geno_text(path.encode(), in_probs, 0, n_ind, n_sites, orc.dp(gl))    # (an error ends the process through error())
np.save(out, gl)
""" % capi.REPO_DIR

_ROWS = ["0\t1\t2", "1\t1\t0", "2\t0\t1", "-1\t2\t2", "0\t0\t1"]
IRREGULAR = [  # (what, file text, n_sites)
    ("a header line repeated among the rows", "\n".join(_ROWS[:2] + ["marker\ta\tb"] + _ROWS[2:]) + "\n", 5),
    ("two header lines at the top, the second with a few numbers", "id\tx\ty\n1\tb\tc\n" + "\n".join(_ROWS) + "\n", 5),
    ("a line of words as the very last line", "\n".join(_ROWS + ["end\tof\tfile"]) + "\n", 5),
    ("a genotype of 3 in row 2 AND a row too many", "\n".join([_ROWS[0], "0\t3\t1"] + _ROWS[1:] + ["1\t1\t1"]) + "\n", 5),
    ("a genotype of 3 in row 4 AND a row too few", "\n".join(_ROWS[:3] + ["0\t3\t1"]) + "\n", 5),
    ("a short row AND a row too many", "\n".join(_ROWS[:2] + ["1\t1"] + _ROWS[2:] + ["1\t1\t1"]) + "\n", 5),
    ("every row short (n_ind given too large)", "\n".join("\t".join(r.split("\t")[:2]) for r in _ROWS) + "\n", 5),
    ("an empty line after the rows", "\n".join(_ROWS) + "\n\n", 5),
    ("an empty line among the rows and as many rows as sites", "\n".join(_ROWS[:2] + [""] + _ROWS[2:]) + "\n", 5),
    ("a header line among the rows and a row too few", "\n".join(_ROWS[:2] + ["marker\ta\tb"] + _ROWS[2:4]) + "\n", 5),
    ("rows, then a header line, then surplus rows", "\n".join(_ROWS + ["marker\ta\tb", "1\t1\t1"]) + "\n", 5),
]


@pytest.mark.parametrize("what,text,n_sites", IRREGULAR, ids=[c[0] for c in IRREGULAR])
def test_irregular_text_genotype_files_read_like_the_reference(what, text, n_sites, tmp_path):
    """The reference's reader takes a text file line by line (read_data.cpp:46-109): headers are whatever has too few numbers while
    no site is stored, a line without any number is skipped WHEREVER it stands, a row's own errors come up in row order, the
    end-of-file checks last.  The product's reader (parallel pass for well-formed files, the reference's walk for the rest) and
    the oracle's must give the reference's values, or fail where it fails with its message."""
    import ctypes as C
    import subprocess
    import sys
    from oracle import orc
    if orc.ref() is None:
        pytest.skip("oracle/_ref not built")
    p = tmp_path / "in.geno"
    p.write_text(text)
    out = str(tmp_path / "ref.npy")
    r = subprocess.run([sys.executable, "-c", _REF_TEXT_CHILD, str(p), "0", "3", str(n_sites), out], capture_output=True, text=True,
                       timeout=120)
    gl_orc = np.empty((n_sites, 3, 3))
    err = C.create_string_buffer(256)
    rc_orc = orc.lib().orc_read_geno_text(str(p).encode(), 0, 0, 3, n_sites, orc.dp(gl_orc), err, 256)
    try:
        raw, is_log = capi.read_geno_text(str(p), False, False, 3, n_sites)
        msg_hip = None
    except capi.NgsldError as e:
        raw, msg_hip = None, e.msg
    if r.returncode == 0:
        want = np.load(out)
        assert rc_orc == 0 and np.array_equal(gl_orc, want), (what, err.value)
        assert raw is not None, (what, msg_hip)
        got = raw.copy()
        for t in got.reshape(-1, 3):
            orc.lib().orc_post_prob(orc.dp(t), orc.dp(t.copy()), 3)   # (text reader output is log scale: is_log)
        assert is_log and np.array_equal(got, want), what
    else:
        ref_msg = next((ln for ln in r.stderr.splitlines() if "ERROR" in ln), "")
        assert rc_orc != 0 and raw is None, (what, ref_msg, rc_orc, msg_hip)
        assert err.value.decode() == msg_hip, (what, err.value, msg_hip)
        if "empty line" not in msg_hip:   # (the one outcome of its own -- the reference goes on with the site unfilled)
            assert msg_hip in ref_msg, (what, ref_msg, msg_hip)


@pytest.mark.parametrize("what,last", [("geno", 500), ("pos", 700)])
def test_mutated_files_read_like_the_reference(what, last):
    """tools/reader_fuzz.py: small text genotype / positions files, mutated a few times each, through the reference's compiled
    reader (forked: its errors end the process), the oracle's and the product's -- the reference's values, or its message.  (The
    documented exceptions -- an empty line in a site's place, a last positions line without newline, the header line on which the
    reference's loop never ends -- are counted by the tool; 3,000 + 4,000 cases ran clean when the readers were last changed.)"""
    import subprocess
    import sys
    from oracle import orc
    if orc.ref() is None:
        pytest.skip("oracle/_ref not built")
    if "asan" in os.environ.get("LD_PRELOAD", ""):   # (the sanitizer's allocator aborts the REFERENCE's reader on its own overruns)
        pytest.skip("under tests/run_asan.sh the reference's reader is not a usable yardstick")
    r = subprocess.run([sys.executable, os.path.join(capi.REPO_DIR, "tools", "reader_fuzz.py"), what, "0", str(last)], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and ", 0 differ;" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


def test_product_formatter_against_the_references_own_fprintf_lines():
    """ngsld_host_format_pair (the rows the drop-in binary writes when they are not formatted on the device) against
    ngsLD.cpp:296-351 compiled from the reference (oracle/_ref ref_format_row: hap-derived maf, D / D' / r2, the float chi2 and
    both fprintf formats) on 4,000 haplotype vectors -- simplex points, exact zeros, nearly monomorphic sites, sums off by
    rounding, NaN / inf entries -- with inf distances, NaN allele frequencies, r2_ExpG values on rounding points.  The oracle's
    formatter is held to the same lines in test_oracle_vs_ref.py; this closes the triangle without going through the oracle.
    (NaN is given with the sign the pipeline produces -- x86's 0/0, which the kernels reproduce: printf writes "-nan".)"""
    import ctypes as C
    from oracle import orc
    from test_oracle_vs_ref import _hap_vectors
    ref = orc.ref()
    if ref is None or not hasattr(ref, "ref_format_row"):
        pytest.skip("oracle/_ref predates ref_format_row (oracle/build_ref.sh)")
    L = orc.lib()
    rng = np.random.default_rng(21)
    maf, maf_nan = rng.uniform(0, 0.5, 2), np.array([-np.nan, 0.25])
    l1, l2 = "chr1:1234\tsnpA", "chr22:99999999"
    std, ext = np.zeros(1, dtype=capi.REC_STD), np.zeros(1, dtype=capi.REC_EXT)
    n = 0
    for k, h in enumerate(_hap_vectors(4_000, 33)):
        h = np.ascontiguousarray(h)
        mm = maf_nan if k % 97 == 0 else maf
        dist = float("inf") if k % 5 == 0 else float(rng.integers(1, 10 ** 9))
        r2p = [float(rng.uniform()), -float("nan"), 0.0, 1.0, 0.9999995, 1e-7, 0.0078125][k % 7]
        n_data, n_iter = int(rng.integers(0, 5000)), int(rng.integers(0, 101))
        D, Dp, r2, hm, c = np.zeros(1), np.zeros(1), np.zeros(1), np.zeros(2), C.c_float()
        L.orc_pair_stats(orc.dp(h), orc.dp(D), orc.dp(Dp), orc.dp(r2), orc.dp(hm), C.byref(c))
        if any(np.isnan(v) and not np.signbit(v) for v in (D[0], Dp[0], r2[0], *h)):
            continue   # (a NaN of the other sign: only an input NaN could produce one, and the readers refuse those)
        std["r2_ExpG"], std["D"], std["Dp"], std["r2"] = r2p, D[0], Dp[0], r2[0]
        ext["hap"], ext["n_ind_data"], ext["n_iter"] = h, n_data, n_iter
        for extend in (0, 1):
            b = C.create_string_buffer(2048)
            nb = ref.ref_format_row(b, 2048, l1.encode(), l2.encode(), dist, r2p, orc.dp(h), n_data, float(mm[0]), float(mm[1]), n_iter, extend)
            got = capi.format_pair(l1, l2, dist, std, ext if extend else None, float(mm[0]), float(mm[1]))
            assert nb > 0 and got.encode() == b.raw[:nb], (k, got, b.raw[:nb])
            n += 1
    assert n > 7000
