"""The reference's OWN program flow, run here on the fixtures (CPU).

oracle/build_ref.sh compiles ngsLD.cpp's main() and calc_pair_LD as they stand but for the statements that need GSL: the six
gsl_rng statements and the --rnd_sample block are dropped (fixtures with --rnd_sample are skipped), the one pearson_r call is
replaced by a lookup of the value the caller supplies for that pair.  ref_main(argc, argv) is then the reference's argument
parsing, file checks, reader, call_geno loop, est_maf loop, exp / expected genotypes, positions and labels, thread pool,
per-row walk, EM, statistics and fprintf on real files.  Its TSV must be the oracle CLI's byte for byte (one thread: same
order; several: as sorted lines) -- the assembly of the pieces the other tests pin one by one.  The r2_ExpG column is the
oracle's own (GSL), handed in: the one column this does not pin."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import orc
from util import Fixture, fixtures

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = orc.ref()
pytestmark = pytest.mark.skipif(ref is None or not hasattr(ref, "ref_main"),
                                reason="oracle/_ref predates ref_main (rebuild with oracle/build_ref.sh)")

_CHILD = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %r)
from oracle import orc
R = orc.ref()
tab = np.load(sys.argv[1])
first, s2, val = (np.ascontiguousarray(tab[k]) for k in ("first", "s2", "val"))
R.ref_set_r2pear.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
R.ref_set_r2pear(first.ctypes.data, s2.ctypes.data, val.ctypes.data, len(first) - 1)
argv = [b"ngsLD"] + [a.encode() for a in sys.argv[2:]]
arr = (C.c_char_p * (len(argv) + 1))(*argv, None)
R.ref_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
sys.exit(R.ref_main(len(argv), arr))
""" % REPO

NAMES = [n for n in fixtures() if Fixture(n).rnd_sample >= 1 and Fixture(n).n_ind <= 500]


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("extend,threads", [(False, 1), (True, 1), (True, 3)])
def test_reference_program_flow_writes_the_oracles_tsv(name, extend, threads, tmp_path):
    fx = Fixture(name)
    d = str(tmp_path)
    g, p = fx.write_inputs(d)
    rec = fx.oracle().run()
    first = np.zeros(fx.n_sites + 1, dtype=np.uint64)
    np.add.at(first, rec["s1"].astype(np.int64) + 1, 1)
    first = np.cumsum(first).astype(np.uint64)
    tab = os.path.join(d, "r2.npz")
    np.savez(tab, first=first, s2=rec["s2"].astype(np.uint64), val=rec["r2pear"].astype(np.float64))
    base = ["--geno", g, "--n_ind", str(fx.n_ind), "--n_sites", str(fx.n_sites), "--verbose", "0"]
    if p:
        base += ["--posH" if fx.header else "--pos", p]
    flags = base + fx.cli_flags(extend)
    out_ref = os.path.join(d, "ref.tsv")
    r = subprocess.run([sys.executable, "-c", _CHILD, tab, *flags, "--n_threads", str(threads), "--out", out_ref],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = subprocess.run([orc.ORC_CLI, *flags], check=True, capture_output=True, text=True).stdout
    got = open(out_ref).read()
    if threads == 1:
        assert got == want, "the reference's program flow and the oracle CLI write different bytes"
    else:
        gl, wl = got.splitlines(keepends=True), want.splitlines(keepends=True)
        assert gl[0] == wl[0] and sorted(gl[1:]) == sorted(wl[1:])
    tag = "ext" if extend else "std"
    if f"orc_tsv_{tag}_md5" in fx:   # and the golden md5 (sorted body, as the reference's own test sorts: examples/test.sh:16)
        lines = got.splitlines(keepends=True)
        assert hashlib.md5((lines[0] + "".join(sorted(lines[1:]))).encode()).hexdigest() == str(fx[f"orc_tsv_{tag}_md5"])
