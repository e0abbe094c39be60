"""Shared helpers of the test-suite: golden fixtures, comparison rules."""
from __future__ import annotations

import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
TOL = 1e-9          # BASELINE.json north_star: results match the reference within 1e-9 on r2 / D / D'
MAF_TOL = 1e-12     # est_maf must match to 1e-12 (it seeds the EM)


def fixtures() -> list[str]:
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


class Fixture:
    def __init__(self, name: str):
        self.name = name
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.z = z
        self.raw = z["raw"]
        self.n_sites, self.n_ind = self.raw.shape[:2]
        self.has_pos = bool(z["has_pos"])
        self.pos_text = str(z["pos_text"]) if self.has_pos else None
        self.header = bool(z["header"])
        self.log_scale = bool(z["log_scale"])
        self.ignore_miss = bool(z["ignore_miss"])
        self.max_kb, self.max_snp, self.min_maf = int(z["max_kb"]), int(z["max_snp"]), float(z["min_maf"])
        self.text_mode = str(z["text_mode"]) if "text_mode" in z.files and str(z["text_mode"]) else None
        self.geno_text = str(z["geno_text"]) if self.text_mode else None
        cg = z["call_geno"] if "call_geno" in z.files else np.zeros(0)
        self.call_geno = (float(cg[0]), float(cg[1])) if len(cg) == 2 else None
        self.rnd_sample = float(z["rnd_sample"]) if "rnd_sample" in z.files else 1.0
        self.seed = int(z["seed"]) if "seed" in z.files else 0
        if self.text_mode == "called":
            self.n_sites, self.n_ind = self.raw.shape
        self.pos_dist = z["ref_pos_dist"] if self.has_pos else None
        self.labels = [str(x) for x in z["ref_labels"]] if self.has_pos else None

    def oracle(self, n_threads: int = 2):
        """The oracle on this fixture's input, through the same reader the fixture's mode uses."""
        import ctypes as C
        import tempfile
        from oracle import orc
        kw = dict(ignore_miss_data=self.ignore_miss, max_kb_dist=self.max_kb, max_snp_dist=self.max_snp,
                  min_maf=self.min_maf, n_threads=n_threads, call_geno=self.call_geno, rnd_sample=self.rnd_sample,
                  seed=self.seed)
        if not self.text_mode:
            return orc.Oracle(self.raw, self.pos_dist, log_scale=self.log_scale, **kw)
        with tempfile.TemporaryDirectory() as d:
            g, _ = self.write_inputs(d)
            gl = np.empty((self.n_sites, self.n_ind, 3))
            err = C.create_string_buffer(256)
            rc = orc.lib().orc_read_geno_text(g.encode(), int(self.text_mode == "probs"), int(self.log_scale), self.n_ind,
                                              self.n_sites, orc.dp(gl), err, 256)
            assert rc == 0, err.value
        return orc.Oracle(gl, self.pos_dist, already_normalised_log=True, **kw)

    def engine_load(self, engine, tmp_dir: str):
        """Feed this fixture to the HIP engine the way the CLI would (binary raw, or text reader + text semantics)."""
        from ngsld_amd import capi
        if not self.text_mode:
            engine.set_geno_raw(self.raw, log_scale=self.log_scale, ignore_miss_data=self.ignore_miss,
                                call_geno=self.call_geno)
        else:
            g, _ = self.write_inputs(tmp_dir)
            raw, is_log = capi.read_geno_text(g, self.text_mode == "probs", self.log_scale, self.n_ind, self.n_sites)
            engine.set_geno_raw(raw, log_scale=is_log, ignore_miss_data=self.ignore_miss, text=True,
                                call_geno=self.call_geno)
        engine.set_pos_dist(self.pos_dist)

    def __getitem__(self, k):
        return self.z[k]

    def __contains__(self, k):
        return k in self.z.files

    def write_inputs(self, d: str) -> tuple[str, str | None]:
        if self.text_mode:
            import gzip
            g = os.path.join(d, self.name + ".geno.gz")
            with gzip.open(g, "wt") as fh:
                fh.write(self.geno_text)
        else:
            g = os.path.join(d, self.name + ".glf")
            self.raw.tofile(g)
        p = None
        if self.has_pos:
            p = os.path.join(d, self.name + ".pos")
            with open(p, "w") as fh:
                fh.write(self.pos_text)
        return g, p

    def cli_flags(self, extend: bool) -> list[str]:
        f = ["--max_kb_dist", str(self.max_kb), "--max_snp_dist", str(self.max_snp), "--min_maf", repr(self.min_maf)]
        if self.log_scale:
            f.append("--log_scale")
        if self.ignore_miss:
            f.append("--ignore_miss_data")
        if self.text_mode == "probs" or self.call_geno:
            f.append("--probs")
        if self.rnd_sample < 1:
            f += ["--rnd_sample", repr(self.rnd_sample), "--seed", str(self.seed)]
        if self.call_geno:
            f += ["--call_geno", "--N_thresh", repr(self.call_geno[0]), "--call_thresh", repr(self.call_geno[1])]
        if extend:
            f.append("--extend_out")
        return f


def close(a, b, tol=TOL):
    """|a-b| <= tol, NaN == NaN, inf == inf of the same sign."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        return (np.isnan(a) & np.isnan(b)) | (np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))) | \
               (np.abs(a - b) <= tol)


def degenerate_rows(hap_maf: np.ndarray) -> np.ndarray:
    """Pairs with a site whose hap-derived allele frequency is 0 or 1 up to rounding noise (or NaN): D' and r2 are
    0/0-type expressions there, decided by the last bit of the reference's own accumulation order.  The engine replays
    those pairs in that order (ngsld.h, exact-order replay), so they are held to BIT equality, not to a tolerance."""
    hm = np.asarray(hap_maf)
    with np.errstate(invalid="ignore"):
        return np.any((np.abs(hm) < 1e-12) | (np.abs(1 - hm) < 1e-12) | np.isnan(hm), axis=1)


def pearson_tolerance(gl: np.ndarray, s1: np.ndarray, s2: np.ndarray, tol=TOL) -> np.ndarray:
    """Kept for callers of the round-1 interface: the per-pair allowance on r2_ExpG is gone (sites whose expected
    genotypes are constant up to rounding are replayed in the reference's order); every pair is held to `tol`."""
    return np.full(len(np.asarray(s1)), tol)


def same_bits(a, b) -> np.ndarray:
    """Element-wise: equal bit patterns, any NaN equal to any NaN."""
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))


REPORT = {"pairs": 0, "over_tol": 0, "degenerate": 0, "max_diff": 0.0}   # running totals, printed by conftest at exit


def check_records(std, ext, want: dict, tol=TOL, pearson_tol=None, exact_degenerate=True):
    """HIP records vs expected columns (hap, n_iter, n_ind_data, D, Dp, r2, r2pear, hap_maf).
    Bars: nIter and sample_size equal; hap, D, D', r2, r2_ExpG within `tol` (1e-9) on EVERY pair, NaN only where the
    reference is NaN, inf only where it is inf of the same sign -- no widened tolerance anywhere; pairs with a
    monomorphic site (hap_maf 0 or 1 up to rounding: the reference's -nan / 0.000000 / inf outcomes) BIT-equal
    (exact_degenerate: the engine had the caller's raw values to replay from).  Returns the number of pairs."""
    assert np.array_equal(ext["n_ind_data"], want["n_ind_data"]), "sample_size must be bit-exact"
    bad = np.flatnonzero(ext["n_iter"] != want["n_iter"])
    assert len(bad) == 0, f"nIter differs on {len(bad)} pairs, first {bad[:5]}"
    degen = degenerate_rows(want["hap_maf"])
    over = np.zeros(len(degen), dtype=bool)
    for name, got, exp in (("hap", ext["hap"], want["hap"]), ("D", std["D"], want["D"]), ("Dp", std["Dp"], want["Dp"]),
                           ("r2", std["r2"], want["r2"]), ("r2_ExpG", std["r2_ExpG"], want["r2pear"])):
        ok = close(got, exp, tol)
        assert np.all(ok), (f"{name}: {np.count_nonzero(~ok)} of {ok.size} outside {tol}; got "
                            f"{np.asarray(got)[~ok][:3]} want {np.asarray(exp)[~ok][:3]}")
        with np.errstate(invalid="ignore"):
            d = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(exp, dtype=np.float64))
        d = d if d.ndim == 1 else np.nanmax(np.where(np.isnan(d), 0.0, d), axis=1)
        d = np.where(np.isfinite(d), d, 0.0)
        over |= d > tol
        REPORT["max_diff"] = max(REPORT["max_diff"], float(d.max()) if d.size else 0.0)
        if exact_degenerate and degen.any() and name != "r2_ExpG":
            g, e = np.asarray(got)[degen], np.asarray(exp)[degen]
            sb = same_bits(g, e)
            assert np.all(sb), (f"{name}: {np.count_nonzero(~sb)} degenerate pairs are not the reference's bits; got "
                                f"{g[~sb][:3]} want {e[~sb][:3]}")
    REPORT["pairs"] += len(degen)
    REPORT["over_tol"] += int(over.sum())
    REPORT["degenerate"] += int(degen.sum())
    return len(degen)


# ---- the reference's own program (oracle/_ref: ngsLD.cpp's main() and calc_pair_LD compiled as they stand, minus the GSL
# statements -- oracle/build_ref.sh) run on real files in a child process (an invalid argument ends the process through error()).
# Its one GSL column, r2_ExpG, is looked up per pair in the table the caller supplies (the oracle's values).
_REF_CHILD = r"""
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
""" % os.path.dirname(HERE)


def have_ref_program() -> bool:
    from oracle import orc
    r = orc.ref()
    return r is not None and hasattr(r, "ref_main")


def run_ref_program(rec, n_sites: int, flags: list[str], out_path: str, work_dir: str, threads: int = 1, timeout: int = 600):
    """ref_main(argv) with `flags` + --n_threads + --out; rec = the oracle's records of the same run (s1, s2 increasing: where
    the r2_ExpG of every pair is looked up).  Returns the CompletedProcess."""
    import subprocess
    import sys
    first = np.zeros(n_sites + 1, dtype=np.uint64)
    np.add.at(first, rec["s1"].astype(np.int64) + 1, 1)
    first = np.cumsum(first).astype(np.uint64)
    tab = os.path.join(work_dir, "r2_table.npz")
    np.savez(tab, first=first, s2=rec["s2"].astype(np.uint64), val=rec["r2pear"].astype(np.float64))
    return subprocess.run([sys.executable, "-c", _REF_CHILD, tab, *flags, "--n_threads", str(threads), "--out", out_path],
                          capture_output=True, text=True, timeout=timeout)


# ---- the same program with the library plugged in: oracle/_ref/libngsld_ref_hip.so = the reference's main() with its thread-pool
# section (ngsLD.cpp:153-198) replaced by integration/ngsld_binding.h's one call, compiled by oracle/build_ref.sh
_REF_HIP_CHILD = r"""
import ctypes as C, os, sys
R = C.CDLL(os.path.join(%r, "oracle", "_ref", "libngsld_ref_hip.so"))
argv = [b"ngsLD"] + [a.encode() for a in sys.argv[1:]]
arr = (C.c_char_p * (len(argv) + 1))(*argv, None)
R.ref_main_hip.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
sys.exit(R.ref_main_hip(len(argv), arr))
""" % os.path.dirname(HERE)


def have_patched_ref_program() -> bool:
    return os.path.exists(os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libngsld_ref_hip.so"))


def run_patched_ref_program(flags: list[str], out_path: str, threads: int = 1, timeout: int = 600, env: dict | None = None):
    """ref_main_hip(argv): the reference's own main(), pair loop on the device through the C-ABI.  Returns the CompletedProcess."""
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", _REF_HIP_CHILD, *flags, "--n_threads", str(threads), "--out", out_path],
                          capture_output=True, text=True, timeout=timeout, env=e)
