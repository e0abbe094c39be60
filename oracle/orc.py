"""ctypes binding of the CPU ORACLE (oracle/liborc.so) and, when present, of the reference's own
functions (oracle/_ref/libngsld_ref.so).  TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBORC = os.path.join(HERE, "liborc.so")
ORC_CLI = os.path.join(HERE, "ngsld_oracle")
LIBREF = os.path.join(HERE, "_ref", "libngsld_ref.so")

c_double_p = C.POINTER(C.c_double)


class OrcParams(C.Structure):
    _fields_ = [
        ("in_geno", C.c_char_p), ("in_logscale", C.c_int), ("n_ind", C.c_uint64), ("n_sites", C.c_uint64),
        ("in_pos", C.c_char_p), ("in_pos_header", C.c_int), ("max_kb_dist", C.c_uint64),
        ("max_snp_dist", C.c_uint64), ("min_maf", C.c_double), ("ignore_miss_data", C.c_int),
        ("extend_out", C.c_int), ("n_threads", C.c_int), ("rnd_sample", C.c_double), ("seed", C.c_uint64),
        ("geno_lkl", c_double_p), ("maf", c_double_p), ("expected_geno", c_double_p), ("pos_dist", c_double_p),
        ("labels", C.POINTER(C.c_char_p)),
    ]


PAIR_DTYPE = np.dtype([
    ("s1", "<u8"), ("s2", "<u8"), ("dist", "<f8"), ("r2pear", "<f8"), ("D", "<f8"), ("Dp", "<f8"), ("r2", "<f8"),
    ("n_ind_data", "<u8"), ("hap", "<f8", (4,)), ("hap_maf", "<f8", (2,)), ("chi2", "<f4"), ("_pad", "<u4"),
    ("n_iter", "<u8")])


def build(force: bool = False) -> None:
    srcs = [os.path.join(HERE, f) for f in ("ngsld_oracle.c", "ngsld_oracle.h", "ngsld_oracle_main.c")]
    if force or not (os.path.exists(LIBORC) and os.path.exists(ORC_CLI)) or \
            min(os.path.getmtime(LIBORC), os.path.getmtime(ORC_CLI)) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def build_ref() -> bool:
    """Build oracle/_ref from /root/reference when that tree exists; returns whether the .so is available."""
    if os.path.isdir("/root/reference/shared"):
        subprocess.check_call([os.path.join(HERE, "build_ref.sh")], stdout=subprocess.DEVNULL)
    return os.path.exists(LIBREF)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIBORC)
        assert C.sizeof(OrcParams) > 0 and PAIR_DTYPE.itemsize == 128
        L.orc_logsum.restype = C.c_double
        L.orc_logsum.argtypes = [c_double_p, C.c_uint64]
        L.orc_est_maf.restype = C.c_double
        L.orc_est_maf.argtypes = [C.c_uint64, c_double_p, C.c_int]
        L.orc_pair_freq_iter.restype = C.c_uint64
        L.orc_pair_freq_iter.argtypes = [c_double_p, c_double_p, c_double_p, C.c_uint64, C.c_int, C.POINTER(C.c_int)]
        L.orc_haplo_freq.restype = C.c_uint64
        L.orc_haplo_freq.argtypes = [c_double_p, C.POINTER(C.c_uint64), c_double_p, c_double_p, C.c_double,
                                     C.c_double, C.c_uint64, C.c_int, C.POINTER(C.c_int)]
        L.orc_correlation.restype = C.c_double
        L.orc_correlation.argtypes = [c_double_p, c_double_p, C.c_uint64]
        L.orc_pearson_r2.restype = C.c_double
        L.orc_pearson_r2.argtypes = [c_double_p, c_double_p, C.c_uint64]
        L.orc_normalise_raw.restype = C.c_int
        L.orc_normalise_raw.argtypes = [c_double_p, C.c_int, C.c_uint64, C.c_uint64, c_double_p]
        L.orc_read_geno_bin.restype = C.c_int
        L.orc_read_geno_bin.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_uint64, c_double_p, C.c_char_p,
                                        C.c_size_t]
        L.orc_read_geno_text.restype = C.c_int
        L.orc_read_geno_text.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, c_double_p, C.c_char_p,
                                         C.c_size_t]
        L.orc_call_geno.restype = None
        L.orc_call_geno.argtypes = [c_double_p, C.c_double, C.c_double]
        L.orc_call_geno_all.restype = None
        L.orc_call_geno_all.argtypes = [C.POINTER(OrcParams), C.c_double, C.c_double]
        L.orc_post_prob.restype = None
        L.orc_post_prob.argtypes = [c_double_p, c_double_p, C.c_uint64]
        L.orc_preprocess.restype = None
        L.orc_preprocess.argtypes = [C.POINTER(OrcParams)]
        L.orc_read_pos.restype = C.c_int
        L.orc_read_pos.argtypes = [C.POINTER(OrcParams), C.c_char_p, C.c_size_t]
        L.orc_free_pos.restype = None
        L.orc_free_pos.argtypes = [C.POINTER(OrcParams)]
        L.orc_run.restype = C.c_uint64
        L.orc_run.argtypes = [C.POINTER(OrcParams), C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                              C.POINTER(C.c_int)]
        L.orc_taus_set.restype = None
        L.orc_taus_set.argtypes = [C.c_void_p, C.c_ulong]
        L.orc_taus_get.restype = C.c_uint32
        L.orc_taus_get.argtypes = [C.c_void_p]
        L.orc_row_seeds.restype = None
        L.orc_row_seeds.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        L.orc_bench.restype = C.c_uint64
        L.orc_bench.argtypes = [C.POINTER(OrcParams), C.c_uint64, C.c_uint64, c_double_p, C.POINTER(C.c_uint64)]
        L.orc_bench_pearson.restype = C.c_uint64
        L.orc_bench_pearson.argtypes = [C.POINTER(OrcParams), C.c_uint64, C.c_uint64, c_double_p]
        L.orc_row_end.restype = C.c_uint64
        L.orc_row_end.argtypes = [C.POINTER(OrcParams), C.c_uint64]
        L.orc_pair_stats.restype = None
        L.orc_pair_stats.argtypes = [c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(C.c_float)]
        L.orc_format_header.restype = C.c_long
        L.orc_format_header.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
        L.orc_format_pair.restype = C.c_long
        L.orc_format_pair.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(OrcParams), C.c_void_p]
        _lib = L
    return _lib


_ref = None


def ref():
    """The reference's own functions (None when oracle/_ref was not built / did not travel)."""
    global _ref
    if _ref is None and os.path.exists(LIBREF):
        R = C.CDLL(LIBREF)
        R.ref_logsum.restype = C.c_double
        R.ref_logsum.argtypes = [c_double_p, C.c_uint64]
        R.ref_est_maf.restype = C.c_double
        R.ref_est_maf.argtypes = [C.c_uint64, c_double_p, C.c_int]
        R.ref_pair_freq_iter.restype = C.c_uint64
        R.ref_pair_freq_iter.argtypes = [c_double_p, c_double_p, c_double_p, C.c_uint64, C.c_int]
        R.ref_haplo_freq.restype = C.c_uint64
        R.ref_haplo_freq.argtypes = [c_double_p, C.POINTER(C.c_uint64), c_double_p, c_double_p, C.c_double,
                                     C.c_double, C.c_uint64, C.c_int]
        R.ref_read_geno_bin.restype = C.c_int
        R.ref_read_geno_bin.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_uint64, c_double_p]
        R.ref_read_geno_text.restype = C.c_int
        R.ref_read_geno_text.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, c_double_p]
        R.ref_call_geno.restype = None
        R.ref_call_geno.argtypes = [c_double_p, C.c_double, C.c_double]
        R.ref_preprocess.restype = None
        R.ref_preprocess.argtypes = [c_double_p, C.c_uint64, C.c_uint64, C.c_int, c_double_p, c_double_p]
        R.ref_read_dist.restype = C.c_int
        R.ref_read_dist.argtypes = [C.c_char_p, C.c_int, C.c_uint64, c_double_p]
        R.ref_read_labels.restype = C.c_uint64
        R.ref_read_labels.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_uint64, C.c_uint64]
        if hasattr(R, "ref_pair_stats"):   # the GSL-free lines of ngsLD.cpp (build_ref.sh cuts them out by anchor)
            R.ref_pair_stats.restype = None
            R.ref_pair_stats.argtypes = [c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(C.c_float)]
            R.ref_format_row.restype = C.c_long
            R.ref_format_row.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_double, C.c_double, c_double_p,
                                         C.c_uint64, C.c_double, C.c_double, C.c_uint64, C.c_int]
            R.ref_print_header.restype = C.c_long
            R.ref_print_header.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
            R.ref_walk.restype = C.c_uint64
            R.ref_walk.argtypes = [C.c_uint64, c_double_p, c_double_p, C.c_uint64, C.c_uint64, C.c_double, C.c_uint64,
                                   C.POINTER(C.c_uint64), c_double_p, C.c_uint64]
        if hasattr(R, "ref_parse_args"):   # parse_args.cpp compiled whole (tests/test_cli_args_vs_ref.py)
            R.ref_parse_args.restype = C.c_int
            R.ref_parse_args.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        if hasattr(R, "ref_bench_haplo_freq"):
            R.ref_bench_haplo_freq.restype = C.c_uint64
            R.ref_bench_haplo_freq.argtypes = [c_double_p, c_double_p, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64,
                                               C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64),
                                               c_double_p]
        _ref = R
    return _ref


def dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


class Oracle:
    """In-memory run of the oracle: raw GL array [n_sites, n_ind, 3] (+ pos_dist) -> pair records."""

    def __init__(self, raw_gl: np.ndarray, pos_dist: np.ndarray | None = None, log_scale: bool = False,
                 ignore_miss_data: bool = False, max_kb_dist: int = 0, max_snp_dist: int = 0, min_maf: float = 0.0,
                 n_threads: int = 1, already_normalised_log: bool = False, call_geno: tuple | None = None,
                 rnd_sample: float = 1.0, seed: int = 0):
        """raw_gl: what the binary reader would read, or (already_normalised_log) the log-normalised output of
        a reader.  call_geno = (N_thresh, call_thresh) applies ngsLD.cpp:92-98 before est_maf."""
        L = lib()
        raw_gl = np.ascontiguousarray(raw_gl, dtype=np.float64)
        self.n_sites, self.n_ind = raw_gl.shape[0], raw_gl.shape[1]
        if already_normalised_log:
            self.gl = raw_gl.copy()
        else:
            self.gl = np.empty_like(raw_gl)
            rc = L.orc_normalise_raw(dp(raw_gl), int(log_scale), self.n_ind, self.n_sites, dp(self.gl))
            if rc:
                raise ValueError("NaN found! Is the file format correct?")
        if call_geno is not None:
            for k in range(self.n_sites):
                for i in range(self.n_ind):
                    L.orc_call_geno(dp(self.gl[k, i]), float(call_geno[0]), float(call_geno[1]))
        self.gl_log = self.gl.copy()
        self.maf = np.empty(self.n_sites)
        self.expg = np.empty((self.n_sites, self.n_ind))
        if pos_dist is None:
            pos_dist = np.full(self.n_sites, np.inf)
        self.pos_dist = np.ascontiguousarray(pos_dist, dtype=np.float64)
        self.p = OrcParams()
        self.p.n_ind, self.p.n_sites = self.n_ind, self.n_sites
        self.p.max_kb_dist, self.p.max_snp_dist, self.p.min_maf = max_kb_dist, max_snp_dist, min_maf
        self.p.ignore_miss_data, self.p.n_threads = int(ignore_miss_data), n_threads
        self.p.rnd_sample, self.p.seed = rnd_sample, seed
        self.p.geno_lkl, self.p.maf, self.p.expected_geno = dp(self.gl), dp(self.maf), dp(self.expg)
        self.p.pos_dist = dp(self.pos_dist)
        L.orc_preprocess(C.byref(self.p))  # gl -> normal space in place

    def count(self, s1_begin: int = 0, s1_end: int | None = None) -> int:
        e = C.c_int(0)
        return lib().orc_run(C.byref(self.p), s1_begin, self.n_sites if s1_end is None else s1_end, None, 0,
                             C.byref(e))

    def run(self, s1_begin: int = 0, s1_end: int | None = None) -> np.ndarray:
        s1_end = self.n_sites if s1_end is None else s1_end
        n = self.count(s1_begin, s1_end)
        out = np.zeros(n, dtype=PAIR_DTYPE)
        e = C.c_int(0)
        lib().orc_run(C.byref(self.p), s1_begin, s1_end, out.ctypes.data_as(C.c_void_p), n, C.byref(e))
        if e.value:
            raise RuntimeError(f"oracle error code {e.value}")
        return out

    def bench(self, s1_begin: int, s1_end: int) -> tuple[int, float, int]:
        """Compute rows [s1_begin, s1_end) and discard the records: (#pairs, checksum, executed EM iterations)."""
        chk, it = C.c_double(0.0), C.c_uint64(0)
        n = lib().orc_bench(C.byref(self.p), s1_begin, s1_end, C.byref(chk), C.byref(it))
        return n, chk.value, it.value

    def bench_pearson(self, s1_begin: int, s1_end: int) -> tuple[int, float]:
        """pearson_r alone over the pairs of rows [s1_begin, s1_end): (#pairs, sum of finite r2_ExpG)."""
        chk = C.c_double(0.0)
        n = lib().orc_bench_pearson(C.byref(self.p), s1_begin, s1_end, C.byref(chk))
        return n, chk.value

    def bench_reference(self, s1_begin: int, s1_end: int, n_threads: int) -> tuple[int, int, float] | None:
        """The REFERENCE's own compiled haplo_freq (oracle/_ref) over the pairs of rows [s1_begin, s1_end) of this
        matrix: (#pairs, executed EM iterations, sum of hap[0]).  None when oracle/_ref is not there."""
        R = ref()
        if R is None or not hasattr(R, "ref_bench_haplo_freq"):
            return None
        ends = np.ascontiguousarray(self.row_ends(), dtype=np.uint64)
        it, chk = C.c_uint64(0), C.c_double(0.0)
        n = R.ref_bench_haplo_freq(dp(self.gl), dp(self.maf), ends.ctypes.data_as(C.POINTER(C.c_uint64)), self.n_ind,
                                   self.n_sites, s1_begin, s1_end, int(self.p.ignore_miss_data), n_threads, C.byref(it),
                                   C.byref(chk))
        return int(n), int(it.value), float(chk.value)

    def row_ends(self) -> np.ndarray:
        if getattr(self, "_row_ends", None) is None:
            self._row_ends = np.array([lib().orc_row_end(C.byref(self.p), s) for s in range(self.n_sites)], dtype=np.uint64)
        return self._row_ends
