"""ctypes binding of the C-ABI in include/ngsld.h and include/ngsld_host.h (libngsld.so, built in-tree).

This is plumbing: every call goes straight into the HIP library.  There is no Python or CPU fallback --
if the library is missing or no gfx950 device is present the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("NGSLD_LIB") or os.path.join(PKG_DIR, "libngsld.so")  # NGSLD_LIB: A/B builds of the same C-ABI
CLI_PATH = os.path.join(PKG_DIR, "bin", "ngsLD")

OK, ERR_INVALID, ERR_DEVICE, ERR_NOMEM, ERR_NAN, ERR_MAF_RANGE, ERR_SINK, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7


class Params(C.Structure):
    _fields_ = [("max_kb_dist", C.c_uint64), ("max_snp_dist", C.c_uint64), ("min_maf", C.c_double),
                ("ignore_miss_data", C.c_int32), ("extend_out", C.c_int32), ("rnd_sample", C.c_double),
                ("seed", C.c_uint64), ("first_row", C.c_uint64)]


class GenoOpts(C.Structure):
    _fields_ = [("log_scale", C.c_int32), ("ignore_miss_data", C.c_int32), ("on_device", C.c_int32),
                ("text_semantics", C.c_int32), ("call_geno", C.c_int32), ("per_individual_only", C.c_int32),
                ("N_thresh", C.c_double), ("call_thresh", C.c_double)]


REC_STD = np.dtype([("r2_ExpG", "<f8"), ("D", "<f8"), ("Dp", "<f8"), ("r2", "<f8")])
REC_EXT = np.dtype([("hap", "<f8", (4,)), ("n_ind_data", "<u4"), ("n_iter", "<u4")])


ITEM = np.dtype([("s1", "<u4"), ("s2_begin", "<u4"), ("count", "<u4"), ("reserved", "<u4"), ("mask", "<u8"),
                 ("first_record", "<u8")])


class Batch(C.Structure):
    _fields_ = [("s1_begin", C.c_uint64), ("s1_end", C.c_uint64), ("n_pairs", C.c_uint64), ("n_items", C.c_uint64),
                ("items", C.c_void_p), ("std", C.c_void_p), ("ext", C.c_void_p), ("text", C.c_void_p),
                ("text_len", C.c_uint64)]


def items_to_pairs(items: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """(s1, s2) of every record described by an item array, in record order."""
    if len(items) == 0:
        return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64)
    bits = (items["mask"][:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)
    bits &= (np.arange(64)[None, :] < items["count"][:, None]).astype(np.uint64)
    idx, c = np.nonzero(bits)
    return items["s1"][idx].astype(np.uint64), (items["s2_begin"][idx].astype(np.uint64) + c.astype(np.uint64))


def items_from_pairs(s1: np.ndarray, s2: np.ndarray, span: int = 64) -> np.ndarray:
    """Build an item array for a list of pairs sorted by (s1, s2): chunks of `span` candidates from s1 + 1."""
    s1, s2 = np.asarray(s1, dtype=np.int64), np.asarray(s2, dtype=np.int64)
    chunk = (s2 - s1 - 1) // span
    key = s1 * (1 << 32) + chunk
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    items = np.zeros(len(uniq), dtype=ITEM)
    items["s1"] = s1[first]
    items["s2_begin"] = s1[first] + 1 + chunk[first] * span
    items["first_record"] = first
    np.bitwise_or.at(items["mask"], inv, np.uint64(1) << (s2 - items["s2_begin"][inv]).astype(np.uint64))
    last = np.zeros(len(uniq), dtype=np.int64)
    np.maximum.at(last, inv, s2 - items["s2_begin"][inv].astype(np.int64))
    items["count"] = last + 1
    return items


SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Batch))
MULTI_SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(Batch))
READ_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p)
SLAB = np.dtype([("row_begin", "<u8"), ("row_end", "<u8"), ("site_end", "<u8")])

# every symbol the two headers declare (checked by tests/test_abi.py against the headers' text)
SYMBOLS = [
    "ngsld_version", "ngsld_create", "ngsld_destroy", "ngsld_last_error", "ngsld_set_geno_raw",
    "ngsld_set_geno_raw_opts", "ngsld_set_geno_lkl",
    "ngsld_get_maf", "ngsld_set_pos_dist", "ngsld_plan", "ngsld_plan_rows", "ngsld_run", "ngsld_run_device", "ngsld_set_text_output",
    "ngsld_set_replay_source", "ngsld_set_replay_matrix", "ngsld_set_replay", "ngsld_replay_stats", "ngsld_finish_device",
    "ngsld_set_exact_store", "ngsld_replay_info",
    "ngsld_plan_parts", "ngsld_run_multi", "ngsld_multi_last_distribution", "ngsld_rccl_selftest",
    "ngsld_last_kernel_time", "ngsld_pair_kernel", "ngsld_describe_dispatch", "ngsld_set_tuning", "ngsld_selftest", "ngsld_reserve_text_buffers",
    "ngsld_window_ends", "ngsld_plan_slabs", "ngsld_slab_sites_for_budget", "ngsld_sites_for_budget", "ngsld_streamed_replay_info", "ngsld_set_memory_budget", "ngsld_device_memory", "ngsld_run_streamed", "ngsld_run_streamed_text",
    "ngsld_host_read_geno_bin_range",
    "ngsld_host_set_threads", "ngsld_host_read_pos", "ngsld_host_pos_dist", "ngsld_host_label", "ngsld_host_free_pos", "ngsld_host_pos_slice",
    "ngsld_host_geno_size_ok", "ngsld_host_read_geno_bin", "ngsld_host_read_geno_text", "ngsld_host_format_header", "ngsld_host_format_pair",
    "ngsld_host_format_double", "ngsld_host_write_batch", "ngsld_host_replay_pair", "ngsld_host_missing_call_log",
    "ngsld_host_gz_open", "ngsld_host_gz_close",
]


class ReplayStats(C.Structure):
    """ngsld_replay_stats_t (include/ngsld.h)."""
    _fields_ = [("pairs_flagged", C.c_uint64), ("pairs_replayed", C.c_uint64), ("pairs_on_device", C.c_uint64),
                ("pairs_on_host", C.c_uint64), ("sites_reevaluated", C.c_uint64), ("exact_store", C.c_int32),
                ("text_rows_patched", C.c_int32), ("exact_store_build_s", C.c_double), ("sites_degenerate", C.c_uint64)]


class NgsldError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ngsld error {code}: {msg}")
        self.code = code
        self.msg = msg


def build(force: bool = False, jobs: int = 8) -> None:
    """Compile the HIP library and the CLI in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(PKG_DIR, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", f"-j{jobs}", "-C", csrc, "all"])


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: run ngsld_amd.capi.build() (or make -C ngsld_amd/csrc); "
                                    "there is no fallback implementation")
        L = C.CDLL(LIB_PATH)
        vp, u64, dbl = C.c_void_p, C.c_uint64, C.c_double
        L.ngsld_version.restype = C.c_char_p
        L.ngsld_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.ngsld_destroy.argtypes = [vp]
        L.ngsld_destroy.restype = None
        L.ngsld_last_error.argtypes = [vp]
        L.ngsld_last_error.restype = C.c_char_p
        L.ngsld_set_geno_raw.argtypes = [vp, vp, u64, u64, C.c_int, C.c_int, C.c_int]
        L.ngsld_set_geno_raw_opts.argtypes = [vp, vp, u64, u64, C.POINTER(GenoOpts)]
        L.ngsld_set_geno_lkl.argtypes = [vp, vp, vp, u64, u64, C.c_int]
        L.ngsld_get_maf.argtypes = [vp, vp]
        L.ngsld_set_pos_dist.argtypes = [vp, vp]
        L.ngsld_plan.argtypes = [vp, C.POINTER(Params), C.POINTER(u64)]
        L.ngsld_plan_rows.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint32))]
        L.ngsld_run.argtypes = [vp, u64, u64, SINK_FN, vp]
        L.ngsld_run_device.argtypes = [vp, u64, u64, vp, vp, vp]
        if hasattr(L, "ngsld_set_replay_source"):  # (absent only from older A/B builds loaded through NGSLD_LIB)
            L.ngsld_set_replay_source.argtypes = [vp, READ_FN, vp]
            if hasattr(L, "ngsld_set_replay_matrix"):
                L.ngsld_set_replay_matrix.argtypes = [vp, vp]
            L.ngsld_set_replay.argtypes = [vp, C.c_int]
            L.ngsld_replay_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        if hasattr(L, "ngsld_replay_info"):
            L.ngsld_set_exact_store.argtypes = [vp, C.c_int]
            L.ngsld_replay_info.argtypes = [vp, C.POINTER(ReplayStats)]
            L.ngsld_finish_device.argtypes = [vp]
        if hasattr(L, "ngsld_set_text_output"):  # (absent only from older A/B builds loaded through NGSLD_LIB)
            L.ngsld_set_text_output.argtypes = [vp, C.POINTER(C.c_char_p), C.c_int]
        L.ngsld_last_kernel_time.argtypes = [vp, C.POINTER(dbl), C.POINTER(u64), C.POINTER(u64)]
        L.ngsld_set_tuning.argtypes = [vp, C.c_uint32, u64]
        if hasattr(L, "ngsld_pair_kernel"):
            L.ngsld_pair_kernel.argtypes = [vp]
            L.ngsld_pair_kernel.restype = C.c_char_p
        L.ngsld_selftest.argtypes = [vp]
        L.ngsld_window_ends.argtypes = [vp, u64, C.POINTER(Params), vp]
        L.ngsld_plan_slabs.argtypes = [vp, u64, C.POINTER(Params), u64, vp, u64, C.POINTER(u64)]
        L.ngsld_slab_sites_for_budget.argtypes = [u64, u64]
        L.ngsld_slab_sites_for_budget.restype = u64
        if hasattr(L, "ngsld_sites_for_budget"):  # (absent only from older A/B builds loaded through NGSLD_LIB)
            L.ngsld_sites_for_budget.argtypes = [u64, u64, C.c_int]
            L.ngsld_sites_for_budget.restype = u64
        L.ngsld_device_memory.argtypes = [C.c_int, C.POINTER(u64), C.POINTER(u64)]
        L.ngsld_run_streamed.argtypes = [C.c_int, u64, u64, vp, C.POINTER(Params), C.POINTER(GenoOpts), u64, READ_FN, vp,
                                         vp, SINK_FN, vp, C.POINTER(u64), C.POINTER(u64), C.c_char_p, C.c_size_t]
        if hasattr(L, "ngsld_run_multi"):
            L.ngsld_plan_parts.argtypes = [vp, u64, C.POINTER(Params), C.c_int, vp]
            L.ngsld_run_multi.argtypes = [C.POINTER(C.c_int), C.c_int, u64, u64, vp, C.POINTER(Params), C.POINTER(GenoOpts),
                                          vp, READ_FN, vp, vp, MULTI_SINK_FN, vp, C.POINTER(C.c_char_p), C.c_int,
                                          C.POINTER(u64), C.c_char_p, C.c_size_t]
        if hasattr(L, "ngsld_multi_last_distribution"):
            L.ngsld_multi_last_distribution.argtypes = []
        if hasattr(L, "ngsld_rccl_selftest"):
            L.ngsld_rccl_selftest.argtypes = [C.c_int, u64, C.c_char_p, C.c_size_t]
        L.ngsld_host_read_geno_bin_range.argtypes = [C.c_char_p, u64, u64, u64, vp, C.c_char_p, C.c_size_t]
        L.ngsld_host_set_threads.argtypes = [C.c_int]
        L.ngsld_host_set_threads.restype = None
        L.ngsld_host_read_pos.argtypes = [C.c_char_p, C.c_int, u64, C.POINTER(vp), C.c_char_p, C.c_size_t]
        L.ngsld_host_pos_dist.argtypes = [vp]
        L.ngsld_host_pos_dist.restype = C.POINTER(dbl)
        L.ngsld_host_label.argtypes = [vp, u64]
        L.ngsld_host_label.restype = C.c_char_p
        L.ngsld_host_free_pos.argtypes = [vp]
        L.ngsld_host_free_pos.restype = None
        L.ngsld_host_pos_slice.argtypes = [vp, u64, u64]
        L.ngsld_host_pos_slice.restype = vp
        L.ngsld_host_geno_size_ok.argtypes = [u64, u64, u64]
        L.ngsld_host_read_geno_bin.argtypes = [C.c_char_p, u64, u64, vp, C.c_char_p, C.c_size_t]
        L.ngsld_host_read_geno_text.argtypes = [C.c_char_p, C.c_int, C.c_int, u64, u64, vp, C.POINTER(C.c_int),
                                                C.c_char_p, C.c_size_t]
        L.ngsld_host_format_header.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
        L.ngsld_host_format_header.restype = C.c_size_t
        L.ngsld_host_format_pair.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, dbl, vp, vp, dbl, dbl]
        L.ngsld_host_format_pair.restype = C.c_size_t
        L.ngsld_host_format_double.argtypes = [C.c_char_p, C.c_size_t, dbl, C.c_int]
        L.ngsld_host_format_double.restype = C.c_size_t
        L.ngsld_host_write_batch.argtypes = [C.POINTER(Batch), vp, vp, vp, C.c_int, C.c_int]
        L.ngsld_host_replay_pair.argtypes = [vp, vp, u64, C.POINTER(GenoOpts), vp, vp, vp]
        if hasattr(L, "ngsld_host_gz_open"):
            L.ngsld_host_gz_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(C.c_int)]
            L.ngsld_host_gz_close.argtypes = [vp]
        _lib = L
    return _lib


# ------------------------------------------------------------------------------------------------
# host helpers (no device)
# ------------------------------------------------------------------------------------------------
def read_pos(path: str, header: bool, n_sites: int) -> tuple[np.ndarray, list[str]]:
    L = lib()
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = L.ngsld_host_read_pos(path.encode(), int(header), n_sites, C.byref(h), err, len(err))
    if rc != OK:
        raise NgsldError(rc, err.value.decode())
    try:
        pd = np.ctypeslib.as_array(L.ngsld_host_pos_dist(h), shape=(n_sites,)).copy()
        labels = [L.ngsld_host_label(h, s).decode() for s in range(n_sites)]
    finally:
        L.ngsld_host_free_pos(h)
    return pd, labels


def read_geno_bin(path: str, n_ind: int, n_sites: int) -> np.ndarray:
    L = lib()
    out = np.empty((n_sites, n_ind, 3), dtype=np.float64)
    err = C.create_string_buffer(512)
    rc = L.ngsld_host_read_geno_bin(path.encode(), n_ind, n_sites, out.ctypes.data, err, len(err))
    if rc != OK:
        raise NgsldError(rc, err.value.decode())
    return out


def read_geno_bin_range(path: str, n_ind: int, site_begin: int, n_sites: int) -> np.ndarray:
    L = lib()
    out = np.empty((n_sites, n_ind, 3), dtype=np.float64)
    err = C.create_string_buffer(512)
    rc = L.ngsld_host_read_geno_bin_range(path.encode(), n_ind, site_begin, n_sites, out.ctypes.data, err, len(err))
    if rc != OK:
        raise NgsldError(rc, err.value.decode())
    return out


def _params(max_kb_dist=0, max_snp_dist=0, min_maf=0.0, ignore_miss_data=False, extend_out=True, rnd_sample=1.0, seed=0,
            first_row=0) -> Params:
    return Params(max_kb_dist, max_snp_dist, min_maf, int(ignore_miss_data), int(extend_out), rnd_sample, seed, first_row)


def window_ends(pos_dist: np.ndarray | None, n_sites: int, **kw) -> np.ndarray:
    """row_end[s1] from the distance / SNP-count limits alone (host only)."""
    pd = None if pos_dist is None else np.ascontiguousarray(pos_dist, dtype=np.float64)
    out = np.empty(n_sites, dtype=np.uint32)
    p = _params(**kw)
    rc = lib().ngsld_window_ends(None if pd is None else pd.ctypes.data, n_sites, C.byref(p), out.ctypes.data)
    if rc != OK:
        raise NgsldError(rc, "ngsld_window_ends")
    return out


def plan_slabs(pos_dist: np.ndarray | None, n_sites: int, max_slab_sites: int, **kw) -> np.ndarray:
    """Slabs (row_begin, row_end, site_end) of a streamed run (host only)."""
    pd = None if pos_dist is None else np.ascontiguousarray(pos_dist, dtype=np.float64)
    out = np.zeros(n_sites, dtype=SLAB)
    n = C.c_uint64()
    p = _params(**kw)
    rc = lib().ngsld_plan_slabs(None if pd is None else pd.ctypes.data, n_sites, C.byref(p), max_slab_sites,
                                out.ctypes.data, n_sites, C.byref(n))
    if rc != OK:
        raise NgsldError(rc, "the window of a single site does not fit the slab" if rc == ERR_NOMEM else "ngsld_plan_slabs")
    return out[:n.value].copy()


def slab_sites_for_budget(n_ind: int, budget_bytes: int) -> int:
    return int(lib().ngsld_slab_sites_for_budget(n_ind, budget_bytes))


def sites_for_budget(n_ind: int, budget_bytes: int, matrix_copies: int) -> int:
    """ngsld_sites_for_budget: the matrix priced matrix_copies times per context (1: the planes alone; 3: with the exact store)."""
    return int(lib().ngsld_sites_for_budget(n_ind, budget_bytes, matrix_copies))


def streamed_replay_info() -> dict:
    """ngsld_streamed_replay_info: where the flagged pairs of the last streamed job were replayed, summed over its slabs."""
    st = ReplayStats()
    L = lib()
    L.ngsld_streamed_replay_info.argtypes = [C.POINTER(ReplayStats)]
    rc = L.ngsld_streamed_replay_info(C.byref(st))
    if rc != OK:
        raise NgsldError(rc, "ngsld_streamed_replay_info")
    return {k: getattr(st, k) for k, _ in ReplayStats._fields_}


def set_memory_budget(device: int, n_bytes: int) -> None:
    """ngsld_set_memory_budget: a cap on what this process takes of the device's memory from now on (0: none)."""
    L = lib()
    L.ngsld_set_memory_budget.argtypes = [C.c_int, C.c_uint64]
    rc = L.ngsld_set_memory_budget(device, n_bytes)
    if rc != OK:
        raise NgsldError(rc, "ngsld_set_memory_budget")


def device_memory(device: int = 0) -> tuple[int, int]:
    f, t = C.c_uint64(), C.c_uint64()
    rc = lib().ngsld_device_memory(device, C.byref(f), C.byref(t))
    if rc != OK:
        raise NgsldError(rc, "ngsld_device_memory")
    return f.value, t.value


def run_streamed(read_sites, n_sites: int, n_ind: int, pos_dist: np.ndarray | None, max_slab_sites: int, device: int = 0,
                 log_scale: bool = False, call_geno: tuple[float, float] | None = None, text: bool = False, **kw):
    """The whole job slab by slab (ngsld_run_streamed).  read_sites(site_begin, n) -> array [n, n_ind, 3] of raw values.
    Returns (s1, s2, std, ext, maf, n_slabs) with global site indices."""
    L = lib()
    pd = None if pos_dist is None else np.ascontiguousarray(pos_dist, dtype=np.float64)
    p = _params(**kw)
    o = GenoOpts(int(log_scale), p.ignore_miss_data, 0, int(text), int(call_geno is not None), 0,
                 call_geno[0] if call_geno else 0.0, call_geno[1] if call_geno else 0.0)
    s1s, s2s, stds, exts = [], [], [], []

    def reader(_user, site_begin, n, dst):
        try:
            a = np.ascontiguousarray(read_sites(int(site_begin), int(n)), dtype=np.float64)
            assert a.size == n * n_ind * 3
            C.memmove(dst, a.ctypes.data, a.nbytes)
            return 0
        except Exception:  # noqa: BLE001 -- reported through the return code
            return 1

    def sink(_user, bp):
        b = bp.contents
        n = b.n_pairs
        if n:
            stds.append(np.frombuffer(C.string_at(b.std, n * REC_STD.itemsize), dtype=REC_STD).copy())
            if b.ext:
                exts.append(np.frombuffer(C.string_at(b.ext, n * REC_EXT.itemsize), dtype=REC_EXT).copy())
        items = np.frombuffer(C.string_at(b.items, b.n_items * ITEM.itemsize), dtype=ITEM) if b.n_items else \
            np.zeros(0, dtype=ITEM)
        a, bb = items_to_pairs(items)
        assert len(a) == n and (n == 0 or (a.min() >= b.s1_begin and a.max() < b.s1_end))
        s1s.append(a)
        s2s.append(bb)
        return 0

    maf = np.full(n_sites, np.nan)
    n_pairs, n_slabs = C.c_uint64(), C.c_uint64()
    err = C.create_string_buffer(512)
    rcb, scb = READ_FN(reader), SINK_FN(sink)
    rc = L.ngsld_run_streamed(device, n_sites, n_ind, None if pd is None else pd.ctypes.data, C.byref(p), C.byref(o),
                              max_slab_sites, rcb, None, maf.ctypes.data, scb, None, C.byref(n_pairs), C.byref(n_slabs),
                              err, len(err))
    if rc != OK:
        raise NgsldError(rc, err.value.decode())
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dtype=dt)
    s1 = cat(s1s, np.uint64)
    assert len(s1) == n_pairs.value
    return (s1, cat(s2s, np.uint64), cat(stds, REC_STD), cat(exts, REC_EXT) if p.extend_out else None, maf,
            int(n_slabs.value))


def plan_parts(pos_dist: np.ndarray | None, n_sites: int, n_parts: int, **kw) -> np.ndarray:
    """Row parts (row_begin, row_end, site_end) of a multi-device run (host only)."""
    pd = None if pos_dist is None else np.ascontiguousarray(pos_dist, dtype=np.float64)
    out = np.zeros(n_parts, dtype=SLAB)
    p = _params(**kw)
    rc = lib().ngsld_plan_parts(None if pd is None else pd.ctypes.data, n_sites, C.byref(p), n_parts, out.ctypes.data)
    if rc != OK:
        raise NgsldError(rc, "ngsld_plan_parts")
    return out


def run_multi(raw: np.ndarray, pos_dist: np.ndarray | None, devices: list[int], log_scale: bool = False,
              call_geno: tuple[float, float] | None = None, text: bool = False, labels: list[str] | None = None,
              text_output: bool = False, **kw):
    """One job over several devices in this process (ngsld_run_multi).  Returns per part a tuple
    (s1, s2, std, ext) -- or the part's TSV bytes with text_output -- plus the maf vector and the pairs per part."""
    L = lib()
    raw = np.ascontiguousarray(raw, dtype=np.float64)
    n_sites, n_ind = raw.shape[0], raw.shape[1]
    pd = None if pos_dist is None else np.ascontiguousarray(pos_dist, dtype=np.float64)
    p = _params(**kw)
    o = GenoOpts(int(log_scale), p.ignore_miss_data, 0, int(text), int(call_geno is not None), 0,
                 call_geno[0] if call_geno else 0.0, call_geno[1] if call_geno else 0.0)
    n = len(devices)
    acc = [dict(s1=[], s2=[], std=[], ext=[], text=[]) for _ in range(n)]

    def sink(_user, part, bp):
        b = bp.contents
        a = acc[part]
        if b.text:
            a["text"].append(C.string_at(b.text, b.text_len))
            return 0
        k = b.n_pairs
        if k:
            a["std"].append(np.frombuffer(C.string_at(b.std, k * REC_STD.itemsize), dtype=REC_STD).copy())
            if b.ext:
                a["ext"].append(np.frombuffer(C.string_at(b.ext, k * REC_EXT.itemsize), dtype=REC_EXT).copy())
        items = np.frombuffer(C.string_at(b.items, b.n_items * ITEM.itemsize), dtype=ITEM) if b.n_items else \
            np.zeros(0, dtype=ITEM)
        x, y = items_to_pairs(items)
        a["s1"].append(x)
        a["s2"].append(y)
        return 0

    maf = np.full(n_sites, np.nan)
    per = (C.c_uint64 * n)()
    err = C.create_string_buffer(512)
    devs = (C.c_int * n)(*devices)
    lab = None
    if labels is not None:
        lab = (C.c_char_p * len(labels))(*[l.encode() for l in labels])
    cb = MULTI_SINK_FN(sink)
    rc = L.ngsld_run_multi(devs, n, n_sites, n_ind, None if pd is None else pd.ctypes.data, C.byref(p), C.byref(o),
                           raw.ctypes.data, READ_FN(0), None, maf.ctypes.data, cb, None, lab, int(text_output), per, err,
                           len(err))
    if rc != OK:
        raise NgsldError(rc, err.value.decode())
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dtype=dt)
    out = []
    for a in acc:
        if text_output:
            out.append(b"".join(a["text"]))
        else:
            out.append((cat(a["s1"], np.uint64), cat(a["s2"], np.uint64), cat(a["std"], REC_STD),
                        cat(a["ext"], REC_EXT) if p.extend_out else None))
    return out, maf, [int(v) for v in per]


def run_multi_sink(raw: np.ndarray, pos_dist: np.ndarray | None, devices: list[int], sink=None, log_scale: bool = False, **kw):
    """ngsld_run_multi with the records left where they arrive: sink(part, n_pairs, std, ext) sees every batch as numpy
    views of the library's host buffers (valid during the call; called on the part's own thread), or nothing is looked at
    when sink is None.  Returns the pairs per part."""
    L = lib()
    raw = np.ascontiguousarray(raw, dtype=np.float64)
    n_sites, n_ind = raw.shape[0], raw.shape[1]
    pd = None if pos_dist is None else np.ascontiguousarray(pos_dist, dtype=np.float64)
    p = _params(**kw)
    o = GenoOpts(int(log_scale), p.ignore_miss_data, 0, 0, 0, 0, 0.0, 0.0)
    n = len(devices)

    def cb(_user, part, bp):
        if sink is None:
            return 0
        b = bp.contents
        k = b.n_pairs
        if k == 0:
            return 0
        std = np.ctypeslib.as_array(C.cast(b.std, C.POINTER(C.c_uint8)), shape=(k * REC_STD.itemsize,)).view(REC_STD)
        ext = np.ctypeslib.as_array(C.cast(b.ext, C.POINTER(C.c_uint8)), shape=(k * REC_EXT.itemsize,)).view(REC_EXT) \
            if b.ext else None
        return int(sink(part, k, std, ext) or 0)

    per = (C.c_uint64 * n)()
    err = C.create_string_buffer(512)
    devs = (C.c_int * n)(*devices)
    rc = L.ngsld_run_multi(devs, n, n_sites, n_ind, None if pd is None else pd.ctypes.data, C.byref(p), C.byref(o),
                           raw.ctypes.data, READ_FN(0), None, None, MULTI_SINK_FN(cb), None, None, 0, per, err, len(err))
    if rc != OK:
        raise NgsldError(rc, err.value.decode())
    return [int(v) for v in per]


DIST_NAMES = {0: "none", 1: "upload", 2: "peer_copy", 3: "rccl"}


def multi_last_distribution() -> str:
    """How the last run_multi of this process handed the matrix to its devices: "upload" (slab by slab), "peer_copy"
    (device to device from the first) or "rccl" (one ncclBroadcast over xGMI)."""
    return DIST_NAMES[int(lib().ngsld_multi_last_distribution())]


def rccl_selftest(device: int = 0, n_bytes: int = 64 << 20) -> None:
    """librccl's part in ngsld_run_multi on a one-device communicator (ngsld_rccl_selftest); raises NgsldError."""
    err = C.create_string_buffer(512)
    rc = lib().ngsld_rccl_selftest(device, n_bytes, err, len(err))
    if rc != OK:
        raise NgsldError(rc, err.value.decode())


def describe_dispatch(n_ind: int, ignore_miss_data: bool = False) -> str:
    """Kernel family and shape a likelihood matrix of n_ind individuals runs on (no device needed): ngsld_describe_dispatch."""
    buf = C.create_string_buffer(96)
    L = lib()
    L.ngsld_describe_dispatch.argtypes = [C.c_uint64, C.c_int, C.c_char_p, C.c_size_t]
    rc = L.ngsld_describe_dispatch(int(n_ind), int(bool(ignore_miss_data)), buf, 96)
    if rc != OK:
        raise NgsldError(rc, f"no kernel for n_ind = {n_ind}")
    return buf.value.decode()


def dispatch_table(n_max: int = 12_000) -> str:
    """describe_dispatch over 1..n_max for both settings of ignore_miss_data, as ranges of equal text."""
    import re
    lines = []

    def describe(n, masked):  # (the streaming kernel pads to the next 64 whatever the size: one line for all of them)
        d = describe_dispatch(n, masked)
        return re.sub(r"np=\d+", "np=n_ind rounded up to 64", d) if d.startswith("stream") else d

    for masked in (False, True):
        lines.append(f"# ignore_miss_data = {int(masked)}")
        lo, cur = 1, describe(1, masked)
        for n in range(2, n_max + 2):
            d = describe(n, masked) if n <= n_max else None
            if d != cur:
                lines.append(f"{lo}..{n - 1}\t{cur}")
                lo, cur = n, d
    return "\n".join(lines) + "\n"


def device_count() -> int:
    """GPUs visible to this process (hipGetDeviceCount through the library's own runtime; 0 without a GPU)."""
    n = 0
    while n < 64:
        free, total = C.c_uint64(0), C.c_uint64(0)
        if lib().ngsld_device_memory(n, C.byref(free), C.byref(total)) != OK:
            break
        n += 1
    return n


def read_geno_text(path: str, in_probs: bool, log_scale: bool, n_ind: int, n_sites: int) -> tuple[np.ndarray, bool]:
    """Text/.gz genotype file -> (raw [n_sites, n_ind, 3], values_are_logs) for Engine.set_geno_raw(text=True)."""
    L = lib()
    out = np.empty((n_sites, n_ind, 3), dtype=np.float64)
    err = C.create_string_buffer(512)
    is_log = C.c_int(0)
    rc = L.ngsld_host_read_geno_text(path.encode(), int(in_probs), int(log_scale), n_ind, n_sites, out.ctypes.data,
                                     C.byref(is_log), err, len(err))
    if rc != OK:
        raise NgsldError(rc, err.value.decode())
    return out, bool(is_log.value)


def replay_pair(raw1: np.ndarray, raw2: np.ndarray, log_scale: bool = False, ignore_miss_data: bool = False,
                text: bool = False, call_geno: tuple | None = None):
    """One pair in the reference's own operation order on the host (ngsld_host_replay_pair).
    raw1, raw2: [n_ind, 3] raw values of the two sites.  Returns (std record, ext record, (maf1, maf2))."""
    a = np.ascontiguousarray(raw1, dtype=np.float64)
    b = np.ascontiguousarray(raw2, dtype=np.float64)
    o = GenoOpts(int(log_scale), int(ignore_miss_data), 0, int(text), int(call_geno is not None), 0,
                 float(call_geno[0]) if call_geno else 0.0, float(call_geno[1]) if call_geno else 0.0)
    s, e, m = np.zeros(1, dtype=REC_STD), np.zeros(1, dtype=REC_EXT), np.zeros(2)
    rc = lib().ngsld_host_replay_pair(a.ctypes.data, b.ctypes.data, a.shape[0], C.byref(o), s.ctypes.data, e.ctypes.data,
                                      m.ctypes.data)
    if rc not in (OK, ERR_MAF_RANGE):
        raise NgsldError(rc, "ngsld_host_replay_pair")
    return s[0], e[0], (float(m[0]), float(m[1]))


def format_double(v: float, decimals: int = 6) -> str:
    buf = C.create_string_buffer(512)
    n = lib().ngsld_host_format_double(buf, len(buf), v, decimals)
    return buf.raw[:n].decode()


def format_header(extend_out: bool) -> str:
    buf = C.create_string_buffer(512)
    n = lib().ngsld_host_format_header(buf, len(buf), int(extend_out))
    return buf.raw[:n].decode()


def missing_call_log() -> float:
    """log(1/3) as the text reader stores it for a missing call (ngsld_host_missing_call_log)."""
    f = lib().ngsld_host_missing_call_log
    f.restype = C.c_double
    f.argtypes = []
    return float(f())


def format_pair(l1, l2, dist: float, std_rec: np.ndarray, ext_rec: np.ndarray | None, maf1: float, maf2: float) -> str:
    buf = C.create_string_buffer(8192)
    s = np.ascontiguousarray(std_rec).reshape(1).view(REC_STD) if std_rec.dtype != REC_STD else std_rec.reshape(1)
    e_ptr = None
    if ext_rec is not None:
        e = ext_rec.reshape(1)
        e_ptr = e.ctypes.data
    n = lib().ngsld_host_format_pair(buf, len(buf), None if l1 is None else l1.encode(),
                                     None if l2 is None else l2.encode(), dist, s.ctypes.data, e_ptr, maf1, maf2)
    return buf.raw[:n].decode()


# ------------------------------------------------------------------------------------------------
# the engine
# ------------------------------------------------------------------------------------------------
class Engine:
    """One ngsld_ctx.  Mirrors the call order of the reference's main(): data -> positions -> run."""

    def __init__(self, device: int = 0):
        self._L = lib()
        self._h = C.c_void_p()
        rc = self._L.ngsld_create(device, C.byref(self._h))
        if rc != OK:
            raise NgsldError(rc, self._L.ngsld_last_error(None).decode())
        self.n_sites = self.n_ind = 0
        self.extend_out = False
        self._source = None

    def close(self) -> None:
        if self._h:
            self._L.ngsld_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc != OK:
            raise NgsldError(rc, self._L.ngsld_last_error(self._h).decode())

    def selftest(self) -> None:
        self._check(self._L.ngsld_selftest(self._h))

    def set_geno_raw(self, gl, n_sites: int | None = None, n_ind: int | None = None, log_scale: bool = False,
                     ignore_miss_data: bool = False, text: bool = False, call_geno: tuple | None = None,
                     per_individual_only: bool = False, replay_source: bool = True) -> None:
        """gl: numpy float64 [n_sites, n_ind, 3] (host) or an int device pointer with explicit sizes.
        text: values come from a text genotype file (read_geno_text); call_geno = (N_thresh, call_thresh)."""
        if isinstance(gl, np.ndarray):
            gl = np.ascontiguousarray(gl, dtype=np.float64)
            n_sites, n_ind = gl.shape[0], gl.shape[1]
            ptr, on_dev = gl.ctypes.data, 0
        else:
            ptr, on_dev = int(gl), 1
        o = GenoOpts(int(log_scale), int(ignore_miss_data), on_dev, int(text), int(call_geno is not None), int(per_individual_only),
                     float(call_geno[0]) if call_geno else 0.0, float(call_geno[1]) if call_geno else 0.0)
        self._check(self._L.ngsld_set_geno_raw_opts(self._h, ptr, n_sites, n_ind, C.byref(o)))
        self.n_sites, self.n_ind = n_sites, n_ind
        self._source = None
        if isinstance(gl, np.ndarray) and replay_source:
            self.set_replay_source(gl)

    def set_geno_lkl(self, geno_lkl: np.ndarray, maf: np.ndarray, replay_source: bool = True) -> None:
        g = np.ascontiguousarray(geno_lkl, dtype=np.float64)
        m = np.ascontiguousarray(maf, dtype=np.float64)
        self._check(self._L.ngsld_set_geno_lkl(self._h, g.ctypes.data, m.ctypes.data, g.shape[0], g.shape[1], 0))
        self.n_sites, self.n_ind = g.shape[0], g.shape[1]
        self._source = None
        if replay_source:
            self.set_replay_source(g)

    def set_replay_source(self, values: np.ndarray | None) -> None:
        """The matrix the exact-order replay reads the flagged pairs' sites from again: the array given to set_geno_raw /
        set_geno_lkl (kept alive by this object).  None: the library reads its own planes back from the device."""
        if values is None or not hasattr(self._L, "ngsld_set_replay_source"):
            self._source = None
            if hasattr(self._L, "ngsld_set_replay_source"):
                self._check(self._L.ngsld_set_replay_source(self._h, READ_FN(0), None))
            return
        arr = np.ascontiguousarray(values, dtype=np.float64).reshape(self.n_sites, -1)
        row = arr.shape[1] * 8
        if hasattr(self._L, "ngsld_set_replay_matrix") and os.environ.get("NGSLD_PY_REPLAY_CALLBACK") != "1":
            self._source = (arr, None)   # read in place by the library's replay threads: no callback, no GIL
            self._check(self._L.ngsld_set_replay_matrix(self._h, arr.ctypes.data))
            return

        def reader(_user, site_begin, n, dst):
            if site_begin + n > arr.shape[0]:
                return 1
            C.memmove(dst, arr.ctypes.data + int(site_begin) * row, int(n) * row)
            return 0

        cb = READ_FN(reader)
        self._source = (arr, cb)
        self._check(self._L.ngsld_set_replay_source(self._h, cb, None))

    def set_replay(self, enable: bool) -> None:
        self._check(self._L.ngsld_set_replay(self._h, int(enable)))

    def replay_stats(self) -> tuple[int, int]:
        """(pairs replayed by the last run, sites re-evaluated by the last plan + run)."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._L.ngsld_replay_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_exact_store(self, mode: int) -> None:
        """0: flagged pairs of a likelihood matrix are replayed on host threads only; 1 (default): on the device once a run
        flags more of them than the host should replay; 2: from the first flagged pair on (ngsld_set_exact_store)."""
        self._check(self._L.ngsld_set_exact_store(self._h, int(mode)))

    def replay_info(self) -> dict:
        """ngsld_replay_info: where the flagged pairs of the last run were replayed."""
        st = ReplayStats()
        self._check(self._L.ngsld_replay_info(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in ReplayStats._fields_}

    def finish_device(self) -> None:
        self._check(self._L.ngsld_finish_device(self._h))

    def maf(self) -> np.ndarray:
        out = np.empty(self.n_sites, dtype=np.float64)
        self._check(self._L.ngsld_get_maf(self._h, out.ctypes.data))
        return out

    def set_pos_dist(self, pos_dist: np.ndarray | None) -> None:
        if pos_dist is None:
            self._check(self._L.ngsld_set_pos_dist(self._h, None))
        else:
            pd = np.ascontiguousarray(pos_dist, dtype=np.float64)
            assert pd.shape[0] == self.n_sites
            self._check(self._L.ngsld_set_pos_dist(self._h, pd.ctypes.data))

    def set_tuning(self, pairs_per_item: int = 0, batch_pairs: int = 0) -> None:
        self._check(self._L.ngsld_set_tuning(self._h, pairs_per_item, batch_pairs))

    def plan(self, max_kb_dist: int = 0, max_snp_dist: int = 0, min_maf: float = 0.0, ignore_miss_data: bool = False,
             extend_out: bool = True, rnd_sample: float = 1.0, seed: int = 0, first_row: int = 0) -> int:
        p = Params(max_kb_dist, max_snp_dist, min_maf, int(ignore_miss_data), int(extend_out), rnd_sample, seed,
                   first_row)
        n = C.c_uint64()
        self._check(self._L.ngsld_plan(self._h, C.byref(p), C.byref(n)))
        self.extend_out = extend_out
        return n.value

    def plan_rows(self) -> tuple[np.ndarray, np.ndarray]:
        ro, re = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)()
        self._check(self._L.ngsld_plan_rows(self._h, C.byref(ro), C.byref(re)))
        return (np.ctypeslib.as_array(ro, shape=(self.n_sites + 1,)).copy(),
                np.ctypeslib.as_array(re, shape=(self.n_sites,)).copy())

    def run(self, s1_begin: int = 0, s1_end: int | None = None):
        """Run rows [s1_begin, s1_end) through the sink path; returns (s1, s2, std, ext) arrays."""
        s1_end = self.n_sites if s1_end is None else s1_end
        s1s, s2s, stds, exts = [], [], [], []

        def sink(_user, bp):
            b = bp.contents
            n = b.n_pairs
            if n:
                stds.append(np.frombuffer(C.string_at(b.std, n * REC_STD.itemsize), dtype=REC_STD).copy())
                if b.ext:
                    exts.append(np.frombuffer(C.string_at(b.ext, n * REC_EXT.itemsize), dtype=REC_EXT).copy())
            items = np.frombuffer(C.string_at(b.items, b.n_items * ITEM.itemsize), dtype=ITEM) if b.n_items else \
                np.zeros(0, dtype=ITEM)
            a, bb = items_to_pairs(items)
            assert len(a) == n
            if len(items):
                assert np.array_equal(items["first_record"], np.concatenate([[0], np.cumsum(
                    [bin(int(m)).count("1") for m in items["mask"]])[:-1]]).astype(np.uint64))
            s1s.append(a)
            s2s.append(bb)
            return 0

        cb = SINK_FN(sink)
        self._check(self._L.ngsld_run(self._h, s1_begin, s1_end, cb, None))
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dtype=dt)
        return (cat(s1s, np.uint64), cat(s2s, np.uint64), cat(stds, REC_STD),
                cat(exts, REC_EXT) if self.extend_out else None)

    def run_to_fd(self, s1_begin: int, s1_end: int, fd: int, pos_handle, pos_dist: np.ndarray | None, maf: np.ndarray,
                  n_threads: int) -> int:
        """Rows [s1_begin, s1_end) -> TSV rows on file descriptor fd: text batches (after set_text_output) are written
        as they come, record batches go through ngsld_host_write_batch."""
        maf = np.ascontiguousarray(maf, dtype=np.float64)
        pd_ptr = None if pos_dist is None else np.ascontiguousarray(pos_dist, dtype=np.float64).ctypes.data
        total, rc_box = [0], [0]

        libc = C.CDLL(None, use_errno=True)
        libc.write.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        libc.write.restype = C.c_ssize_t

        def sink(_user, bp):
            b = bp.contents
            total[0] += b.n_pairs
            if b.text:                                   # rows formatted on the device (set_text_output): only bytes to write
                off = 0
                while off < b.text_len:
                    w = libc.write(fd, b.text + off, min(b.text_len - off, 1 << 30))
                    if w <= 0:
                        return 1
                    off += w
                return 0
            rc_box[0] = self._L.ngsld_host_write_batch(bp, pos_handle, pd_ptr, maf.ctypes.data, n_threads, fd)
            return 0 if rc_box[0] == OK else 1

        self._check(self._L.ngsld_run(self._h, s1_begin, s1_end, SINK_FN(sink), None))
        return total[0]

    def set_text_output(self, labels: list[str] | None, enable: bool = True) -> None:
        """Device-side TSV: ngsld_run then hands over text instead of records (labels None = "(null)")."""
        arr = None
        if labels is not None:
            arr = (C.c_char_p * len(labels))(*[l.encode() for l in labels])
        self._check(self._L.ngsld_set_text_output(self._h, arr, int(enable)))

    def run_text(self, s1_begin: int = 0, s1_end: int | None = None) -> tuple[bytes, int]:
        """Run through the sink path with device-side TSV on; returns (text of all batches, batches that arrived
        as records instead)."""
        s1_end = self.n_sites if s1_end is None else s1_end
        parts, fallbacks = [], [0]

        def sink(_user, bp):
            b = bp.contents
            if b.text:
                parts.append(C.string_at(b.text, b.text_len))
            else:
                fallbacks[0] += 1
            return 0

        self._check(self._L.ngsld_run(self._h, s1_begin, s1_end, SINK_FN(sink), None))
        return b"".join(parts), fallbacks[0]

    def run_discard(self, s1_begin: int = 0, s1_end: int | None = None) -> int:
        """Run through the sink path (kernel + D2H of every record into pinned host memory) and only count the
        pairs delivered: the host-buffer hand-off rate without any formatting."""
        s1_end = self.n_sites if s1_end is None else s1_end
        total = [0]

        def sink(_user, bp):
            total[0] += bp.contents.n_pairs
            return 0

        self._check(self._L.ngsld_run(self._h, s1_begin, s1_end, SINK_FN(sink), None))
        return total[0]

    def run_device(self, s1_begin: int, s1_end: int, d_std: int, d_ext: int | None, stream: int | None = None) -> None:
        self._check(self._L.ngsld_run_device(self._h, s1_begin, s1_end, d_std, d_ext, stream))

    def pair_kernel(self) -> str:
        """Kernel family of the data set last: group / run / wave / multi / stream / direct / hard."""
        return self._L.ngsld_pair_kernel(self._h).decode()

    def last_kernel_time(self) -> tuple[float, int, int]:
        ms, nl, npairs = C.c_double(), C.c_uint64(), C.c_uint64()
        self._check(self._L.ngsld_last_kernel_time(self._h, C.byref(ms), C.byref(nl), C.byref(npairs)))
        return ms.value, nl.value, npairs.value
