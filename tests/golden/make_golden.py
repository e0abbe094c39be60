#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ (run in the build container, where /root/reference exists).

Every fixture is DATA: a synthetic input (raw GL matrix, position file text, flags) and the expected
outputs.  Which code produced which expected field:

  * `ref_*` fields  -- the REFERENCE's own functions, compiled from /root/reference by oracle/build_ref.sh
                       (read_geno binary branch, est_maf, conv_space, haplo_freq, read_dist, labels):
                       reader hash, maf, hap[4], n_iter, n_ind_data, pos_dist, labels -- and, since round 4, the
                       GSL-free LINES of ngsLD.cpp compiled from where they lie (build_ref.sh cuts them out by
                       anchor): the s2 walk with its running dist (ngsLD.cpp:240-275; fixtures without
                       --rnd_sample), D, D', r2, hap_maf (:296-306), the float chi2 (:328-333) and every TSV row
                       through the reference's own fprintf lines (:314-351; r2_ExpG is the one value that
                       enters them from the oracle).
  * `orc_*` fields  -- the CPU oracle (oracle/ngsld_oracle.c).  Only r2_ExpG (gsl_stats_correlation: GSL is not
                       in the image) and the --rnd_sample draws (gsl_rng_taus) have no reference-compiled twin;
                       every other `orc_*` field is kept next to its `ref_*` twin and the script asserts
                       oracle == reference bit for bit before it writes anything.

No reference source text is stored; the fixtures hold inputs and numbers only.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from ngsld_amd import shard, synth  # noqa: E402
from oracle import orc  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pos_text(chrs, pos, extra_col=False, header=False) -> str:
    lines = []
    if header:
        lines.append("chr\tpos\tid" if extra_col else "chr\tpos")
    for k, (c, p) in enumerate(zip(chrs, pos)):
        lines.append(f"{c}\t{int(p)}\tsnp{k}" if extra_col else f"{c}\t{int(p)}")
    return "\n".join(lines) + "\n"


def geno_text(mat, header: bool, prefix_cols: bool) -> str:
    """A beagle-like text genotype file: optional header, optional non-numeric leading columns."""
    lines = []
    if header:
        lines.append("marker\tallele1\tallele2\t" + "\t".join(f"Ind{i}" for i in range(mat.shape[1])))
    for s in range(mat.shape[0]):
        pre = f"chr1_{s}\tA\tC\t" if prefix_cols else ""
        lines.append(pre + "\t".join(str(int(x)) if np.isfinite(x) and float(x) == int(x) and abs(x) <= 9
                                     else repr(float(x)) for x in mat[s].reshape(-1)))
    return "\n".join(lines) + "\n"


def run_cli(raw, ptxt, flags, header=False, gtext=None):
    """Oracle CLI -> TSV text (sorted the way the reference's test does, examples/test.sh:16)."""
    import gzip
    with tempfile.TemporaryDirectory() as d:
        if gtext is None:
            g = os.path.join(d, "in.glf")
            raw.tofile(g)
        else:
            g = os.path.join(d, "in.geno.gz")
            with gzip.open(g, "wt") as fh:
                fh.write(gtext)
        cmd = [orc.ORC_CLI, "--geno", g, "--n_ind", str(raw.shape[1]), "--n_sites", str(raw.shape[0]), "--verbose", "0"]
        if ptxt is not None:
            p = os.path.join(d, "in.pos")
            open(p, "w").write(ptxt)
            cmd += ["--posH" if header else "--pos", p]
        cmd += flags
        out = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    return out


ONLY: list[str] = []   # fixture-name prefixes given on the command line: regenerate just those


def make(name, raw, *a, **kw):
    if ONLY and not any(name.startswith(p) for p in ONLY):
        return
    _make(name, raw, *a, **kw)


def _make(name, raw, chrs=None, pos=None, log_scale=False, ignore_miss=False, max_kb=0, max_snp=0, min_maf=0.0,
         extra_col=False, header=False, with_text=True, text_mode=None, call=None, geno_header=True,
         rnd_sample=1.0, seed=0):
    """text_mode: None (binary GL file) | "probs" (text GL triples) | "called" (text genotypes, raw = [sites, ind]
    of {-1,0,1,2}); call = (N_thresh, call_thresh) adds --call_geno."""
    import gzip
    R = orc.ref()
    assert R is not None, "oracle/_ref/libngsld_ref.so missing: run oracle/build_ref.sh (needs /root/reference)"
    raw = np.ascontiguousarray(raw, dtype=np.float64)
    n_sites, n_ind = raw.shape[:2]
    gtext = None
    if text_mode is not None:
        gtext = geno_text(raw.reshape(n_sites, -1) if text_mode == "probs" else raw, geno_header, text_mode == "probs")
    ptxt = pos_text(chrs, pos, extra_col, header) if chrs is not None else None
    pd = shard.pos_dist_from_positions(chrs, pos) if chrs is not None else None

    # ---- reference: reader, maf, preprocessing ----
    with tempfile.TemporaryDirectory() as d:
        gl_log = np.empty((n_sites, n_ind, 3))
        gl_orc = np.empty((n_sites, n_ind, 3))
        if text_mode is None:
            g = os.path.join(d, "in.glf")
            raw.tofile(g)
            R.ref_read_geno_bin(g.encode(), int(log_scale), n_ind, n_sites, orc.dp(gl_log))
        else:
            g = os.path.join(d, "in.geno.gz")
            with gzip.open(g, "wt") as fh:
                fh.write(gtext)
            R.ref_read_geno_text(g.encode(), int(text_mode == "probs"), int(log_scale), n_ind, n_sites, orc.dp(gl_log))
            err = C.create_string_buffer(256)
            assert orc.lib().orc_read_geno_text(g.encode(), int(text_mode == "probs"), int(log_scale), n_ind, n_sites,
                                                orc.dp(gl_orc), err, 256) == 0, err.value
            assert np.array_equal(gl_orc, gl_log, equal_nan=True), "text reader: oracle != reference"
        if call is not None:
            for k in range(n_sites):
                for i in range(n_ind):
                    R.ref_call_geno(orc.dp(gl_log[k, i]), float(call[0]), float(call[1]))
        ref_pd, ref_labels = None, None
        if ptxt is not None:
            p = os.path.join(d, "in.pos")
            open(p, "w").write(ptxt)
            ref_pd = np.empty(n_sites)
            R.ref_read_dist(p.encode(), int(header), n_sites, orc.dp(ref_pd))
            buf = C.create_string_buffer(n_sites * 128)
            n = R.ref_read_labels(p.encode(), int(header), buf, 128, n_sites)
            assert n == n_sites
            ref_labels = [buf.raw[s * 128:(s + 1) * 128].split(b"\0")[0].decode() for s in range(n_sites)]
    gl = gl_log.copy()
    maf = np.empty(n_sites)
    expg = np.empty((n_sites, n_ind))
    R.ref_preprocess(orc.dp(gl), n_ind, n_sites, int(ignore_miss), orc.dp(maf), orc.dp(expg))

    # ---- oracle on the same input; must equal the reference wherever the reference can be built ----
    if text_mode is None:
        o = orc.Oracle(raw, pd, log_scale=log_scale, ignore_miss_data=ignore_miss, max_kb_dist=max_kb,
                       max_snp_dist=max_snp, min_maf=min_maf, n_threads=4, call_geno=call, rnd_sample=rnd_sample,
                       seed=seed)
    else:
        o = orc.Oracle(gl_orc, pd, ignore_miss_data=ignore_miss, max_kb_dist=max_kb, max_snp_dist=max_snp,
                       min_maf=min_maf, n_threads=4, already_normalised_log=True, call_geno=call,
                       rnd_sample=rnd_sample, seed=seed)
    assert np.array_equal(o.gl_log, gl_log, equal_nan=True), "reader: oracle != reference"
    assert np.array_equal(o.maf, maf, equal_nan=True) and np.array_equal(o.gl, gl) and np.array_equal(o.expg, expg)
    if ref_pd is not None:
        assert np.array_equal(ref_pd, pd), "pos_dist: host mirror != reference read_dist"
    rec = o.run()
    hap = np.empty((len(rec), 4))
    n_iter = np.empty(len(rec), dtype=np.uint64)
    n_data = np.empty(len(rec), dtype=np.uint64)
    for k, r in enumerate(rec):
        h = np.zeros(4)
        n = C.c_uint64()
        n_iter[k] = R.ref_haplo_freq(orc.dp(h), C.byref(n), orc.dp(gl[r["s1"]]), orc.dp(gl[r["s2"]]), maf[r["s1"]],
                                     maf[r["s2"]], n_ind, int(ignore_miss))
        hap[k], n_data[k] = h, n.value
    assert np.array_equal(hap, rec["hap"], equal_nan=True), "EM: oracle != reference haplo_freq"
    assert np.array_equal(n_iter, rec["n_iter"]) and np.array_equal(n_data, rec["n_ind_data"])

    # ---- ngsLD.cpp's own lines (compiled by build_ref.sh): D / D' / r2 / hap_maf / chi2 of every pair, the s2 walk ----
    def bits(a):
        a = np.ascontiguousarray(a)
        return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)
    stats = np.empty((len(rec), 5))
    chi2 = np.empty(len(rec), dtype=np.float32)
    for k in range(len(rec)):
        c = C.c_float()
        R.ref_pair_stats(orc.dp(np.ascontiguousarray(hap[k])), orc.dp(stats[k, 0:1]), orc.dp(stats[k, 1:2]),
                         orc.dp(stats[k, 2:3]), orc.dp(stats[k, 3:5]), C.byref(c))
        chi2[k] = c.value
    for col, got in (("D", stats[:, 0]), ("Dp", stats[:, 1]), ("r2", stats[:, 2]), ("hap_maf", stats[:, 3:5]), ("chi2", chi2)):
        assert np.array_equal(bits(rec[col].copy()), bits(got.copy())), f"{col}: oracle != the reference's lines"
    ref_walk = None
    if rnd_sample >= 1:
        pdw = np.ascontiguousarray(o.pos_dist.copy())
        mafw = np.ascontiguousarray(maf.copy())
        s2buf, dbuf = np.empty(n_sites, dtype=np.uint64), np.empty(n_sites)
        ws1, ws2, wd = [], [], []
        for s1 in range(n_sites):
            n = R.ref_walk(n_sites, orc.dp(pdw), orc.dp(mafw), max_kb, max_snp, float(min_maf), s1,
                           s2buf.ctypes.data_as(C.POINTER(C.c_uint64)), orc.dp(dbuf), n_sites)
            ws1 += [s1] * n
            ws2 += list(s2buf[:n])
            wd += list(dbuf[:n])
        ref_walk = (np.array(ws1, dtype=np.uint64), np.array(ws2, dtype=np.uint64), np.array(wd, dtype=np.float64))
        assert np.array_equal(ref_walk[0], rec["s1"]) and np.array_equal(ref_walk[1], rec["s2"]), "walk: oracle != reference"
        assert np.array_equal(bits(ref_walk[2]), bits(rec["dist"].copy())), "dist: oracle != reference"

    fx = dict(
        raw=raw, geno_text=np.array(gtext if gtext is not None else ""),
        text_mode=np.array(text_mode if text_mode is not None else ""),
        call_geno=np.array(list(call) if call is not None else [], dtype=np.float64),
        rnd_sample=np.array(rnd_sample), seed=np.array(seed),
        ref_gl_log=gl_log if text_mode is not None else np.zeros(0),
        pos_text=np.array(ptxt if ptxt is not None else ""), has_pos=np.array(ptxt is not None),
        header=np.array(header), log_scale=np.array(log_scale), ignore_miss=np.array(ignore_miss),
        max_kb=np.array(max_kb), max_snp=np.array(max_snp), min_maf=np.array(min_maf),
        ref_reader_sha=np.array(sha(gl_log)), ref_gl_sha=np.array(sha(gl)), ref_expg_sha=np.array(sha(expg)),
        ref_maf=maf, ref_hap=hap, ref_n_iter=n_iter, ref_n_ind_data=n_data,
        ref_pos_dist=ref_pd if ref_pd is not None else np.zeros(0),
        ref_labels=np.array(ref_labels if ref_labels is not None else [], dtype=str),
        orc_s1=rec["s1"], orc_s2=rec["s2"], orc_dist=rec["dist"], orc_r2pear=rec["r2pear"], orc_D=rec["D"],
        orc_Dp=rec["Dp"], orc_r2=rec["r2"], orc_hap_maf=rec["hap_maf"], orc_chi2=rec["chi2"],
        ref_D=stats[:, 0].copy(), ref_Dp=stats[:, 1].copy(), ref_r2=stats[:, 2].copy(), ref_hap_maf=stats[:, 3:5].copy(),
        ref_chi2=chi2,
        ref_walk_s1=ref_walk[0] if ref_walk else np.zeros(0, dtype=np.uint64),
        ref_walk_s2=ref_walk[1] if ref_walk else np.zeros(0, dtype=np.uint64),
        ref_walk_dist=ref_walk[2] if ref_walk else np.zeros(0),
    )
    if with_text:
        flags = ["--max_kb_dist", str(max_kb), "--max_snp_dist", str(max_snp), "--min_maf", repr(min_maf)]
        if log_scale:
            flags.append("--log_scale")
        if ignore_miss:
            flags.append("--ignore_miss_data")
        if text_mode == "probs" or call is not None:   # --call_geno needs --probs even for binary input (parse_args.cpp:178)
            flags.append("--probs")
        if rnd_sample < 1:
            flags += ["--rnd_sample", repr(rnd_sample), "--seed", str(seed)]
        if call is not None:
            flags += ["--call_geno", "--N_thresh", repr(float(call[0])), "--call_thresh", repr(float(call[1]))]
        for tag, extra in (("std", []), ("ext", ["--extend_out"])):
            txt = run_cli(raw, ptxt, flags + extra, header, gtext)
            lines = txt.splitlines(keepends=True)
            body = "".join(sorted(lines[1:]))
            # every row (and the header) once more through the reference's OWN fprintf lines (ngsLD.cpp:77, :314-351)
            buf = C.create_string_buffer(4096)
            n = R.ref_print_header(buf, 4096, int(tag == "ext"))
            assert buf.raw[:n].decode() == lines[0], "header: oracle != the reference's fprintf"
            assert len(lines) - 1 == len(rec)
            for k, r in enumerate(rec):
                l1 = ref_labels[r["s1"]] if ref_labels is not None else None
                l2 = ref_labels[r["s2"]] if ref_labels is not None else None
                n = R.ref_format_row(buf, 4096, l1.encode() if l1 is not None else None, l2.encode() if l2 is not None else None,
                                     float(r["dist"]), float(r["r2pear"]), orc.dp(np.ascontiguousarray(hap[k])),
                                     int(n_data[k]), float(maf[r["s1"]]), float(maf[r["s2"]]), int(n_iter[k]), int(tag == "ext"))
                assert buf.raw[:n].decode() == lines[1 + k], f"row {k}: oracle != the reference's fprintf"
            fx[f"ref_tsv_{tag}_rows_equal"] = np.array(True)
            fx[f"orc_tsv_{tag}_header"] = np.array(lines[0])
            fx[f"orc_tsv_{tag}_md5"] = np.array(hashlib.md5((lines[0] + body).encode()).hexdigest())
            if len(rec) <= 3000:
                fx[f"orc_tsv_{tag}"] = np.array(txt)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"{name:24s} {n_sites:5d} sites x {n_ind:4d} ind  {len(rec):6d} pairs  mean nIter {rec['n_iter'].mean():6.2f}"
          f"  max {rec['n_iter'].max():3d}  nan-r2 {np.isnan(rec['r2']).sum():3d}  {os.path.getsize(path) / 1024:7.1f} KiB")


def main():
    orc.build()
    assert orc.build_ref(), "need the reference tree to (re)generate goldens"

    # F1: BASELINE configs[0] shape -- 24 ind x 100 sites, depth 2 (slow convergence, some nIter == 100)
    raw = synth.make_gl_numpy(100, 24, seed=1, depth=2.0)
    chrs, pos = synth.make_positions(100, 1)
    make("f1_c1_24x100", raw, chrs, pos)
    # F4: the same GLs as natural logs, --log_scale
    with np.errstate(divide="ignore"):
        make("f4_logscale", np.log(raw), chrs, pos, log_scale=True)
    # F2: two chromosomes, 3-column pos file (label keeps the extra TAB), windows
    raw = synth.make_gl_numpy(60, 24, seed=2, depth=4.0)
    chrs, pos = synth.make_positions(60, 2, n_chr=2)
    make("f2_twochr_all", raw, chrs, pos, extra_col=True)
    make("f2_twochr_kb5", raw, chrs, pos, extra_col=True, max_kb=5)
    make("f2_twochr_snp7", raw, chrs, pos, extra_col=True, max_snp=7, header=True)
    # F3: degenerate sites, with and without --ignore_miss_data
    n_ind = 12
    raw = synth.make_gl_numpy(10, n_ind, seed=61, depth=3.0)
    raw[2] = np.array([1.0, 0.0, 0.0])
    raw[3] = np.array([0.0, 0.0, 1.0])
    g = np.random.default_rng(61).integers(0, 3, size=n_ind)
    raw[4] = np.eye(3)[g]
    raw[6] = raw[4]
    raw[8] = 1.0 / 3.0                                   # missing for everybody
    raw[9, ::2] = 0.25                                   # missing for half
    chrs, pos = synth.make_positions(10, 3)
    make("f3_degenerate", raw, chrs, pos)
    make("f3_degenerate_ignmiss", raw, chrs, pos, ignore_miss=True)
    # F5: --min_maf: s1 below -> row empty, s2 below -> skipped
    raw = synth.make_gl_numpy(100, 24, seed=5, depth=3.0)
    chrs, pos = synth.make_positions(100, 5)
    thr = float(np.round(np.quantile(orc.Oracle(raw).maf, 0.25), 3))
    make("f5_minmaf", raw, chrs, pos, min_maf=thr, max_kb=10)
    # F8: text (.gz) inputs -- beagle-like GL triples with header + name columns, called genotypes, and
    # --call_geno on a binary GL file
    raw = synth.make_gl_numpy(40, 20, seed=81, depth=3.0)
    raw /= raw.sum(axis=2, keepdims=True)
    raw[5, 3] = [0.0, 0.0, 1.0]                             # exact zeros: log(0) stays -inf in the text branch
    chrs, pos = synth.make_positions(40, 81)
    make("f8_text_probs", raw, chrs, pos, text_mode="probs")
    with np.errstate(divide="ignore"):
        make("f8_text_probs_log", np.log(raw), chrs, pos, text_mode="probs", log_scale=True, geno_header=False)
    called = np.random.default_rng(82).choice([-1, 0, 1, 2], size=(40, 20), p=[0.1, 0.45, 0.3, 0.15]).astype(float)
    make("f8_text_called", called, chrs, pos, text_mode="called", ignore_miss=True)
    raw = synth.make_gl_numpy(60, 30, seed=83, depth=4.0)
    chrs, pos = synth.make_positions(60, 83)
    make("f8_call_geno", raw, chrs, pos, call=(0.4, 0.9))
    make("f8_call_geno_default", raw, chrs, pos, call=(0.0, 0.0), ignore_miss=True)
    # F9: --rnd_sample / --seed (per-row Tausworthe streams), alone and with the other filters
    raw = synth.make_gl_numpy(80, 24, seed=91, depth=4.0)
    chrs, pos = synth.make_positions(80, 91, n_chr=2)
    make("f9_rnd_sample", raw, chrs, pos, rnd_sample=0.4, seed=7)
    thr = float(np.round(np.quantile(orc.Oracle(raw).maf, 0.2), 3))
    make("f9_rnd_sample_filters", raw, chrs, pos, rnd_sample=0.25, seed=123456789, max_kb=6, min_maf=thr)
    # F6/F7: the benchmark n_ind values (slot / multi-wavefront paths of the kernel)
    for name, ns, ni, seed in (("f7_n100", 128, 100, 7), ("f6_n500", 48, 500, 6), ("f6_n1000", 24, 1000, 8),
                               ("f6_n2000", 12, 2000, 9)):
        raw = synth.make_gl_numpy(ns, ni, seed=seed, depth=10.0)
        chrs, pos = synth.make_positions(ns, seed)
        make(name, raw, chrs, pos, with_text=(ni <= 500))
    # F10 (round 5): matrices that are NOT SNP-called (the reference's README.md:73; its examples/test.sh feeds such input): 30 %
    # of the sites monomorphic in the population, three singletons (one heterozygous individual), one site without data for
    # anybody -- at the cohort sizes of every kernel family, so that each of them flags, and each shape of the device-side
    # replay (1 / 2 / 4 wavefronts per pair) settles, its share of such pairs
    for name, ns, ni, seed in (("f10_unfiltered_n100", 48, 100, 101), ("f10_unfiltered_n500", 40, 500, 105),
                               ("f10_unfiltered_n1000", 22, 1000, 110), ("f10_unfiltered_n2000", 12, 2000, 120)):
        raw = synth.make_gl_numpy(ns, ni, seed=seed, depth=10.0, mono_frac=0.3)
        rng = np.random.default_rng(seed)
        maf0 = orc.Oracle(raw).maf
        mono = np.flatnonzero(maf0 < 0.01)
        for s in mono[:3]:                                   # singletons: one individual with five of ten reads alternative
            raw[s, rng.integers(ni)] = [0.01 ** 5 * 0.99 ** 5, 0.5 ** 10, 0.99 ** 5 * 0.01 ** 5]
        raw[ns // 2] = 1.0                                   # no data for anybody
        chrs, pos = synth.make_positions(ns, seed)
        make(name, raw, chrs, pos, with_text=(ni <= 500))
    make("f10_unfiltered_n100_ignmiss", synth.make_gl_numpy(48, 100, seed=131, depth=3.0, mono_frac=0.3),
         *synth.make_positions(48, 131), ignore_miss=True)


if __name__ == "__main__":
    ONLY = sys.argv[1:]
    main()
