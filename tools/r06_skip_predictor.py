#!/usr/bin/env python
"""Round 6 (CPU, the tests' checker): which pairs of a matrix that is NOT SNP-called does the exact-order replay settle, and how
well does a PER-SITE mark predict them?  The mark (ld_prep.hip, site_skip_kernel): the one-locus EM of the site from its est_maf
frequency, run until its own step is below EPSILON, ends below tau.  Per tau: the share of the flagged pairs that have a marked
site (caught), the share of ALL pairs that have one without being flagged (marked without need), and a cost model
(2.0 ns of pair-kernel EM saved per caught pair, 6.6 ns of replay spent per pair marked without need).

    python tools/r06_skip_predictor.py [n_sites] [n_ind]  ->  profiles/r06/skip/predictor.txt
"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from ngsld_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402  (test infrastructure: this tool is not product code)

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
n_ind = int(sys.argv[2]) if len(sys.argv) > 2 else 500


def step(gl, m):
    pr = np.stack([(1 - m) ** 2, 2 * m * (1 - m), m * m], 1)[:, None, :]
    pp = gl * pr
    pp /= pp.sum(2, keepdims=True)
    return (pp[:, :, 1] + 2 * pp[:, :, 2]).sum(1) / (2 * gl.shape[1])


for tag, kw in (("20 % monomorphic sites", dict(mono_frac=0.2)), ("log-uniform spectrum", dict(sfs=True))):
    raw = synth.make_gl_numpy(n_sites, n_ind, seed=11, depth=10.0, **kw)
    o = orc.Oracle(raw, n_threads=8)
    res = o.run()
    hap, Dp, r2 = res["hap"], res["Dp"], res["r2"]
    hm0, hm1 = 1 - (hap[:, 0] + hap[:, 1]), 1 - (hap[:, 0] + hap[:, 2])
    q0, q1 = np.minimum(abs(hm0), abs(1 - hm0)), np.minimum(abs(hm1), abs(1 - hm1))
    with np.errstate(all="ignore"):
        flag = ~(q0 >= 2.0 ** -30) | ~(q1 >= 2.0 ** -30) | ~(2.0 ** -49 * (1 / q0 + 1 / q1) * np.fmax(abs(Dp), r2) <= 2.5e-10)
    s1, s2 = res["s1"].astype(int), res["s2"].astype(int)
    gl = o.gl.reshape(n_sites, n_ind, 3)
    maf = o.maf.copy()
    m = np.minimum(maf, 1 - maf)
    glf = np.where((maf > 0.5)[:, None, None], gl[:, :, ::-1], gl)
    done, q = np.zeros(n_sites, bool), m.copy()
    for it in range(100):
        mn = step(glf, m)
        conv = np.abs(mn - m) < 1e-5
        m = mn
        q[~done] = mn[~done]
        done |= conv
        if done.all():
            break
    mm = np.minimum(q[s1], q[s2])
    print(f"{tag}: {n_sites} sites x {n_ind} individuals, all {len(res)} pairs; flagged {flag.mean():.4f}; "
          f"executed iterations: flagged {res['n_iter'][flag].mean() + 1:.2f}, others {res['n_iter'][~flag].mean() + 1:.2f}")
    for tau in (1e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4):
        pred = mm < tau
        tp, fp = (pred & flag).sum(), (pred & ~flag).sum()
        print(f"  tau {tau:g}: {int((q < tau).sum())} sites marked; caught {tp / flag.sum():.4f} of the flagged pairs; "
              f"marked without need {fp / len(flag):.5f} of all pairs; model gain {tp / len(flag) * 2.0 - fp / len(flag) * 6.6:.3f} ns a pair")
