"""Dev tool (CPU): would the group kernel's lockstep tail on configs[1] shrink if a run's candidates were dealt to the
wavefronts in another ORDER?  The oracle's iteration counts of 40 rows x ~2,850 candidates of a 5,000 x 100 all-pairs matrix;
a model of one workgroup (four wavefronts, each taking the next four candidates of its run of 512 when it is done, a
generation costing the slowest of its four pairs + 1.2 iterations of staging); occupied wavefront time over the pairs' own.
Orders: as the kernel claims them (consecutive sites), sorted by the candidate's allele frequency (the one per-site
predictor at hand before the EM runs), and -- as the bound -- sorted by the true iteration count.
python tools/sim_lockstep_order.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ngsld_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402

STAGE = 1.2


def run_time(iters, order):
    t = np.zeros(4)
    for j in range(0, len(order), 4):
        w = int(np.argmin(t))
        t[w] += iters[order[j:j + 4]].max() + STAGE
    return t.max() * 4, t.sum()


def main():
    n_sites, n_ind = 5000, 100
    raw = synth.make_gl_numpy(n_sites, n_ind, 7, depth=10.0)
    o = orc.Oracle(raw, None, max_kb_dist=0, n_threads=8)
    rec = np.concatenate([o.run(r, r + 1)[["s1", "s2", "n_iter"]] for r in range(0, 4400, 110)])
    s1, s2, it = rec["s1"], rec["s2"], np.minimum(rec["n_iter"] + 1, 100).astype(float)
    maf = np.minimum(o.maf, 1 - o.maf)
    print(f"{len(rec)} pairs, mean executed iterations {it.mean():.2f}, percentiles 10/50/90/99: {np.percentile(it, [10, 50, 90, 99])}")

    def evaluate(name, key):
        occ = used = ideal = 0.0
        for r in np.unique(s1):
            m = s1 == r
            iters, cand = it[m], s2[m]
            for b in range(0, len(iters), 512):
                ii, cc = iters[b:b + 512], cand[b:b + 512]
                a, u = run_time(ii, key(cc, ii))
                occ += a
                used += u
                ideal += (ii.sum() + STAGE * len(ii)) / 4
        print(f"{name:36s} wavefront time / the pairs' own {occ / ideal:.4f}   (lockstep alone {used / ideal:.4f})")

    evaluate("as claimed (consecutive sites)", lambda cc, ii: np.arange(len(ii)))
    evaluate("sorted by the candidate's maf", lambda cc, ii: np.argsort(-maf[cc], kind="stable"))
    evaluate("sorted by the true iteration count", lambda cc, ii: np.argsort(-ii, kind="stable"))
    rho = np.corrcoef(np.argsort(np.argsort(it)), np.argsort(np.argsort(maf[s2])))[0, 1]
    print(f"rank correlation of a pair's iterations with its candidate's maf: {rho:+.3f}")


if __name__ == "__main__":
    main()
