"""Dev probe (GPU box): matrices of values at the edges of the double range (zeros, denormals, 1e300, negatives, -0.0, both
scales) through ngsld_set_geno_raw -- the device's prep must fail where the oracle's reader fails ("NaN found", read_data.cpp:
106-116 as pinned by tests/test_oracle_vs_ref.py) and give its allele frequencies (1e-12) where it does not.
python tools/probe_special_values.py [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ngsld_amd import capi  # noqa: E402
from oracle import orc  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(5)
special = np.array([0.0, 1.0, 1 / 3, 0.5, 1e-310, 5e-324, 1e-300, 1e300, 1.7e308, -1.0, -0.0, 1e-17, 1 - 1e-16, 2.0, 3.0])
eng = capi.Engine(0)
same_fail = same_ok = bad = 0
for trial in range(n_cases):
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
    try:
        with np.errstate(all="ignore"):
            o = orc.Oracle(raw, None, log_scale=log_scale)
        want = o.maf
    except ValueError:
        want = None
    try:
        eng.set_geno_raw(np.ascontiguousarray(raw), log_scale=log_scale)
        got = eng.maf()
    except capi.NgsldError as e:
        got = None
        msg = e.msg
    if (want is None) != (got is None):
        bad += 1
        print(f"case {trial}: oracle {'fails' if want is None else 'returns'}, device {'fails: ' + msg if got is None else 'returns'}")
    elif want is None:
        same_fail += 1
    elif np.allclose(got, want, rtol=0, atol=1e-12, equal_nan=True):
        same_ok += 1
    else:
        bad += 1
        print(f"case {trial}: maf differs by {np.nanmax(np.abs(got - want)):.3e}")
print(f"special values: {n_cases} matrices, {same_ok} equal allele frequencies, {same_fail} fail on both sides, {bad} differ")
sys.exit(1 if bad else 0)
