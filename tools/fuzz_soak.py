"""Dev soak (GPU box): the seeded fuzz of tests/test_gpu_fuzz.py over many more cases than the test-suite runs.
python tools/fuzz_soak.py [first] [last]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401  (HIP runtime load order, see tests/conftest.py)
from ngsld_amd import capi
from oracle import orc
from test_gpu_fuzz import _case, pick_min_maf
from util import MAF_TOL, check_records, close, pearson_tolerance

first, last = int(sys.argv[1]) if len(sys.argv) > 1 else 48, int(sys.argv[2]) if len(sys.argv) > 2 else 400
eng = capi.Engine(0)
bad = 0
pairs = 0
for k in range(first, last):
    raw, pd, kw, call = _case(k)
    o0 = orc.Oracle(raw, pd, log_scale=kw["log_scale"], call_geno=call)
    min_maf = pick_min_maf(o0.maf, k)
    o = orc.Oracle(raw, pd, min_maf=min_maf, n_threads=16, call_geno=call, **kw)
    rec = o.run()
    try:
        eng.set_exact_store(2 if k >= 40_000 else 1)   # (un-called cases: replayed on the device from the first flagged pair on)
        eng.set_geno_raw(raw, log_scale=kw["log_scale"], ignore_miss_data=kw["ignore_miss_data"], call_geno=call)
        eng.set_pos_dist(pd)
        assert np.all(close(eng.maf(), o.maf, MAF_TOL))
        n = eng.plan(kw["max_kb_dist"], kw["max_snp_dist"], min_maf, kw["ignore_miss_data"], True, kw["rnd_sample"], kw["seed"])
        assert n == len(rec)
        s1, s2, std, ext = eng.run()
        assert np.array_equal(s1, rec["s1"]) and np.array_equal(s2, rec["s2"])
        check_records(std, ext, rec, pearson_tol=pearson_tolerance(o.gl, s1, s2))
        pairs += n
    except (AssertionError, capi.NgsldError) as e:
        bad += 1
        print(f"case {k}: n_ind {raw.shape[1]} n_sites {raw.shape[0]} FAILED: {str(e)[:300]}")
from util import REPORT
print(f"fuzz soak: cases {first}..{last - 1}, {pairs} pairs compared, {bad} failing cases; beyond 1e-9: {REPORT['over_tol']}, "
      f"degenerate pairs held to bit equality: {REPORT['degenerate']}, largest difference {REPORT['max_diff']:.3e}")
sys.exit(1 if bad else 0)
