"""Dev: one fuzz case in detail (GPU box).  python tools/fuzz_one.py <k>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401
from ngsld_amd import capi
from oracle import orc
from test_gpu_fuzz import _case, pick_min_maf
k = int(sys.argv[1])
raw, pd, kw, call = _case(k)
o0 = orc.Oracle(raw, pd, log_scale=kw["log_scale"], call_geno=call)
min_maf = pick_min_maf(o0.maf, k)
o = orc.Oracle(raw, pd, min_maf=min_maf, n_threads=4, call_geno=call, **kw)
rec = o.run()
eng = capi.Engine(0)
eng.set_geno_raw(raw, log_scale=kw["log_scale"], ignore_miss_data=kw["ignore_miss_data"], call_geno=call)
eng.set_pos_dist(pd)
n = eng.plan(kw["max_kb_dist"], kw["max_snp_dist"], min_maf, kw["ignore_miss_data"], True, kw["rnd_sample"], kw["seed"])
s1, s2, std, ext = eng.run()
print("case", k, "shape", raw.shape, kw, "call", call, "min_maf", min_maf, "pairs", n)
np.set_printoptions(precision=17, linewidth=200)
for name, got, want in (("Dp", std["Dp"], rec["Dp"]), ("r2", std["r2"], rec["r2"]), ("D", std["D"], rec["D"]), ("r2_ExpG", std["r2_ExpG"], rec["r2pear"])):
    with np.errstate(invalid="ignore"):
        d = np.abs(got - want)
    bad = np.flatnonzero(d > 1e-9)
    for i in bad[:4]:
        print(name, "pair", int(s1[i]), int(s2[i]), "got", got[i], "want", want[i], "\n   hap got", ext["hap"][i], "\n   hap want", rec["hap"][i],
              "\n   n_iter", ext["n_iter"][i], rec["n_iter"][i], "n", ext["n_ind_data"][i], rec["n_ind_data"][i], "hap_maf want", rec["hap_maf"][i],
              "D got/want", std["D"][i], rec["D"][i])
