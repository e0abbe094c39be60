"""Dev soak (GPU box): device-side TSV against the host formatter over the fuzz generator's cases.
python tools/text_soak.py [first] [last]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401
from ngsld_amd import capi
from test_gpu_fuzz import _case

first, last = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 300
eng = capi.Engine(0)
bad = rows_total = fallbacks_total = patched = on_host = 0
for k in range(first, last):
    raw, pd, kw, call = _case(k)
    n_sites = raw.shape[0]
    labels = [f"chr{1 + s % 3}:{1000 + 7 * s}" for s in range(n_sites)]
    eng.set_geno_raw(raw, log_scale=kw["log_scale"], ignore_miss_data=kw["ignore_miss_data"], call_geno=call)
    eng.set_pos_dist(pd)
    n = eng.plan(kw["max_kb_dist"], kw["max_snp_dist"], 0.0, kw["ignore_miss_data"], True, kw["rnd_sample"], kw["seed"])
    maf = eng.maf()
    s1, s2, std, ext = eng.run()
    eng.set_text_output(labels)
    text, fallbacks = eng.run_text()
    eng.set_text_output(None, enable=False)
    info = eng.replay_info()
    patched += info["text_rows_patched"]
    on_host += info["pairs_on_host"]
    fallbacks_total += fallbacks
    if fallbacks:
        continue          # the batch went out as records: nothing to compare
    want = []
    for i in range(len(s1)):
        a, b = int(s1[i]), int(s2[i])
        dist = float(np.sum(pd[a + 1:b + 1]))
        want.append(capi.format_pair(labels[a], labels[b], dist, std[i], ext[i], maf[a], maf[b]))
    want = "".join(want).encode()
    rows_total += len(s1)
    if text != want:
        bad += 1
        rows_g, rows_w = text.split(b"\n"), want.split(b"\n")
        for g, w in zip(rows_g, rows_w):
            if g != w:
                print(f"case {k}: first differing row\n  device {g}\n  host   {w}")
                break
print(f"text soak: cases {first}..{last - 1}, {rows_total} rows compared, {fallbacks_total} batches fell back to records, "
      f"{on_host} pairs replayed on the host of which {patched} rows overwritten in the host's text, {bad} differing cases")
sys.exit(1 if bad else 0)
