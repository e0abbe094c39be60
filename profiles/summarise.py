#!/usr/bin/env python
"""Turn one profiles/collect_pmc.sh output directory into summary.json (+ the hbm_traffic.json bench.py reads).

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide (16 B/lane) coalesced stream -- the pair kernel's site copies are exactly that
(global_load_lds_dwordx4), so its FETCH_SIZE is doubled.  Other widths are uncalibrated in the guide, so the
prep kernel (8 B/lane, known byte counts: reads n_sites*n_ind*24 B once, writes n_sites*np*24 B once) is
profiled in the same passes and its measured/known ratio is recorded next to the numbers.
"""
import csv
import json
import os
import sys

d = sys.argv[1]
out = {}


def counters(path):
    res = {}
    if not os.path.exists(path):
        return res
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = "pair" if "pair_ld" in r["Kernel_Name"] else "prep" if "prep_sites" in r["Kernel_Name"] else None
            if k:
                res.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}


bench = {}
try:
    bench = json.load(open(os.path.join(d, "bench_under_rocprof.json")))
except (OSError, ValueError):
    pass
with open(os.path.join(d, "kernel_stats.csv")) as fh:
    for r in csv.DictReader(fh):
        if "pair_ld" in r["Name"]:
            out["pair_kernel"] = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
        if "prep_sites" in r["Name"]:
            out["prep_kernel"] = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
allc = {}
for f in sorted(os.listdir(d)):
    if f.startswith("pmc_") and f.endswith(".csv"):
        for k, cs in counters(os.path.join(d, f)).items():
            allc.setdefault(k, {}).update(cs)
out["counters_per_launch"] = allc
cfg = bench.get("config", {})
if "pair" in allc and "FETCH_SIZE" in allc["pair"]:
    p = allc["pair"]
    fetch = p["FETCH_SIZE"] * 1024 * 2            # wide-stream correction (guide §HBM)
    write = p.get("WRITE_SIZE", 0.0) * 1024
    out["pair_hbm_bytes_per_launch"] = {"fetch_corrected": fetch, "write_raw": write, "total": fetch + write}
    pairs = cfg.get("pairs_per_step")
    if pairs:
        out["pair_hbm_bytes_per_pair"] = (fetch + write) / pairs
        out["algorithmic_bytes_per_pair"] = bench.get("roofline", {}).get("algorithmic_bytes_per_pair")
    if "SQ_ACTIVE_INST_VALU" in p and "GRBM_GUI_ACTIVE" in p:
        cyc_xcd = p["GRBM_GUI_ACTIVE"] / 8.0
        out["pair_derived"] = {
            "clock_GHz": cyc_xcd / (out["pair_kernel"]["avg_ms"] * 1e6) if "pair_kernel" in out else None,
            "valu_busy_frac": p["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc_xcd),
            "waves_per_simd_avg": p["SQ_WAVE_CYCLES"] * 4 / (1024 * cyc_xcd),
            "wait_mem_frac_of_wave_cycles": p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"],
            "issue_stall_frac_of_wave_cycles": p["SQ_WAIT_INST_ANY"] / p["SQ_WAVE_CYCLES"],
            "valu_insts_per_pair": p["SQ_INSTS_VALU"] / pairs if pairs else None,
        }
if "prep" in allc and cfg:
    n_sites, n_ind = cfg.get("n_sites_total"), 500
    known_r, known_w = n_sites * n_ind * 24.0, n_sites * 512 * 24.0
    q = allc["prep"]
    out["prep_calibration"] = {"known_read_bytes": known_r, "known_write_bytes": known_w,
                               "FETCH_SIZE_bytes_raw": q.get("FETCH_SIZE", 0) * 1024,
                               "WRITE_SIZE_bytes_raw": q.get("WRITE_SIZE", 0) * 1024,
                               "fetch_measured_over_known": q.get("FETCH_SIZE", 0) * 1024 / known_r,
                               "write_measured_over_known": q.get("WRITE_SIZE", 0) * 1024 / known_w}
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
