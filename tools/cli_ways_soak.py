"""Dev soak (GPU box): the drop-in binary's own ways of cutting / degrading a job, against the reference's own program over the fuzz
generator's cases as files (tests/test_gpu_vs_ref_program.py::test_streamed_and_multi_part_runs_through_both_programs is the same on a
fixed list): streamed row slabs, several parts on one device, text in forced groups (alone and inside slabs), no room for the exact
store, a host that cannot pin, the host formatter.
python tools/cli_ways_soak.py [first] [last]"""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_gpu_vs_ref_program import both_programs, case_files, same_tsv

first, last = int(sys.argv[1]) if len(sys.argv) > 1 else 40100, int(sys.argv[2]) if len(sys.argv) > 2 else 40400
bad = rows = 0
per_way = {}
for k in range(first, last):
    with tempfile.TemporaryDirectory() as d:
        flags, rec, n_sites = case_files(k, d)
        slab = str(max(2, n_sites // (2 + k % 3)))
        ways = [("slabs", {"hip_env": {"NGSLD_TEST_SLAB_SITES": slab}}),
                ("parts", {"hip_flags": ("--devices", "0,0,0")}),
                ("groups", {"hip_env": {"NGSLD_TEST_TEXT_GROUP_PAIRS": str(1 + 37 * (k % 11))}}),
                ("slabs+groups", {"hip_env": {"NGSLD_TEST_SLAB_SITES": slab, "NGSLD_TEST_TEXT_GROUP_PAIRS": "150"}}),
                ("no store", {"hip_env": {"NGSLD_TEST_EXACT_STORE_NO_ROOM": "1"}}),
                ("no pinning", {"hip_env": {"NGSLD_TEST_PIN_LIMIT_BYTES": "4096"}}),
                ("host text", {"hip_env": {"NGSLD_HOST_TEXT": "1"}})]
        name, kw = ways[k % len(ways)]
        got, want = both_programs(flags, rec, n_sites, d, threads=1 + k % 3, **kw)
    rows += max(0, len(want.splitlines()) - 1)
    per_way[name] = per_way.get(name, 0) + 1
    why = same_tsv(got, want)
    if why is not None:
        bad += 1
        print(f"case {k} ({name}): {why}\n  {' '.join(flags)}", flush=True)
print(f"cli ways soak: cases {first}..{last - 1} ({per_way}), {rows} rows through both programs, {bad} cases differ")
sys.exit(1 if bad else 0)
