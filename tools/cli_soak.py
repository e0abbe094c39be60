"""Dev soak (GPU box): the drop-in binary against the reference's own program (oracle/_ref ref_main) over the fuzz generator's
cases written out as files -- same argv, same files; header equal, body equal as sorted lines.
python tools/cli_soak.py [first] [last]      (tests/test_gpu_vs_ref_program.py is the same comparison on a fixed list)"""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_gpu_vs_ref_program import both_programs, case_files, same_tsv

first, last = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = rows = 0
for k in range(first, last):
    with tempfile.TemporaryDirectory() as d:
        flags, rec, n_sites = case_files(k, d)
        got, want = both_programs(flags, rec, n_sites, d, threads=1 + k % 3)
    rows += max(0, len(want.splitlines()) - 1)
    why = same_tsv(got, want)
    if why is not None:
        bad += 1
        print(f"case {k}: {why}\n  {' '.join(flags)}", flush=True)
print(f"cli soak: cases {first}..{last - 1}, {rows} rows through both programs, {bad} cases differ")
sys.exit(1 if bad else 0)
