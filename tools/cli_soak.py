"""Dev soak (GPU box): the drop-in binary against the reference's own program (oracle/_ref ref_main) over the fuzz generator's
cases written out as files -- same argv, same files; header equal, body equal as sorted lines.
python tools/cli_soak.py [first] [last] [text]    (tests/test_gpu_vs_ref_program.py is the same comparison on a fixed list;
"text": the text (.gz) genotype cases -- called genotypes, likelihood triples, --call_geno -- instead of the binary ones)"""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_gpu_vs_ref_program import both_programs, case_files, same_tsv, text_case_files

first, last = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 300
make = text_case_files if len(sys.argv) > 3 and sys.argv[3] == "text" else case_files
bad = rows = 0
for k in range(first, last):
    with tempfile.TemporaryDirectory() as d:
        flags, rec, n_sites = make(k, d)
        got, want = both_programs(flags, rec, n_sites, d, threads=1 + k % 3)
    rows += max(0, len(want.splitlines()) - 1)
    why = same_tsv(got, want)
    if why is not None:
        bad += 1
        print(f"case {k}: {why}\n  {' '.join(flags)}", flush=True)
print(f"cli soak ({make.__name__}): cases {first}..{last - 1}, {rows} rows through both programs, {bad} cases differ")
sys.exit(1 if bad else 0)
