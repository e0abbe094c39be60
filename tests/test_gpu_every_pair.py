"""GPU: EVERY pair of a BASELINE configuration against the oracle (tools/parity_config.py): configs[1], 5,000 sites x 100
individuals, all 12,497,500 pairs -- nIter and sample_size equal on every pair, hap / D / D' / r2 / r2_ExpG within 1e-9 on
every pair.  ~20 s of oracle on the box's real cores; the driver sees what the builder's profiles/ claim."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _parity(*args):
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "parity_config.py"), *args], capture_output=True, text=True,
                       timeout=1500)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert line, r.stderr[-2000:]
    d = json.loads(line[-1])
    assert r.returncode == 0, d
    return d


def test_every_pair_of_configs1_equals_the_oracle():
    d = _parity("c1")
    assert d["pairs"] == 12_497_500 and d["pairs_and_order_equal"]
    assert d["n_iter_equal"] and d["sample_size_equal"] and d["within_tolerances"]
    assert max(d[k] for k in d if k.startswith("max_abs_diff_")) <= 1e-9
    print("every pair of configs[1]:", {k: d[k] for k in d if k.startswith("max_abs_diff_") or k.startswith("pairs")})


@pytest.mark.parametrize("which,rows,min_pairs", [("c3", "24", 1_100_000), ("c4", "1500", 700_000)])
def test_first_rows_of_the_large_cohort_configurations_equal_the_oracle(which, rows, min_pairs):
    """configs[3] at its real size (50,000 x 1,000 all pairs: two wavefronts per pair, tiled workgroup order -- the matrix is
    1.2 GB) and configs[4]'s shape (60,000 x 2,000, 500 kb window: four wavefronts per pair): every pair of the first rows."""
    d = _parity(which, rows)
    assert d["pair_kernel"] == "multi" and d["pairs"] >= min_pairs and d["pairs_and_order_equal"]
    assert d["n_iter_equal"] and d["sample_size_equal"] and d["within_tolerances"]
    assert max(d[k] for k in d if k.startswith("max_abs_diff_")) <= 1e-9
    print(f"every pair of the first {rows} rows of {which}:", {k: d[k] for k in d if k.startswith("max_abs_diff_") or k.startswith("pairs")})


def test_first_rows_of_the_headline_configuration_equal_the_oracle():
    """configs[2] -- the configuration the metric is quoted on: 100,000 x 500, 100 kb window, the run kernel at eight slots --
    every pair of its first 3,000 rows (~3.0e6 pairs, ~10 s of oracle): nIter / sample_size exact, the rest within 1e-9."""
    d = _parity("c2", "3000")
    assert d["pair_kernel"] == "run" and d["pairs"] >= 2_900_000 and d["pairs_and_order_equal"]
    assert d["n_iter_equal"] and d["sample_size_equal"] and d["within_tolerances"]
    assert max(d[k] for k in d if k.startswith("max_abs_diff_")) <= 1e-9
    print("every pair of the first 3,000 rows of configs[2]:", {k: d[k] for k in d if k.startswith("max_abs_diff_") or k.startswith("pairs")})


def test_first_rows_of_the_headline_shape_not_snp_called_equal_the_oracle():
    """configs[2]'s shape with 20 % of the sites monomorphic (README.md:73): every pair of the first 2,000 rows -- a third of them
    pairs the reference's own rounding decides, replayed on the device -- nIter / sample_size exact, everything within 1e-9, NaN
    and inf where the reference has them, the degenerate pairs bit for bit (tests/util.py)."""
    d = _parity("c2mono", "2000")
    assert d["pair_kernel"] == "run" and d["pairs"] >= 1_900_000 and d["pairs_and_order_equal"]
    assert d["n_iter_equal"] and d["sample_size_equal"] and d["within_tolerances"]
    assert d["replay"]["pairs_on_device"] > d["pairs"] // 5 and d["replay"]["pairs_on_host"] * 10_000 <= d["pairs"]
    print("every pair of the first 2,000 rows of configs[2]'s shape, 20 % monomorphic sites:",
          {k: d[k] for k in d if k.startswith("max_abs_diff_") or k.startswith("pairs") or k == "replay"})


# (round 5: c2mono -- the same shape NOT SNP-called, 20 % of the sites monomorphic: a third of its rows are pairs the reference's own
# rounding decides, replayed in its operation order on the device; the patched reference main hands over normal-space values,
# whose replay is on the device from the first pair on)
@pytest.mark.parametrize("args,rows", [(("c2", "4000"), 3_000_000), (("c2mono", "4000"), 3_000_000)])
def test_whole_table_of_both_programs(args, rows):
    """tools/cli_vs_ref_config.py: the drop-in binary, the reference's OWN program (oracle/_ref ref_main: ngsLD.cpp's main +
    calc_pair_LD compiled minus the GSL statements, all host cores) and the reference's main with the binding compiled in, over
    the same argv and files -- the first 4,000 sites of configs[2]'s matrix (3.7e6 extended rows): first line equal, sorted
    bodies byte-identical.  (ALL of configs[1] -- 12,497,500 rows, 2 GB of TSV, 47 s of the reference program on 16 threads --
    is the same tool with `c1`: profiles/r04/cli_vs_ref_c1.json; it ran inside the suite until the suite reached eight minutes.)"""
    from util import have_ref_program
    if not have_ref_program():
        pytest.skip("oracle/_ref predates ref_main (rebuild with oracle/build_ref.sh)")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "cli_vs_ref_config.py"), *args], capture_output=True, text=True,
                       timeout=1500)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert line, r.stderr[-2000:]
    d = json.loads(line[-1])
    assert r.returncode == 0 and d["identical"] and d["first_line_equal"], d
    assert d["rows"] == d["rows_hip"] and d["rows"] >= rows
    print("whole table through both programs:", d)
