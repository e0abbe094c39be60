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


def test_every_pair_of_configs1_equals_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "parity_config.py"), "c1"], capture_output=True, text=True,
                       timeout=1500)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert line, r.stderr[-2000:]
    d = json.loads(line[-1])
    assert r.returncode == 0, d
    assert d["pairs"] == 12_497_500 and d["pairs_and_order_equal"]
    assert d["n_iter_equal"] and d["sample_size_equal"] and d["within_tolerances"]
    assert max(d[k] for k in d if k.startswith("max_abs_diff_")) <= 1e-9
    print("every pair of configs[1]:", {k: d[k] for k in d if k.startswith("max_abs_diff_") or k.startswith("pairs")})
