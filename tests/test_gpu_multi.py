"""GPU: the multi-GPU front end (ngsld_amd.multi).  The box has one GPU, so two ranks share it over gloo
(NGSLD_BENCH_ONE_DEVICE=1); what is tested is everything that differs from the single-process path: the
broadcast, the row split, slab-local indices and labels, the master-stream offset of --rnd_sample, the shards."""
import os
import subprocess
import sys

import pytest

from ngsld_amd import capi
from util import Fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,world", [("f2_twochr_kb5", 2), ("f9_rnd_sample_filters", 2), ("f8_text_probs", 2),
                                        ("f5_minmaf", 3)])
def test_shards_concatenate_to_single_gpu_output(name, world, tmp_path):
    fx = Fixture(name)
    g, p = fx.write_inputs(str(tmp_path))
    flags = fx.cli_flags(True)
    single = subprocess.run([capi.CLI_PATH, "--geno", g, "--n_ind", str(fx.n_ind), "--n_sites", str(fx.n_sites),
                             "--verbose", "0", "--posH" if fx.header else "--pos", p] + flags,
                            capture_output=True, text=True)
    assert single.returncode == 0, single.stderr
    out = str(tmp_path / "multi.ld")
    env = dict(os.environ, NGSLD_BENCH_ONE_DEVICE="1", PYTHONPATH=capi.REPO_DIR)
    port = 29600 + (os.getpid() % 300)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "ngsld_amd.multi",
                        "--geno", g, "--n_ind", str(fx.n_ind), "--n_sites", str(fx.n_sites), "--verbose", "0",
                        "--posH" if fx.header else "--pos", p, "--out", out] + flags,
                       capture_output=True, text=True, env=env, cwd=capi.REPO_DIR, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    merged = "".join(open(f"{out}.rank{k}").read() for k in range(world))
    assert merged == single.stdout
