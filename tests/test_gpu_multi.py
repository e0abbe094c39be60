"""GPU: the multi-GPU front end (ngsld_amd.multi).  The box has one GPU, so two ranks share it over gloo
(NGSLD_BENCH_ONE_DEVICE=1); what is tested is everything that differs from the single-process path: the
broadcast or the per-rank slab reads, the row split, slab-local indices and labels, the master-stream offset of --rnd_sample, the shards."""
import os
import subprocess
import sys

import pytest

from ngsld_amd import capi
from util import Fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,world,path", [("f2_twochr_kb5", 2, "slab"), ("f2_twochr_kb5", 3, "slab"),
                                             ("f2_twochr_snp7", 2, "slab"), ("f9_rnd_sample_filters", 2, None),
                                             ("f8_text_probs", 2, "broadcast"), ("f2_twochr_all", 2, "broadcast"),
                                             ("f5_minmaf", 3, None)])
def test_shards_concatenate_to_single_gpu_output(name, world, path, tmp_path):
    """path: the distribution that must be taken -- per-rank slab reads (windowed run on a binary file) or one broadcast
    (all pairs / text input); None = whichever the fixture's flags lead to."""
    fx = Fixture(name)
    g, p = fx.write_inputs(str(tmp_path))
    flags = fx.cli_flags(True)
    single = subprocess.run([capi.CLI_PATH, "--geno", g, "--n_ind", str(fx.n_ind), "--n_sites", str(fx.n_sites),
                             "--verbose", "0", "--posH" if fx.header else "--pos", p] + flags,
                            capture_output=True, text=True)
    assert single.returncode == 0, single.stderr
    out = str(tmp_path / "multi.ld")
    env = dict(os.environ, NGSLD_BENCH_ONE_DEVICE="1", PYTHONPATH=capi.REPO_DIR)
    if path:
        env["NGSLD_MULTI_EXPECT"] = path
    port = 29600 + (os.getpid() % 300)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "ngsld_amd.multi",
                        "--geno", g, "--n_ind", str(fx.n_ind), "--n_sites", str(fx.n_sites), "--verbose", "0",
                        "--posH" if fx.header else "--pos", p, "--out", out] + flags,
                       capture_output=True, text=True, env=env, cwd=capi.REPO_DIR, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    merged = "".join(open(f"{out}.rank{k}").read() for k in range(world))
    assert merged == single.stdout


def test_slab_reads_on_a_larger_windowed_run(tmp_path):
    """6,000 sites x 200 individuals, 30 kb window, three ranks reading their own slabs (rows + halo) from the file:
    shards concatenate to the single-GPU text (about 8.9e5 rows)."""
    from ngsld_amd import synth
    n_sites, n_ind = 6000, 200
    raw = synth.make_gl_numpy(n_sites, n_ind, seed=11, depth=6.0)
    chrs, pos = synth.make_positions(n_sites, 11, n_chr=2)
    g, p = str(tmp_path / "big.glf"), str(tmp_path / "big.pos")
    raw.tofile(g)
    synth.write_pos(p, chrs, pos)
    flags = ["--max_kb_dist", "30", "--min_maf", "0.1", "--extend_out", "--n_threads", "4"]
    single = subprocess.run([capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--verbose", "0",
                             "--pos", p] + flags, capture_output=True, text=True)
    assert single.returncode == 0, single.stderr
    out = str(tmp_path / "multi.ld")
    env = dict(os.environ, NGSLD_BENCH_ONE_DEVICE="1", PYTHONPATH=capi.REPO_DIR, NGSLD_MULTI_EXPECT="slab")
    port = 29600 + ((os.getpid() + 7) % 300)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "ngsld_amd.multi",
                        "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--verbose", "0", "--pos", p,
                        "--out", out] + flags, capture_output=True, text=True, env=env, cwd=capi.REPO_DIR, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    merged = "".join(open(f"{out}.rank{k}").read() for k in range(3))
    assert merged.count("\n") > 100_000 and merged == single.stdout


@pytest.mark.parametrize("mixed", [False, True])
def test_called_genotypes_across_ranks(mixed, tmp_path):
    """Called genotypes over three ranks.  All sites called: every rank runs the genotype-combination kernel, as the single
    process does.  Mixed (the last third of the sites keep their likelihoods): rank 0's slab is all called genotypes, the
    matrix is not -- the ranks agree on the per-individual kernels, the shards still concatenate to the single-GPU text."""
    import numpy as np
    from ngsld_amd import synth
    n_sites, n_ind = 900, 60
    raw = synth.make_gl_numpy(n_sites, n_ind, seed=31, depth=5.0)
    called = np.eye(3)[raw.argmax(axis=2)]
    called[np.random.default_rng(31).random((n_sites, n_ind)) < 0.1] = 1.0 / 3.0
    if mixed:
        called[600:] = raw[600:] / raw[600:].sum(axis=2, keepdims=True)
    chrs, pos = synth.make_positions(n_sites, 31)
    g, p = str(tmp_path / "c.glf"), str(tmp_path / "c.pos")
    called.tofile(g)
    synth.write_pos(p, chrs, pos)
    flags = ["--max_kb_dist", "4", "--extend_out", "--ignore_miss_data"]
    single = subprocess.run([capi.CLI_PATH, "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--verbose", "0",
                             "--pos", p] + flags, capture_output=True, text=True)
    assert single.returncode == 0, single.stderr
    out = str(tmp_path / "multi.ld")
    env = dict(os.environ, NGSLD_BENCH_ONE_DEVICE="1", PYTHONPATH=capi.REPO_DIR, NGSLD_MULTI_EXPECT="slab")
    port = 29600 + ((os.getpid() + 13 + int(mixed)) % 300)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "ngsld_amd.multi",
                        "--geno", g, "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--verbose", "0", "--pos", p,
                        "--out", out] + flags, capture_output=True, text=True, env=env, cwd=capi.REPO_DIR, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    merged = "".join(open(f"{out}.rank{k}").read() for k in range(3))
    assert merged.count("\n") > 10_000 and merged == single.stdout
