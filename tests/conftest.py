"""pytest configuration: the `gpu` marker and shared helpers.

`-m "not gpu"`: oracle vs golden vectors / reference build, host logic, C-ABI symbol checks, gloo sharding.
`-m gpu`     : parity tests proper -- the HIP path, called through the C-ABI, against the oracle.
"""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


# torch carries its own HIP runtime; libngsld.so links the one under /opt/rocm.  Whichever is loaded first serves
# both, and torch does not find the device through the other one, so tests that use both fix the order here.
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _have_gpu() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd missing)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def engine():
    from ngsld_amd import capi
    eng = capi.Engine(0)
    yield eng
    eng.close()


def pytest_terminal_summary(terminalreporter):
    """How many pairs the parity checks compared, and how many of them needed more than the 1e-9 bar (expected: none --
    the ill-conditioned ones are replayed in the reference's operation order)."""
    try:
        from util import REPORT
    except ImportError:
        from tests.util import REPORT
    if REPORT["pairs"]:
        terminalreporter.write_line(
            f"parity: {REPORT['pairs']} pairs compared with the oracle, {REPORT['over_tol']} beyond 1e-9, "
            f"{REPORT['degenerate']} degenerate pairs held to bit equality, largest difference {REPORT['max_diff']:.3e}")
