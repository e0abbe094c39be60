"""Host hardening: the host-side tests and the CLI's error paths under AddressSanitizer + UBSan (tests/run_asan.sh builds
the sanitized host objects with `make -C ngsld_amd/csrc asan`; device code is not instrumented)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_code_is_clean_under_asan_and_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc's libasan is not installed")
    r = subprocess.run([os.path.join(HERE, "run_asan.sh"), "-x"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "asan run clean" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
