"""The host library's C++ self-test under AddressSanitizer + UBSan
(tools/host_sanitize.sh): codec bounds, crypto wrappers, Philox ranges."""

import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_selftest_is_clean_under_asan_and_ubsan(tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path))
    proc = subprocess.run(["bash", os.path.join(ROOT, "tools", "host_sanitize.sh")],
                          capture_output=True, text=True, timeout=600, env=env)
    if proc.returncode != 0 and "cannot find -lasan" in proc.stderr + proc.stdout:
        pytest.skip("libasan is not installed")
    assert proc.returncode == 0, (proc.stdout + proc.stderr)[-3000:]
    assert "host selftest ok" in proc.stdout
