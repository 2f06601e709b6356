"""Runs the C++ parity harness of the host-side mirror (include/jf_pcs.hpp), which follows the
reference's test_gpu_e2e protocol (pcs/src/univariate_kzg/mod.rs:1359-1414) through the C ABI."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror():
    exe = os.path.join(ROOT, "tests", "cpp", "test_pcs")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all ok" in r.stdout
