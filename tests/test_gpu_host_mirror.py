"""Runs the C++ host-mirror test program (tests/cpp/test_host_mirror.cc): the reference's own
TestLinearSearch / TestFilter / IVF TestSimple, written against zvec_amd/csrc/host/hip_index.h."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror():
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all tests passed" in r.stdout
