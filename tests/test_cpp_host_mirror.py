"""The C++ host-side mirror (varpro_amd/cpp/varpro.hpp) of the reference's builder/problem/solver surface,
exercised by a compiled C++ program written like the reference's own integration tests (tests/cpp/)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "cpp", "test_host_mirror")


def _ensure_built():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp"), "-s"])


def test_cpp_builder_errors_and_no_cpu_fallback():
    _ensure_built()
    out = subprocess.run([EXE, "errors"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout


@pytest.mark.gpu
def test_cpp_reference_integration_tests_on_gpu():
    _ensure_built()
    out = subprocess.run([EXE, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout
