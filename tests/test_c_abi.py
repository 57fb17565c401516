"""The C ABI used from plain C99 (tests/c/test_c_abi.c): header validity, linking, loud failure without a device,
and the smallest end-to-end fit with one."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "c", "test_c_abi")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "c"), "-s"])


def test_c_abi_compiles_as_c99_and_fails_loudly_without_a_device():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout


@pytest.mark.gpu
def test_c_abi_fit_from_plain_c():
    _build()
    out = subprocess.run([EXE, "expect_gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout and "fit:" in out.stdout
