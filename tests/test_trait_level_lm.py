"""The trait-level drop-in, end to end (tests/c/test_trait_lm.c): an EXTERNAL lmder driver -- the oracle's
`vpo_lm_minimize`, compiled into the test program only -- calls vp_set_params -> vp_residuals -> vp_jacobian through the
C ABI exactly as `LevenbergMarquardt::minimize` calls the `LeastSquaresProblem` trait of a `SeparableProblem`
(/root/reference/src/solvers/levmar/mod.rs:22-202, 238-254), and must reproduce vp_fit and the oracle: the executable
stand-in for the Rust shim that cannot be compiled in this image."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "c", "test_trait_lm")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "..", "oracle"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(HERE, "c"), "-s"])


def test_trait_lm_program_builds_and_needs_a_device():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_external_lmder_over_the_c_abi_matches_vp_fit_and_the_oracle():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout and "external:" in out.stdout
    for case in ("S=1 configs[0]", "S=2 (branch S<=q)", "S=3 (branch S>q)", "weighted", "failing set_params"):
        assert case in out.stdout
