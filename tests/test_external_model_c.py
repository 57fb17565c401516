"""A model outside the descriptor language through the C ABI (tests/c/test_external_model.c): host C callbacks evaluate a
Gaussian + Lorentzian + offset model, an EXTERNAL lmder driver (the oracle's `vpo_lm_minimize`, compiled into the test
program only) fits it through vp_batch_create_external / vp_set_params_with_basis / vp_residuals /
vp_jacobian_with_derivatives -- the reference's trait boundary for ANY `SeparableNonlinearModel`
(/root/reference/src/model/mod.rs:239-363, 441-512; /root/reference/src/solvers/levmar/mod.rs:42-73, 101-201) -- and must
match the oracle given the same callbacks: c, r, J to 1e-10 at the initial point, the same trajectory and minimum."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "c", "test_external_model")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "..", "oracle"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(HERE, "c"), "-s"])


def test_external_model_program_builds_and_needs_a_device():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_external_lmder_fits_a_gauss_lorentz_model_through_the_c_abi():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout and "external over the C ABI" in out.stdout
    for case in ("S=1 Gauss+Lorentz+offset", "S=1 weighted", "S=3 (branch S<=q)", "S=6 (branch S>q)", "non-finite basis"):
        assert case in out.stdout
