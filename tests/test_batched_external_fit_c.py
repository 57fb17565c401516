"""A batch of models outside the descriptor language FITTED through the C ABI by reverse communication
(tests/c/test_batched_external_fit.c): host C callbacks evaluate a Gaussian + Lorentzian + offset model for 64 problems,
the device keeps one LM driver per problem (vp_fit_begin / vp_fit_step_with_basis / vp_fit_end) -- the batch form of the
reference's `LevMarSolver::fit` over ANY `SeparableNonlinearModel` (/root/reference/src/solvers/levmar/mod.rs:238-254,
/root/reference/src/model/mod.rs:239-363) -- and every problem must end where the oracle ends given the same callbacks."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "c", "test_batched_external_fit")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "..", "oracle"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(HERE, "c"), "-s"])


def test_batched_external_fit_program_builds_and_needs_a_device():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_batched_fit_of_a_gauss_lorentz_model_through_the_c_abi():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout and "batched over the C ABI" in out.stdout
    for case in ("unit weights", "per-row weights"):
        assert case in out.stdout
