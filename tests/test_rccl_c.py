"""The collective path without Python (tests/c/test_rccl_summary.c): a plain C host links librccl, makes a 1-rank
communicator and (1) all-reduces vp_summary_device's 4 doubles on the handle's stream / calls vp_reduce_cost, (2) runs a
right-hand-side-sharded global fit whose per-evaluation exchange (vp_set_rhs_allreduce) is ncclAllReduce -- both compared
with the unsharded numbers.  SURVEY.md 8(b), 8(e); BASELINE.json north_star ("RCCL over xGMI only for the scalar LM cost
reduction")."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "c", "test_rccl_summary")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "c"), "-s"])


def test_rccl_program_builds_and_needs_a_device():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_rccl_allreduce_from_plain_c():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout and "RCCL-reduced" in out.stdout and "all-reduces" in out.stdout
