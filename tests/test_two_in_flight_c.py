"""Throughput mode without Python (tests/c/test_two_in_flight.c): two device-pointer handles on two HIP streams, a stream of
batches enqueued round-robin with no host wait -- every batch bit-identical to the same batch fitted alone."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "c", "test_two_in_flight")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "c"), "-s"])


def test_two_in_flight_program_builds_and_needs_a_device():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_two_batches_in_flight_from_plain_c():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout and "two in flight" in out.stdout
