"""SURVEY.md 8(e), second row: one global fit with the right-hand sides sharded over ranks and ONE all-reduce of
the reduced sums per LM evaluation.  The GPU box has a single device, so the ranks of the 2-rank case share cuda:0
and use gloo (host-bounced all-reduce); the 1-rank case drives the same code through RCCL (backend nccl)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(backend, world, tmp_path, shape="specialised"):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [str(tmp_path / ("r%d.json" % r)) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_global_fit_worker.py"), backend, str(r),
                               str(world), str(port), outs[r], shape], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    return [json.load(open(o)) for o in outs]


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1)])
def test_rhs_sharded_global_fit_matches_the_unsharded_fit(backend, world, tmp_path):
    res = _run(backend, world, tmp_path)
    r0 = res[0]
    a_full = np.array(r0["alpha_full"])
    for r in res:
        # every rank ends with the identical parameters, report and evaluation count (bit-identical LM decisions)
        assert r["alpha"] == r0["alpha"] and r["objective"] == r0["objective"] and r["n_evals"] == r0["n_evals"]
        assert all(t > 0 for t in r["termination"])
    # and they are the unsharded fit's: the summation order over the columns differs, nothing else
    assert np.abs(np.array(r0["alpha"]) - a_full).max() <= 1e-7 * np.abs(a_full).max()
    assert np.abs(np.array(r0["objective"]) - np.array(r0["objective_full"])).max() <= 1e-10 * max(r0["objective_full"])
    assert r0["max_dC_local_vs_full"] <= 1e-6
    assert np.abs(np.sort(a_full, axis=1) - np.sort(np.array(r0["alpha_true"]), axis=1)).max() <= 1e-2
    assert sum(r["count"] for r in res) == 96


def test_rhs_sharded_global_fit_on_the_generic_kernels(tmp_path):
    # a shape without a specialised MRHS kernel set (m = 2500): gen_mrhs_fit_kernel in phases {init, sums, step, results}
    # with the all-reduce of B (2 + q^2 + q) doubles between sums and step; 2 ranks (gloo) against the one-launch fit
    res = _run("gloo", 2, tmp_path, "generic")
    r0 = res[0]
    a_full = np.array(r0["alpha_full"])
    for r in res:
        assert r["alpha"] == r0["alpha"] and r["objective"] == r0["objective"] and r["n_evals"] == r0["n_evals"]
        assert all(t > 0 for t in r["termination"])
    assert np.abs(np.array(r0["alpha"]) - a_full).max() <= 1e-7 * np.abs(a_full).max()
    assert np.abs(np.array(r0["objective"]) - np.array(r0["objective_full"])).max() <= 1e-10 * max(r0["objective_full"])
    # (the summation order over the columns differs: the last, rounding-level iterations may differ in number)
    assert all(abs(a - b) <= 3 for a, b in zip(r0["n_evals"], r0["n_evals_full"]))
    assert r0["max_dC_local_vs_full"] <= 1e-6
    assert sum(r["count"] for r in res) == 24
