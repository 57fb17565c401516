"""bench.py under torch.distributed.run with TWO ranks on the box's one GPU (test hooks VP_BENCH_SINGLE_DEVICE /
VP_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device).  Exercises exactly the control flow the driver's
multi-GPU run uses -- rendezvous, per-rank shard of the global synthetic problem set, per-step all-reduce of the four
scalars, max-over-ranks timing, one JSON line from rank 0 -- before an 8-GPU node exists."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import varpro_amd as vp
from varpro_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_two_ranks_one_json_line_and_sharded_totals():
    B = 4096
    env = dict(os.environ, VP_BENCH_SINGLE_DEVICE="1", VP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", str(B), "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline", "--no-side-configs"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["unit"] == "fits/s"
    cfg = out["config"]
    assert cfg["world_size"] == 2 and cfg["collective_backend"] == "gloo" and cfg["batch_per_gpu"] == B
    assert cfg["fits_successful"] + cfg["fits_failed"] == 2 * B
    assert out["value"] > 0 and abs(out["value"] - 2 * B * 3 / (out["ms_per_step"] * 3e-3)) <= 1e-6 * out["value"]
    # the two ranks fitted the two halves of ONE global problem set: their all-reduced totals equal the totals of a
    # single handle holding problems 0 .. 2B-1 (same generator, first_problem = rank * B on each rank)
    d = synth.double_exp_batch(2 * B, m=1024, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    _a, _c, rep = bp.fit(d["tau_guess"])
    bp.close()
    ok = rep["termination"] > 0
    assert cfg["fits_successful"] == ok.sum() and cfg["fits_failed"] == (~ok).sum()
    assert abs(cfg["mean_evaluations_per_fit"] * 2 * B - rep["n_evals"].sum()) < 0.5
    assert abs(cfg["sum_cost"] - np.nansum(rep["objective"])) <= 1e-9 * np.nansum(rep["objective"])


def test_bench_eight_ranks_dry_run_on_one_device():
    # the driver's `--gpus 8` launch line, 8 ranks sharing the box's one GPU over gloo, B = 512 per rank: the 8-way
    # rendezvous, the 8 contiguous shards of one 4096-problem set and the 8-way all-reduced totals
    B, G = 512, 8
    env = dict(os.environ, VP_BENCH_SINGLE_DEVICE="1", VP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(G), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(G), "--batch", str(B), "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-side-configs"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == G and cfg["world_size"] == G and cfg["batch_per_gpu"] == B and out["scaling"] == "weak"
    assert cfg["fits_successful"] + cfg["fits_failed"] == G * B
    d = synth.double_exp_batch(G * B, m=1024, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    _a, _c, rep = bp.fit(d["tau_guess"])
    bp.close()
    assert cfg["fits_successful"] == (rep["termination"] > 0).sum()
    assert abs(cfg["mean_evaluations_per_fit"] * G * B - rep["n_evals"].sum()) < 0.5
