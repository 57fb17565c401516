"""N > 1 path on CPU: world_size-2 gloo process group.  The sharding and the scalar reduction are the whole
distributed path (problems are independent); the per-rank compute is stood in for by the CPU oracle here
(tests only) -- on the GPU box each rank runs vp_fit on its shard instead (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from varpro_amd import distributed as vd  # noqa: E402
from varpro_amd import synth  # noqa: E402

TOTAL, M = 37, 64  # deliberately not divisible by the world size


def test_shard_ranges_partition_the_batch():
    for total in (0, 1, 7, 37, 4096, 524288):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, count = vd.shard_range(total, r, world)
                seen.extend(range(first, first + count))
            assert seen == list(range(total))
    assert vd.shard_range(524288, 3, 8) == (196608, 65536)  # BASELINE configs[3]: 65536 per GPU


def test_shard_inputs_are_slices_of_the_global_problem_set():
    full = synth.double_exp_batch(TOTAL, m=M)
    for world in (2, 3):
        for r in range(world):
            first, count = vd.shard_range(TOTAL, r, world)
            part = synth.double_exp_batch(count, m=M, first_problem=first)
            assert np.array_equal(part["Y"], full["Y"][first:first + count])
            assert np.array_equal(part["tau_guess"], full["tau_guess"][first:first + count])


def _local_summary(first, count):
    from models import double_exp_builder_model
    from oracle import oracle as O
    d = synth.double_exp_batch(count, m=M, first_problem=first)
    mdl = double_exp_builder_model(d["x"], [1.0, 4.0])
    _a, _c, rep, _s = O.fit_batch(mdl, d["x"], d["Y"], d["tau_guess"])
    ok = rep["termination"] > 0
    return np.array([rep["objective"].sum(), ok.sum(), (~ok).sum(), rep["n_evals"].sum()], dtype=np.float64)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count = vd.shard_range(TOTAL, rank, world)
        local = _local_summary(first, count)
        glob = vd.allreduce_summary(local)
        # max-over-ranks timing as bench.py does it
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        q.put((rank, local, glob, float(t.item())))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_two_rank_gloo_reduction_matches_unsharded():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda t: t[0])
    unsharded = _local_summary(0, TOTAL)
    total = sum(r[1] for r in results)
    for _rank, _local, glob, tmax in results:
        assert np.array_equal(glob[1:], unsharded[1:])          # counts are exact
        assert abs(glob[0] - unsharded[0]) <= 1e-12 * unsharded[0]  # the cost sum up to summation order
        assert np.allclose(glob, total, rtol=1e-15)
        assert tmax == float(world)
    assert results[0][1][1] + results[1][1][1] == unsharded[1]


def test_allreduce_summary_is_identity_without_a_process_group():
    v = np.array([1.5, 2.0, 0.0, 20.0])
    assert np.array_equal(vd.allreduce_summary(v), v)
