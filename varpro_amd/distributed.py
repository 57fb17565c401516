"""Multi-GPU layer: independent problems shard across ranks with one scalar reduction
(ShardedFit); one global fit shards its right-hand sides with one small all-reduce per LM evaluation
(ShardedGlobalFit).

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU
tests).  Rank r owns a contiguous block of the batch; every rank runs the full device-resident LM fit on
its block with no data-path collective.  The ONLY exchange is a sum all-reduce of the 4 doubles
{sum 1/2||r||^2, #successful, #failed, sum evaluations} that ``vp_summary`` returns per rank.
The reference has no distributed code at all (it is single-threaded); this layer is new.
"""
import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def shard_range(total, rank, world):
    """contiguous block [first, first+count) of rank `rank`: ceil(total/world) problems per rank, last ranks
    may get fewer (or none)"""
    per = (int(total) + int(world) - 1) // int(world)
    first = min(int(rank) * per, int(total))
    count = max(0, min(per, int(total) - first))
    return first, count


def allreduce_summary(local4, device=None):
    """sum the per-rank {sum cost, #ok, #failed, sum evals} over all ranks.  Works on whatever backend the
    default process group uses: CUDA tensor for nccl/RCCL, CPU tensor for gloo; identity without a group."""
    v = np.asarray(local4, dtype=np.float64).reshape(4)
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return v.copy()
    backend = dist.get_backend()
    t = torch.from_numpy(v.copy())
    if backend == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


class ShardedFit:
    """fit this rank's shard of a batch of independent problems and reduce the scalar summary"""

    def __init__(self, model, Y_shard, x=None, weights=None, epsilon=None):
        from .batch import BatchProblem
        self.batch = BatchProblem(model, Y_shard, x=x, weights=weights, epsilon=epsilon)

    def fit(self, alpha0_shard, solver=None):
        alpha, C, rep = self.batch.fit(alpha0_shard, solver=solver)
        local = self.batch.summary()
        return alpha, C, rep, local, allreduce_summary(local)

    def close(self):
        self.batch.close()


class ShardedGlobalFit:
    """ONE global fit (shared nonlinear parameters, S right-hand sides) with the COLUMNS sharded over ranks
    (second row).  Every rank passes its block Y[:, first:first+count, :] as a device tensor; the
    path has a real exchange step here -- per LM evaluation one sum all-reduce of B*(1+n*n+p) doubles (RCCL
    over xGMI with backend "nccl") -- after which all ranks take the identical LM step."""

    def __init__(self, model, Y_shard, global_rhs_count, x=None, weights=None, epsilon=None, group=None):
        from .batch import BatchProblem
        self.batch = BatchProblem(model, Y_shard, x=x, weights=weights, epsilon=epsilon)
        self.batch.set_rhs_allreduce(global_rhs_count, group=group)

    def fit(self, alpha0, solver=None):
        """-> (alpha identical on every rank, C of the LOCAL columns, report with the GLOBAL objective)"""
        return self.batch.fit(alpha0, solver=solver)

    def close(self):
        self.batch.close()
