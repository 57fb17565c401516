"""Worker of tests/test_gpu_sharded_global_fit.py: one rank of a global fit whose right-hand sides are sharded.
usage: python sharded_global_fit_worker.py <backend> <rank> <world> <port> <out.json>
Every rank uses cuda:0 (the GPU box has one device; with backend gloo the all-reduce is bounced through the host,
with backend nccl and world 1 it goes through RCCL)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import varpro_amd as vp  # noqa: E402
from varpro_amd.distributed import ShardedGlobalFit, shard_range  # noqa: E402

backend, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
shape = sys.argv[6] if len(sys.argv) > 6 else "specialised"
torch.cuda.set_device(0)
dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
dev = torch.device("cuda", 0)

rng = np.random.default_rng(1234)  # the same problem on every rank
# "generic": m = 2500 rows, beyond every specialised kernel set of this model -> the generic kernels (vp_generic.hpp),
# whose global fit runs in phases around the all-reduce
m, S, B = (512, 96, 2) if shape == "specialised" else (2500, 24, 2)
x = np.linspace(0.0, 12.5, m)
tau_true = np.array([[1.0, 3.0, 7.0], [0.8, 2.5, 9.0]])
Cm = rng.uniform(1, 100, (B, S, 4))
Y = np.zeros((B, S, m))
for b in range(B):
    Phi = np.stack([np.exp(-x / t) for t in tau_true[b]] + [np.ones(m)], axis=0)  # (4, m)
    Y[b] = Cm[b] @ Phi
Y += 1e-4 * np.abs(Y).max() * rng.standard_normal(Y.shape)
guess = tau_true * np.array([1.3, 1.2, 1.25])
mdl = vp.multi_exponential_model(x, guess[0], offset=True)

first, count = shard_range(S, rank, world)
Ys = torch.from_numpy(np.ascontiguousarray(Y[:, first:first + count, :])).to(dev)
xt = torch.from_numpy(x).to(dev)
g = torch.from_numpy(guess.copy()).to(dev)
sg = ShardedGlobalFit(mdl, Ys, global_rhs_count=S, x=xt)
a, C, rep = sg.fit(g)
a = a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
C = C.cpu().numpy() if hasattr(C, "cpu") else np.asarray(C)
rep = sg.batch.report_to_numpy(rep)
res = {"rank": rank, "alpha": a.tolist(), "objective": rep["objective"].tolist(), "n_evals": rep["n_evals"].tolist(),
       "termination": rep["termination"].tolist(), "first": first, "count": count}
sg.close()

if rank == 0:  # the unsharded fit of the same problem on one handle
    full = vp.BatchProblem(mdl, torch.from_numpy(Y).to(dev), x=xt)
    af, Cf, rf = full.fit(g)
    rf = full.report_to_numpy(rf)
    af = af.cpu().numpy()
    Cf = Cf.cpu().numpy()
    res["alpha_full"] = af.tolist()
    res["objective_full"] = rf["objective"].tolist()
    res["n_evals_full"] = rf["n_evals"].tolist()
    res["max_dC_local_vs_full"] = float(np.abs(C - Cf[:, first:first + count, :]).max() / np.abs(Cf).max())
    res["alpha_true"] = tau_true.tolist()
    full.close()
json.dump(res, open(out, "w"))
dist.barrier()
dist.destroy_process_group()
