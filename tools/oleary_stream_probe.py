"""O'Leary pair on the length-agnostic kernels at m = 5000 over batch sizes (throughput vs tail regime)."""
import numpy as np
import torch

import varpro_amd as vp

dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
m = 5000
t = np.linspace(0.0, 1.5, m)
for B in (1024, 4096, 16384, 32768):
    at = np.stack([1.0 * rng.uniform(0.9, 1.1, B), 2.5 * rng.uniform(0.9, 1.1, B), 4.0 * rng.uniform(0.9, 1.1, B)], 1)
    c = np.stack([rng.uniform(4, 8, B), rng.uniform(0.5, 2, B)], 1)
    Y = (c[:, :1] * np.exp(-at[:, 1:2] * t) * np.cos(at[:, 2:3] * t) + c[:, 1:2] * np.exp(-at[:, 0:1] * t) * np.cos(at[:, 1:2] * t))
    Y += 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    g = at * rng.uniform(0.92, 1.08, at.shape)
    mdl = (vp.SeparableModelBuilder(["a1", "a2", "a3"]).initial_parameters(g[0]).independent_variable(t)
           .function(["a2", "a3"], vp.basis.EXP_COS).partial_deriv("a2").partial_deriv("a3")
           .function(["a1", "a2"], vp.basis.EXP_COS).partial_deriv("a1").partial_deriv("a2").build())
    bp = vp.BatchProblem(mdl, torch.from_numpy(Y).to(dev), x=torch.from_numpy(t).to(dev))
    bp.set_timing(True)
    gd = torch.from_numpy(g).to(dev)
    ts = []
    for _ in range(3):
        a, cc, rep = bp.fit(gd, want_coefficients=False)
        ts.append(bp.last_kernel_ms(2))
    r = bp.report_to_numpy(rep)
    print("B=%6d fit %8.3f ms %7.3f M fits/s evals/fit %.2f max %d failed %d  -> %.1f us per evaluation and wave slot (1024 slots)" % (
        B, min(ts), B / min(ts) / 1e3, r["n_evals"].mean(), r["n_evals"].max(), (r["termination"] <= 0).sum(),
        min(ts) * 1e3 * 1024 / r["n_evals"].sum()))
    bp.close()
