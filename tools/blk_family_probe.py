"""streamed (vp_block.hpp) vs default kernel selection for the other model families: PYTHONPATH=. python tools/blk_family_probe.py"""
import numpy as np
import torch

import varpro_amd as vp
from varpro_amd import synth

dev = torch.device("cuda:0")


def run(name, mdl, Y, x, g, B):
    Yd, xd, gd = torch.from_numpy(Y).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(g).to(dev)
    for stream in (True, False):
        bp = vp.BatchProblem(mdl, Yd, x=xd, stream_rows=stream)
        bp.set_timing(True)
        ts = []
        for _ in range(3):
            a, c, rep = bp.fit(gd, want_coefficients=False)
            ts.append(bp.last_kernel_ms(2))
        r = bp.report_to_numpy(rep)
        print("%-28s %-9s %8.3f ms  %8.3f M fits/s  evals/fit %.2f failed %d" % (name, "streamed" if stream else "default", min(ts), B / min(ts) / 1e3,
                                                                                r["n_evals"].mean(), (r["termination"] <= 0).sum()))
        bp.close()


rng = np.random.default_rng(1)
for m in (2100, 3000, 4096, 6000):
    B = 16384
    x = np.linspace(0, 12.5, m)
    for nexp, base in ((1, [2.0]), (3, [0.7, 2.0, 6.0])):
        tau = np.stack([rng.uniform(0.9, 1.1, B) * t0 for t0 in base], 1)
        c = rng.uniform(5, 50, (B, nexp + 1))
        Y = sum(c[:, j:j + 1] * np.exp(-x / tau[:, j:j + 1]) for j in range(nexp)) + c[:, nexp:nexp + 1]
        Y += 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
        g = tau * rng.uniform(0.9, 1.1, tau.shape)
        run("me%d+offset m=%d" % (nexp, m), vp.multi_exponential_model(x, g[0]), Y, x, g, B)
for m in (1100, 3000, 5000):
    B = 4096
    t = np.linspace(0.0, 1.5, m)
    at = np.stack([1.0 * rng.uniform(0.9, 1.1, B), 2.5 * rng.uniform(0.9, 1.1, B), 4.0 * rng.uniform(0.9, 1.1, B)], 1)
    c = np.stack([rng.uniform(4, 8, B), rng.uniform(0.5, 2, B)], 1)
    Y = (c[:, :1] * np.exp(-at[:, 1:2] * t) * np.cos(at[:, 2:3] * t) + c[:, 1:2] * np.exp(-at[:, 0:1] * t) * np.cos(at[:, 1:2] * t))
    Y += 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    g = at * rng.uniform(0.92, 1.08, at.shape)
    mdl = (vp.SeparableModelBuilder(["a1", "a2", "a3"]).initial_parameters(g[0]).independent_variable(t)
           .function(["a2", "a3"], vp.basis.EXP_COS).partial_deriv("a2").partial_deriv("a3")
           .function(["a1", "a2"], vp.basis.EXP_COS).partial_deriv("a1").partial_deriv("a2").build())
    run("oleary m=%d" % m, mdl, Y, t, g, B)
