"""run-time-descriptor models at m = 1024 (16 rows per lane): fit throughput.  PYTHONPATH=. python tools/rt_wave_probe.py"""
import numpy as np
import torch

import varpro_amd as vp

dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
B, m = 32768, 1024
# the builder-made double exponential (RtModel<3,2,2>)
x = np.linspace(0, 12.5, m)
tau = np.stack([rng.uniform(0.9, 1.1, B) * 1.0, rng.uniform(0.9, 1.1, B) * 3.0], 1)
c = rng.uniform(5, 50, (B, 3))
Y = c[:, :1] * np.exp(-x / tau[:, :1]) + c[:, 1:2] * np.exp(-x / tau[:, 1:2]) + c[:, 2:3]
Y += 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
g = tau * rng.uniform(0.9, 1.1, tau.shape)
mdl = (vp.SeparableModelBuilder(["t1", "t2"]).initial_parameters(g[0]).independent_variable(x)
       .function(["t2"], vp.basis.EXP_DECAY).partial_deriv("t2").function(["t1"], vp.basis.EXP_DECAY).partial_deriv("t1")
       .invariant_function(vp.basis.CONST).build())
t = np.linspace(0.0, 1.5, m)
at = np.stack([1.0 * rng.uniform(0.9, 1.1, B), 2.5 * rng.uniform(0.9, 1.1, B), 4.0 * rng.uniform(0.9, 1.1, B)], 1)
c2 = np.stack([rng.uniform(4, 8, B), rng.uniform(0.5, 2, B)], 1)
Y2 = (c2[:, :1] * np.exp(-at[:, 1:2] * t) * np.cos(at[:, 2:3] * t) + c2[:, 1:2] * np.exp(-at[:, 0:1] * t) * np.cos(at[:, 1:2] * t))
Y2 += 1e-3 * np.abs(Y2).max(1, keepdims=True) * rng.standard_normal(Y2.shape)
g2 = at * rng.uniform(0.92, 1.08, at.shape)
mdl2 = (vp.SeparableModelBuilder(["a1", "a2", "a3"]).initial_parameters(g2[0]).independent_variable(t)
        .function(["a2", "a3"], vp.basis.EXP_COS).partial_deriv("a2").partial_deriv("a3")
        .function(["a1", "a2"], vp.basis.EXP_COS).partial_deriv("a1").partial_deriv("a2").build())
for name, md, YY, xx, gg in (("rt(3,2,2) double exp", mdl, Y, x, g), ("rt(2,3,4) oleary", mdl2, Y2, t, g2)):
    for stream in (False, True):
        bp = vp.BatchProblem(md, torch.from_numpy(YY).to(dev), x=torch.from_numpy(xx).to(dev), stream_rows=stream)
        bp.set_timing(True)
        gd = torch.from_numpy(gg).to(dev)
        ts, te = [], []
        for _ in range(3):
            a, cc, rep = bp.fit(gd, want_coefficients=False)
            ts.append(bp.last_kernel_ms(2))
            bp.evaluate(gd)
            te.append(bp.last_kernel_ms(0))
        r = bp.report_to_numpy(rep)
        print("%-24s %-9s fit %8.3f ms %7.3f M fits/s evals/fit %.2f failed %d | evaluate(r,J) %7.3f ms" % (
            name, "streamed" if stream else "default", min(ts), B / min(ts) / 1e3, r["n_evals"].mean(), (r["termination"] <= 0).sum(), min(te)))
        bp.close()
# a larger run-time shape: n = 4, q = 4 (two exp*cos pairs sharing nothing) at m = 1000
t = np.linspace(0.0, 1.5, 1000)
B4 = 8192
a4 = np.stack([1.0 * rng.uniform(0.9, 1.1, B4), 3.0 * rng.uniform(0.9, 1.1, B4), 2.0 * rng.uniform(0.9, 1.1, B4), 6.0 * rng.uniform(0.9, 1.1, B4)], 1)
c4 = rng.uniform(1, 5, (B4, 4))
Y4 = (c4[:, :1] * np.exp(-a4[:, 0:1] * t) + c4[:, 1:2] * np.exp(-a4[:, 1:2] * t) + c4[:, 2:3] * np.exp(-a4[:, 2:3] * t) * 0 + c4[:, 3:4] * np.exp(-a4[:, 3:4] * t))
Y4 = c4[:, :1] * np.exp(-a4[:, 0:1] * t) + c4[:, 1:2] * np.exp(-a4[:, 1:2] * t) + c4[:, 2:3] * np.exp(-a4[:, 2:3] * t) + c4[:, 3:4] * np.exp(-a4[:, 3:4] * t)
Y4 += 1e-3 * np.abs(Y4).max(1, keepdims=True) * rng.standard_normal(Y4.shape)
g4 = a4 * rng.uniform(0.95, 1.05, a4.shape)
b4 = vp.SeparableModelBuilder(["r1", "r2", "r3", "r4"]).initial_parameters(g4[0]).independent_variable(t)
for nme in ("r1", "r2", "r3", "r4"):
    b4 = b4.function([nme], vp.basis.EXP_RATE).partial_deriv(nme)
mdl4 = b4.build()
for stream in (False, True):
    bp = vp.BatchProblem(mdl4, torch.from_numpy(Y4).to(dev), x=torch.from_numpy(t).to(dev), stream_rows=stream)
    bp.set_timing(True)
    gd = torch.from_numpy(g4).to(dev)
    ts = []
    for _ in range(3):
        a, cc, rep = bp.fit(gd, want_coefficients=False)
        ts.append(bp.last_kernel_ms(2))
    r = bp.report_to_numpy(rep)
    print("%-24s %-9s fit %8.3f ms %7.3f M fits/s evals/fit %.2f failed %d" % ("rt(4,4,4) four rates", "streamed" if stream else "default", min(ts), B4 / min(ts) / 1e3,
                                                                              r["n_evals"].mean(), (r["termination"] <= 0).sum()))
    bp.close()
