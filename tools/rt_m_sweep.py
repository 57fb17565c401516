"""run-time-descriptor model (the O'Leary exp*cos example, n = 2, q = 3, p = 4) across problem lengths"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth, _lib
dev = torch.device("cuda", 0)
B = 32768
for m in (64, 128, 129, 200, 256, 512, 1000, 1024, 1100):
    tg = np.linspace(0.0, 1.5, m)
    rg = synth.SplitMix64(np.uint64(0x5EED3000) + np.arange(B, dtype=np.uint64))
    at = np.stack([1.0 * (1 + 0.1 * rg.uniform(-1, 1)), 2.5 * (1 + 0.1 * rg.uniform(-1, 1)), 4.0 * (1 + 0.1 * rg.uniform(-1, 1))], 1)
    cg = np.stack([rg.uniform(4.0, 8.0), rg.uniform(0.5, 2.0)], 1)
    Y = (cg[:, :1] * np.exp(-at[:, 1:2] * tg[None]) * np.cos(at[:, 2:3] * tg[None]) + cg[:, 1:2] * np.exp(-at[:, 0:1] * tg[None]) * np.cos(at[:, 1:2] * tg[None]))
    Y = Y + 1e-3 * np.abs(Y).max(1, keepdims=True) * rg.normal(m)
    g0 = at * np.stack([1 + 0.1 * rg.uniform(-1, 1) for _ in range(3)], 1)
    mdl = (vp.SeparableModelBuilder(["alpha1", "alpha2", "alpha3"]).initial_parameters(g0[0]).independent_variable(tg)
           .function(["alpha2", "alpha3"], vp.basis.EXP_COS).partial_deriv("alpha2").partial_deriv("alpha3")
           .function(["alpha1", "alpha2"], vp.basis.EXP_COS).partial_deriv("alpha1").partial_deriv("alpha2").build())
    bp = vp.BatchProblem(mdl, torch.from_numpy(Y).to(dev), x=torch.from_numpy(tg).to(dev)); bp.set_timing(True)
    g = torch.from_numpy(g0).to(dev)
    ts = []
    for _ in range(4):
        a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
    r = bp.report_to_numpy(rep)
    print("m %5d B %d fit %.3f ms %.2f Mfits/s evals/fit %.1f ok %.3f" % (m, B, min(ts), B / min(ts) / 1e3, r["n_evals"].mean(), (r["termination"] > 0).mean()), flush=True)
    bp.close()
