"""streamed fit kernels (vp_block.hpp) of a library build: the bench's m = 10 000 / 100 000 double-exponential legs and the
O'Leary exp*cos leg (m = 5000), ms per launch + evaluation totals + sum of objectives.
usage: VARPRO_HIP_LIBRARY=lib.so PYTHONPATH=. python tools/stream_ab_probe.py"""
import sys, time
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth
dev = torch.device("cuda", 0)
def timed(bp, g, n=3):
    bp.fit(g, want_coefficients=False); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); a, c, rep = bp.fit(g, want_coefficients=False); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    r = bp.report_to_numpy(rep)
    return min(ts) * 1e3, int(r["n_evals"].sum()), float(np.nansum(r["objective"])), int((r["termination"] <= 0).sum())
FIRST = len(sys.argv) > 1 and sys.argv[1] == "first"  # only the m = 10 000 leg (tools/pmc_blk.sh)
for ms, Bs in ((10000, 16384), (100000, 2048), (3000, 32768))[:1 if FIRST else 3]:
    ds = synth.double_exp_batch(Bs, m=ms, noise=1e-3)
    mdl = vp.multi_exponential_model(ds["x"], ds["tau_guess"][0])
    bp = vp.BatchProblem(mdl, torch.from_numpy(ds["Y"]).to(dev), x=torch.from_numpy(ds["x"]).to(dev), flags=getattr(vp, "FLAG_STREAM_ROWS", 0) if ms <= 4096 else 0) if False else vp.BatchProblem(mdl, torch.from_numpy(ds["Y"]).to(dev), x=torch.from_numpy(ds["x"]).to(dev))
    t, ev, ob, nf = timed(bp, torch.from_numpy(ds["tau_guess"]).to(dev))
    print("double-exp m=%d B=%d: %.3f ms = %.3f M fits/s  evals %d  sum objective %.9e  failed %d" % (ms, Bs, t, Bs / t / 1e3, ev, ob, nf))
    bp.close()
if FIRST: sys.exit(0)
Bg, mg = 4096, 5000
tg = np.linspace(0.0, 1.5, mg)
rg = synth.SplitMix64(np.uint64(0x5EED3000) + np.arange(Bg, dtype=np.uint64))
at = np.stack([1.0 * (1 + 0.1 * rg.uniform(-1, 1)), 2.5 * (1 + 0.1 * rg.uniform(-1, 1)), 4.0 * (1 + 0.1 * rg.uniform(-1, 1))], 1)
cg = np.stack([rg.uniform(4.0, 8.0), rg.uniform(0.5, 2.0)], 1)
Yg = (cg[:, :1] * np.exp(-at[:, 1:2] * tg[None]) * np.cos(at[:, 2:3] * tg[None]) + cg[:, 1:2] * np.exp(-at[:, 0:1] * tg[None]) * np.cos(at[:, 1:2] * tg[None]))
Yg = Yg + 1e-3 * np.abs(Yg).max(1, keepdims=True) * rg.normal(mg)
gg0 = at * np.stack([1 + 0.1 * rg.uniform(-1, 1) for _ in range(3)], 1)
mdl = (vp.SeparableModelBuilder(["alpha1", "alpha2", "alpha3"]).initial_parameters(gg0[0]).independent_variable(tg)
       .function(["alpha2", "alpha3"], vp.basis.EXP_COS).partial_deriv("alpha2").partial_deriv("alpha3")
       .function(["alpha1", "alpha2"], vp.basis.EXP_COS).partial_deriv("alpha1").partial_deriv("alpha2").build())
bp = vp.BatchProblem(mdl, torch.from_numpy(Yg).to(dev), x=torch.from_numpy(tg).to(dev))
t, ev, ob, nf = timed(bp, torch.from_numpy(gg0).to(dev))
print("O'Leary m=%d B=%d: %.3f ms = %.3f M fits/s  evals %d  sum objective %.9e  failed %d" % (mg, Bg, t, Bg / t / 1e3, ev, ob, nf))
