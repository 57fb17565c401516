"""streamed fit kernels on LONG problems of the other shapes (block_rows_long): single / triple / four exponentials fp64, double
exponential fp32, weighted double exponential; ms per launch, evaluation totals, sum of objectives, failures.
usage: VARPRO_HIP_LIBRARY=lib.so PYTHONPATH=. python tools/stream_long_probe.py"""
import time
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
cases = (("1 exp + offset fp64", 1, [2.0], np.float64, 10000, 16384, False), ("3 exp + offset fp64", 3, [0.7, 2.0, 6.0], np.float64, 10000, 8192, False),
         ("4 exp + offset fp64", 4, [0.5, 1.5, 4.0, 9.0], np.float64, 10000, 4096, False), ("2 exp + offset fp32", 2, [1.0, 4.0], np.float32, 20000, 8192, False),
         ("2 exp + offset fp64 weighted", 2, [1.0, 4.0], np.float64, 10000, 16384, True), ("2 exp + offset fp64, B=1024 (4 waves)", 2, [1.0, 4.0], np.float64, 40000, 1024, False),
         ("5 exp + offset fp64 (configs[4]'s shape in double)", 5, [0.5, 1.5, 3.0, 6.0, 12.0], np.float64, 4096, 8192, False),
         ("5 exp + offset fp64", 5, [0.5, 1.5, 3.0, 6.0, 12.0], np.float64, 1024, 16384, False))
import sys
if len(sys.argv) > 1: cases = [c for c in cases if sys.argv[1] in c[0]]
for name, ne, taus, dt, m, B, weighted in cases:
    d = synth.multi_exp_batch(B, ne, m, taus, noise=1e-3, spread=0.1, guess_spread=0.05 if ne == 5 else 0.1, dtype=dt)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=dt)
    kw = {}
    if weighted:
        kw["weights"] = torch.from_numpy((1.0 + 0.5 * np.sin(np.arange(m) * 0.01)).astype(dt)).to(dev)
    bp = vp.BatchProblem(mdl, torch.from_numpy(d["Y"]).to(dev), x=torch.from_numpy(d["x"]).to(dev), **kw)
    t, ev, ob, nf = timed(bp, torch.from_numpy(d["tau_guess"]).to(dev))
    print("%-40s m=%d B=%d: %8.3f ms = %.3f M fits/s  evals %d  sum objective %.9e  failed %d" % (name, m, B, t, B / t / 1e3, ev, ob, nf))
    bp.close()
