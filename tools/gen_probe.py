"""the generic fallback kernels on a descriptor shape WITHOUT a specialised kernel set: six exponentials + offset, fp64 (n = 7, q = 6,
p = 6).  usage: python tools/gen_probe.py [m] [B]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import varpro_amd as vp
from varpro_amd import synth
m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
taus = [0.3, 0.8, 1.8, 4.0, 9.0, 20.0]
d = synth.multi_exp_batch(B, 6, m, taus, noise=1e-3, spread=0.05, guess_spread=0.02)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda", 0)
bp = vp.BatchProblem(mdl, torch.from_numpy(d["Y"]).to(dev), x=torch.from_numpy(d["x"]).to(dev))
g = torch.from_numpy(d["tau_guess"]).to(dev)
def ev(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out
tf, (a, c, rep) = ev(lambda: bp.fit(g, want_coefficients=False))
r = bp.report_to_numpy(rep)
te, _ = ev(lambda: bp.evaluate(g, want_residuals=True, want_jacobian=True))
print("six exponentials + offset, m=%d B=%d: fit %.2f ms (%.3f M fits/s, %.1f evals/fit, %.1f us per evaluation and workgroup-slot), failed %.3f; evaluate r+J %.3f ms"
      % (m, B, tf, B / tf / 1e3, r["n_evals"].mean(), tf * 1e3 * min(B, 1024) / r["n_evals"].sum(), (r["termination"] <= 0).mean(), te))
