"""what does ONE flagged problem cost a streamed batch?  16 384 double-exponential problems of 10 000 rows, problem 1 started inside the
unrepresentable window (tests/test_gpu_census.py _window_batch), re-fit on / off.  usage: python tools/refit_streamed_probe.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import varpro_amd as vp
from varpro_amd import synth
B, m = 16384, 10000
dev = torch.device("cuda", 0)
d = synth.double_exp_batch(B, m=m, noise=1e-3)
g = d["tau_guess"].copy(); g[1, 1] = -0.0356
mdl = vp.multi_exponential_model(d["x"], g[0])
bp = vp.BatchProblem(mdl, torch.from_numpy(d["Y"]).to(dev), x=torch.from_numpy(d["x"]).to(dev))
gd = torch.from_numpy(g).to(dev)
for on in (True, False, True, False):
    bp.set_refit(on)
    bp.fit(gd, want_coefficients=False); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): a, c, rep = bp.fit(gd, want_coefficients=False)
    torch.cuda.synchronize()
    r = bp.report_to_numpy(rep)
    print("refit %-5s %.3f ms per batch; problem 1: termination %d after %d evaluations" % (on, (time.perf_counter() - t0) / 3 * 1e3, r["termination"][1], r["n_evals"][1]))
