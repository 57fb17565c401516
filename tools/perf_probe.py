"""Kernel-level timing probe (HIP events inside the library): evaluate modes, basis, fit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d = synth.double_exp_batch(B, m=1024, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda", 0)
Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=x)
bp.set_timing(True)
def t_eval(wr, wj, n=5):
    ts = []
    for _ in range(n):
        bp.evaluate(g, want_residuals=wr, want_jacobian=wj)
        ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_EVALUATE))
    return min(ts)
print("B", B)
for name, wr, wj in (("eval mode0 (c,cost)", False, False), ("eval mode1 (+r)", True, False), ("eval mode2 (+r,J)", True, True)):
    ms = t_eval(wr, wj)
    print("%-22s %8.3f ms  %8.1f Mevals/s" % (name, ms, B / ms / 1e3))
ts = []
for _ in range(5):
    bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
rep = None
a, c, rep = bp.fit(g)
r = bp.report_to_numpy(rep)
print("fit                    %8.3f ms  %8.2f Mfits/s  evals/fit %.2f max %d  p99 %d" % (min(ts), B / min(ts) / 1e3, r["n_evals"].mean(), r["n_evals"].max(), np.percentile(r["n_evals"], 99)))
print("   => %.1f Mevals/s inside fit" % (r["n_evals"].sum() / min(ts) / 1e3))
import collections
print("terminations", collections.Counter(r["termination"].tolist()))
