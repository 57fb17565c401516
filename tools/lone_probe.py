"""B problems (default 256: one wavefront per SIMD at most -> every wave runs alone), fit_kernel; for tools/pmc_lone.sh"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = synth.double_exp_batch(B, m=1024, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda", 0)
bp = vp.BatchProblem(mdl, torch.from_numpy(d["Y"]).to(dev), x=torch.from_numpy(d["x"]).to(dev))
bp.set_fit_kernel("wave"); bp.set_timing(True)
g = torch.from_numpy(d["tau_guess"]).to(dev)
ts = []
for _ in range(6):
    a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
r = bp.report_to_numpy(rep)
print("B %d fit_kernel min %.3f ms, evals sum %d max %d -> %.2f us per evaluation of the longest fit" % (B, min(ts), r["n_evals"].sum(), r["n_evals"].max(), min(ts) * 1e3 / r["n_evals"].max()))
