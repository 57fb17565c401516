"""configs[2]'s trait-level evaluation (R and J out, mrhs_coop_out_kernel): min / median ms of 25 launches per library build.
usage: VARPRO_HIP_LIBRARY=lib.so PYTHONPATH=. python tools/mrhs_trait_probe.py"""
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib
dev = torch.device("cuda", 0)
S2, m2 = 16384, 2048
d2 = synth.mrhs_triple_exp(S=S2, m=m2)
mdl2 = vp.multi_exponential_model(d2["x"], d2["tau_guess"], offset=True)
bp2 = vp.BatchProblem(mdl2, torch.from_numpy(d2["Y"][None]).to(dev), x=torch.from_numpy(d2["x"]).to(dev))
g2 = torch.from_numpy(d2["tau_guess"][None]).to(dev)
bp2.set_timing(True)
ts = []
for _ in range(25):
    bp2.evaluate(g2, want_residuals=True, want_jacobian=True); ts.append(bp2.last_kernel_ms(_lib.VP_KERNEL_EVALUATE))
ts = sorted(ts[3:])
by = 8 * m2 * S2 * 5
print("trait evaluation min %.4f median %.4f ms -> %.3f / %.3f of 8 TB/s" % (ts[0], ts[len(ts) // 2], by / ts[0] / 8e9, by / ts[len(ts) // 2] / 8e9))
