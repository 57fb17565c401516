import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth, _lib
m, B = 128, 262144
d = synth.double_exp_batch(B, m=m, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda", 0)
Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=x); bp.set_timing(True); bp.set_fit_kernel("slots")
ts = []
for _ in range(6):
    a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
print("m=128 slots %.3f ms %.2f Mfits/s" % (min(ts), B / min(ts) / 1e3))
