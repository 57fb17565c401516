"""basis kernel (Phi/dPhi streaming stores) timing: python tools/basis_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib
B, m = 65536, 1024
d = synth.double_exp_batch(B, m=m, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda", 0)
Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=x); bp.set_timing(True)
ph = torch.empty((B, 2, m), dtype=torch.float64, device=dev); dp = torch.empty((B, 2, m), dtype=torch.float64, device=dev)
ts = []
for _ in range(25):
    bp.basis(g, skip_invariant=True, out_phi=ph, out_dphi=dp); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_BASIS))
ts = ts[3:]
byts = B * (8 * 4 * m + 16) + 8 * m
print("basis min %.4f median %.4f ms -> %.0f / %.0f GB/s (%.3f / %.3f of 8 TB/s)" % (min(ts), sorted(ts)[len(ts)//2], byts / min(ts) / 1e6, byts / sorted(ts)[len(ts)//2] / 1e6, byts / min(ts) / 8e9, byts / sorted(ts)[len(ts)//2] / 8e9))
