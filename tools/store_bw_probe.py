"""ceiling for the basis kernel: how fast does this box take pure streaming stores (torch fill_ / zero_ of 2.1 GB) and a copy?"""
import torch, time
dev = torch.device("cuda", 0)
n = 2148540416 // 8
x = torch.empty(n, dtype=torch.float64, device=dev); y = torch.empty(n, dtype=torch.float64, device=dev)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.fill_(1.5)); print("fill_  %.3f ms  %.0f GB/s written" % (ms, n * 8 / ms / 1e6))
ms = t(lambda: x.zero_()); print("zero_  %.3f ms  %.0f GB/s written" % (ms, n * 8 / ms / 1e6))
ms = t(lambda: y.copy_(x)); print("copy_  %.3f ms  %.0f GB/s read+written" % (ms, 2 * n * 8 / ms / 1e6))
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, varpro_amd as vp
from varpro_amd import synth, _lib
d = synth.double_exp_batch(65536, m=1024, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
bp = vp.BatchProblem(mdl, torch.from_numpy(d["Y"]).to(dev), x=torch.from_numpy(d["x"]).to(dev)); bp.set_timing(True)
g = torch.from_numpy(d["tau_guess"]).to(dev)
ts = []
for _ in range(12):
    bp.basis(g); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_BASIS))
print("basis_kernel min %.3f median %.3f ms -> %.0f GB/s (median)" % (min(ts), sorted(ts)[6], 2148540416 / sorted(ts)[6] / 1e6))
