import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib
B, m = 65536, 1024
d = synth.double_exp_batch(B, m=m, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda", 0)
Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=x); bp.set_timing(True)
for pad in (0, 4096, 1 << 20):
    buf = torch.empty(B * 2 * m * 2 + pad // 8 + 16, dtype=torch.float64, device=dev)
    ph = buf[:B * 2 * m].view(B, 2, m); off = B * 2 * m + pad // 8; dp = buf[off:off + B * 2 * m].view(B, 2, m)
    ts = []
    for _ in range(60):
        bp.basis(g, skip_invariant=True, out_phi=ph, out_dphi=dp); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_BASIS))
    ts = np.array(ts[5:])
    print("pad %8d  min %.3f p25 %.3f median %.3f p75 %.3f max %.3f" % (pad, ts.min(), np.percentile(ts, 25), np.median(ts), np.percentile(ts, 75), ts.max()), " first 12:", np.round(ts[:12], 3))
