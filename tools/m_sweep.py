import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth, _lib
dev = torch.device("cuda", 0)
for m, B in ((1024, 65536), (1000, 65536), (900, 65536), (768, 65536), (520, 65536), (512, 65536), (500, 65536), (300, 131072), (256, 131072), (200, 131072), (128, 262144), (100, 262144)):
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
    bp = vp.BatchProblem(mdl, Y, x=x); bp.set_timing(True)
    ts = []
    for _ in range(5):
        a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
    r = bp.report_to_numpy(rep)
    ne = float(r["n_evals"].sum())
    print("m %5d B %6d fit %.3f ms %.2f Mfits/s  %.1f ns per evaluation  %.2f ns per evaluation and row" % (m, B, min(ts), B / min(ts) / 1e3, min(ts) * 1e6 / ne, min(ts) * 1e6 / ne / m), flush=True)
    bp.close(); del Y
