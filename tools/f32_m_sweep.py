"""double exponential + offset, fp32, across problem lengths"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth, _lib
dev = torch.device("cuda", 0)
B = 32768
for m in (128, 200, 512, 1024, 1100, 2048):
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    x32, Y32, g32 = d["x"].astype(np.float32), d["Y"].astype(np.float32), d["tau_guess"].astype(np.float32)
    mdl = vp.multi_exponential_model(x32, g32[0], dtype=np.float32)
    bp = vp.BatchProblem(mdl, torch.from_numpy(Y32).to(dev), x=torch.from_numpy(x32).to(dev)); bp.set_timing(True)
    g = torch.from_numpy(g32).to(dev)
    ts = []
    for _ in range(4):
        a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
    r = bp.report_to_numpy(rep)
    print("m %5d B %d fit %.3f ms %.2f Mfits/s evals/fit %.1f ok %.3f" % (m, B, min(ts), B / min(ts) / 1e3, r["n_evals"].mean(), (r["termination"] > 0).mean()), flush=True)
    bp.close()
