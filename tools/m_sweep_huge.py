"""double / single exponential + offset beyond 4096 rows"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth, _lib
dev = torch.device("cuda", 0)
B = 8192
for nexp in (2, 1):
    for m in (4096, 5000, 6144, 8192, 8200):
        d = synth.double_exp_batch(B, m=m, noise=1e-3) if nexp == 2 else synth.multi_exp_batch(B, 1, m, [2.0], noise=1e-3, spread=0.2, guess_spread=0.2)
        mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
        bp = vp.BatchProblem(mdl, torch.from_numpy(d["Y"]).to(dev), x=torch.from_numpy(d["x"]).to(dev)); bp.set_timing(True)
        g = torch.from_numpy(d["tau_guess"]).to(dev)
        ts = []
        for _ in range(3):
            a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
        r = bp.report_to_numpy(rep)
        print("%d exp m %5d B %d fit %.3f ms %.3f Mfits/s ok %.3f" % (nexp, m, B, min(ts), B / min(ts) / 1e3, (r["termination"] > 0).mean()), flush=True)
        bp.close()
