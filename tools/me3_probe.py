"""triple exponential + offset (n = 4, q = 3) fp64 at m = 2048 (R = 32 kernels: 232-543 spilled VGPRs) and m = 1024"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib
dev = torch.device("cuda", 0)
for m in [int(a) for a in sys.argv[1:]] or (1024, 2048):
    for B in (16384,):
        d = synth.multi_exp_batch(B, 3, m, [1.0, 3.0, 7.0], noise=1e-3, spread=0.1, guess_spread=0.1)
        mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
        for kern in ("wave", "slots"):
            bp = vp.BatchProblem(mdl, torch.from_numpy(d["Y"]).to(dev), x=torch.from_numpy(d["x"]).to(dev))
            bp.set_timing(True); bp.set_fit_kernel(kern)
            g = torch.from_numpy(d["tau_guess"]).to(dev)
            ts = []
            for _ in range(4):
                a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
            r = bp.report_to_numpy(rep)
            te = []
            for _ in range(4):
                bp.evaluate(g, want_residuals=False, want_jacobian=False); te.append(bp.last_kernel_ms(_lib.VP_KERNEL_EVALUATE))
            print("m %d B %d %-5s fit %.3f ms %.2f Mfits/s evals/fit %.1f max %d ok %.3f | evaluate %.3f ms" % (m, B, kern, min(ts), B / min(ts) / 1e3, r["n_evals"].mean(), r["n_evals"].max(), (r["termination"] > 0).mean(), min(te)))
            bp.close()
