"""double exponential + offset beyond 1024 rows: fit and evaluate throughput per length (which kernel set serves it)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth, _lib
dev = torch.device("cuda", 0)
for m, B in ((1024, 32768), (1100, 32768), (1536, 32768), (2048, 32768), (2100, 16384), (3000, 16384), (4096, 16384)):
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
    bp = vp.BatchProblem(mdl, Y, x=x); bp.set_timing(True)
    ts = []
    for _ in range(4):
        a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
    te = []
    for _ in range(4):
        bp.evaluate(g, want_residuals=False, want_jacobian=False); te.append(bp.last_kernel_ms(_lib.VP_KERNEL_EVALUATE))
    r = bp.report_to_numpy(rep)
    ne = float(r["n_evals"].sum())
    print("m %5d B %6d fit %.3f ms %.2f Mfits/s  %.2f ns per evaluation and row | evaluate %.3f ms" % (m, B, min(ts), B / min(ts) / 1e3, min(ts) * 1e6 / ne / m, min(te)), flush=True)
    bp.close(); del Y
