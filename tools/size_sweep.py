"""fit throughput of the double exponential across problem lengths, one launch at a time: wave kernel vs slot kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth, _lib
for m, B in ((4096, 16384), (2048, 32768), (1024, 65536), (512, 131072), (256, 262144), (128, 262144), (32, 262144)):
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    dev = torch.device("cuda", 0)
    Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
    out = []
    for kern in ("wave", "slots"):
        bp = vp.BatchProblem(mdl, Y, x=x); bp.set_timing(True); bp.set_fit_kernel(kern)
        ts = []
        for _ in range(5):
            a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
        r = bp.report_to_numpy(rep)
        out.append("%s %.3f ms %.2f Mfits/s" % (kern, min(ts), B / min(ts) / 1e3))
        bp.close()
    print("m %5d B %6d  %s   evals/fit %.2f" % (m, B, "  |  ".join(out), r["n_evals"].mean()), flush=True)
    del Y
