import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth, _lib
dev = torch.device("cuda", 0)
for m, B in ((1024, 32768), (1100, 32768), (1280, 32768), (1300, 32768), (1536, 32768), (1600, 32768), (2048, 32768)):
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
    res = []
    for kern in ("auto", "wave", "slots"):
        bp = vp.BatchProblem(mdl, Y, x=x); bp.set_timing(True)
        if kern != "auto": bp.set_fit_kernel(kern)
        ts = []
        for _ in range(4):
            a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
        res.append("%s %.3f" % (kern, min(ts)))
        bp.close()
    print("m %5d B %6d fit ms: %s" % (m, B, "  ".join(res)), flush=True)
    del Y
