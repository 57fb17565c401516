import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth, _lib
from models import double_exp_builder_model
dev = torch.device("cuda", 0)
B = 65536
for m in (128, 256, 512, 1024):
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
    out = []
    for name, mdl in (("multiexp", vp.multi_exponential_model(d["x"], d["tau_guess"][0])), ("builder-RT", double_exp_builder_model(d["x"], d["tau_guess"][0]))):
        bp = vp.BatchProblem(mdl, Y, x=x); bp.set_timing(True)
        ts = []
        for _ in range(4):
            a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
        r = bp.report_to_numpy(rep)
        out.append("%s %.3f ms (%.1f M fits/s, %.2f evals)" % (name, min(ts), B / min(ts) / 1e3, r["n_evals"].mean()))
        bp.close()
    print("m %4d: %s" % (m, " | ".join(out)), flush=True)
