"""fit time of the double exponential + offset at lengths that do not fill their kernel set (row-validity masks in the row
source).  usage: PYTHONPATH=. python tools/pad_probe.py [B]   (VARPRO_HIP_LIBRARY selects the build)"""
import sys

import numpy as np
import torch

import varpro_amd as vp
from varpro_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
for m in (1024, 1000, 900, 800, 700, 500, 400):
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    Y = torch.from_numpy(d["Y"]).to(dev)
    g = torch.from_numpy(d["tau_guess"]).to(dev)
    bp = vp.BatchProblem(mdl, Y, x=torch.from_numpy(d["x"]).to(dev))
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        a, _c, rep = bp.fit(g, want_coefficients=False)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    r = bp.report_to_numpy(rep)
    print("m %5d  fit %.3f ms (median %.3f)  evals %d  cost %.9e" % (m, min(ts[2:]), sorted(ts[2:])[3], r["n_evals"].sum(), np.nansum(r["objective"])))
    bp.close()
    del Y
