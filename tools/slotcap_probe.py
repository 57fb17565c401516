"""slot kernel at m = 512 (R = 8: 4 slots fit the LDS): python tools/slotcap_probe.py  -- run with A/B libraries built with
-DVP_SLOT_CAP=2 / 4 to price the amortisation of the scalar phase over more slots against the longer tail"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth, _lib
dev = torch.device("cuda", 0)
for m in (512,):
    for B in (65536, 262144):
        d = synth.double_exp_batch(B, m=m, noise=1e-3)
        mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
        bp = vp.BatchProblem(mdl, torch.from_numpy(d["Y"]).to(dev), x=torch.from_numpy(d["x"]).to(dev)); bp.set_timing(True)
        bp.set_fit_kernel("slots")
        g = torch.from_numpy(d["tau_guess"]).to(dev)
        ts = []
        for _ in range(6):
            a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
        r = bp.report_to_numpy(rep)
        print("m %d B %d fit min %.3f median %.3f ms  evals %d" % (m, B, min(ts), sorted(ts)[3], r["n_evals"].sum()))
        bp.close()
