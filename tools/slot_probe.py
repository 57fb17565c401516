"""wave kernel vs slot kernel on one box: python tools/slot_probe.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib
dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[1:] if a.isdigit()] or [65536, 4096]:
    d = synth.double_exp_batch(B, m=1024, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
    res = {}
    for kern in ("wave", "slots", "wave", "slots"):
        bp = vp.BatchProblem(mdl, Y, x=x); bp.set_timing(True); bp.set_fit_kernel(kern)
        ts = []
        for _ in range(8):
            a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
        r = bp.report_to_numpy(rep)
        print("B %6d %-6s fit min %.3f median %.3f ms  %.2f Mfits/s  evals %d  ok %d  cost %.9e" % (
            B, kern, min(ts), sorted(ts)[len(ts) // 2], B / min(ts) / 1e3, r["n_evals"].sum(), (r["termination"] > 0).sum(), np.nansum(r["objective"])), flush=True)
        res[kern] = (a.cpu().numpy(), r)
        bp.close()
    print("   identical reports:", np.array_equal(res["wave"][1]["n_evals"], res["slots"][1]["n_evals"]),
          np.array_equal(res["wave"][0], res["slots"][0], equal_nan=True))
