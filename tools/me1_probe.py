import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth, _lib
for off in (True, False):
    B, m = 65536, 1024
    d = synth.multi_exp_batch(B, 1, m, [2.0], noise=1e-3, spread=0.3, guess_spread=0.3)
    Y = d["Y"] if off else d["Y"] - d["c_true"][:, 1:2]
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], offset=off)
    dev = torch.device("cuda", 0)
    bp = vp.BatchProblem(mdl, torch.from_numpy(Y).to(dev), x=torch.from_numpy(d["x"]).to(dev)); bp.set_timing(True)
    g = torch.from_numpy(d["tau_guess"]).to(dev)
    ts = []
    for _ in range(6):
        a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
    r = bp.report_to_numpy(rep)
    print("1 exp offset=%d  fit %.3f ms  %.2f Mfits/s  evals %d ok %.3f" % (off, min(ts), B / min(ts) / 1e3, r["n_evals"].sum(), (r["termination"] > 0).mean()))
    bp.close()
