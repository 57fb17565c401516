"""BASELINE configs[4]: fp32 five-exponential + offset (n=6, q=5), m=4096, batch=8192 on 1 GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, collections
import varpro_amd as vp
from varpro_amd import synth, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = synth.multi_exp_batch(B, 5, 4096, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
dev = torch.device("cuda", 0)
Y = torch.from_numpy(d["Y"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=torch.from_numpy(d["x"]).to(dev))
bp.set_timing(True)
for name, wr, wj in (("evaluate (c,cost)", False, False), ("evaluate (+r,J)", True, True)):
    ts = []
    for _ in range(4):
        bp.evaluate(g, want_residuals=wr, want_jacobian=wj); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_EVALUATE))
    print("%-20s %8.3f ms  %8.2f Mevals/s" % (name, min(ts), B / min(ts) / 1e3))
ts = []
for _ in range(3):
    a, C, rep = bp.fit(g); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
r = bp.report_to_numpy(rep)
print("fit                  %8.3f ms  %8.3f Mfits/s  evals/fit %.1f max %d" % (min(ts), B / min(ts) / 1e3, r["n_evals"].mean(), r["n_evals"].max()))
print("terminations", collections.Counter(r["termination"].tolist()))
