"""multi-wave-group slot kernel vs one-group-per-problem kernel (fp32 5-exp m=4096): python tools/w4_probe.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import varpro_amd as vp
from varpro_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = synth.multi_exp_batch(B, 5, 4096, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
res = {}
for kern in ("wave", "slots"):
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"]); bp.set_fit_kernel(kern)
    a, c, rep = bp.fit(d["tau_guess"])
    print(kern, "evals", rep["n_evals"][:8], "term", rep["termination"][:8], flush=True)
    res[kern] = (a, rep); bp.close()
print("same evals", np.array_equal(res["wave"][1]["n_evals"], res["slots"][1]["n_evals"]), "same alpha", np.array_equal(res["wave"][0], res["slots"][0], equal_nan=True))
