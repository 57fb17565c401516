"""phase clocks of the Gram fit kernel (library built with -DVP_FITG_CLOCKS): group 0, summed over its rounds"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import varpro_amd as vp
from varpro_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = synth.multi_exp_batch(B, 5, 4096, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
a, C, rep, tr = bp.fit_trace(d["tau_guess"], max_rows=128)
ck = tr[0, 127, :4]
print("problem 0 evals", rep["n_evals"][0], "clocks vector/gram/scalar/refill", ck, "sum us @100MHz", ck.sum() / 100.0)
print("per round:", ck / max(1, rep["n_evals"][0]))
