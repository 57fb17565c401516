"""phase clocks of the slot fit kernel (library built with -DVP_FIT2_CLOCKS): wave 0 of workgroup 0, summed over its rounds;
with B = 1 the wave runs one lone fit: cycles per LM evaluation of the vector phase / scalar phase / refill"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import varpro_amd as vp
from varpro_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
d = synth.double_exp_batch(B, m=1024, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
bp.set_fit_kernel("slots")
a, C, rep, tr = bp.fit_trace(d["tau_guess"], max_rows=160)
ck = tr[0, 159, :3]
n = rep["n_evals"][0]
print("problem 0 evals", n, "clocks vector/scalar/refill", ck, " per evaluation:", ck / max(1, n))
print("scalar phase sections [load | update+tests | gtest+diag | lmpar | prered+trial | write-back]:", tr[0, 158, :6])
