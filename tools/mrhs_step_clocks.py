"""phase clocks of mrhs_step_kernel (library built with -DVP_MRHS_STEP_CLOCKS; shader-clock ticks / 2400 = us at 2.4 GHz):
loads + group reduction | LM step | columns (exp) | Householder sweep | R^-1, rank test | back-sweep (G, thin Q) | Gram, stores"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import varpro_amd as vp
from varpro_amd import synth
d = synth.mrhs_triple_exp(S=16384, m=2048)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"], offset=True)
bp = vp.BatchProblem(mdl, d["Y"][None], x=d["x"])
for _ in range(2):
    a, C, rep, tr = bp.fit_trace(d["tau_guess"][None], max_rows=12)
print("evaluations", rep["n_evals"][0])
names = ["loads+reduce", "LM", "columns", "house_qr", "Rinv", "backsweep", "gram+stores"]
for i in range(int(rep["n_evals"][0])):
    print("step %2d: " % i + "  ".join("%s %.2f" % (n, tr[0, i, k] / 2400.0) for k, n in enumerate(names)))
