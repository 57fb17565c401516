"""phase clocks of mrhs_lm_kernel (library built with -DVP_MRHS_LM_CLOCKS): 100 MHz ticks -> microseconds per LM step:
load state + reduce the partial sums | accept/terminate + Gram -> QR | lmpar + next trial point"""
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
for i in range(int(rep["n_evals"][0])):
    print("step %2d: reduce %.2f us  after_eval+gram %.2f us  next_step %.2f us" % (i, tr[0, i, 4] / 100, tr[0, i, 5] / 100, tr[0, i, 6] / 100))
