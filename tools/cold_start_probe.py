"""Wall time from process start to the first fit's results (library load, code-object extraction, first launches): the cost a
compressed offload bundle (--offload-compress) could add.  usage: PYTHONPATH=. python tools/cold_start_probe.py"""
import time
t0 = time.time()
import numpy as np
import varpro_amd as vp
from varpro_amd import synth
t1 = time.time()
d = synth.double_exp_batch(4096, m=1024, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
t2 = time.time()
bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
t3 = time.time()
alpha, C, rep = bp.fit(d["tau_guess"])
t4 = time.time()
ev = bp.evaluate(d["tau_guess"])
t5 = time.time()
alpha, C, rep = bp.fit(d["tau_guess"])
t6 = time.time()
print({"import_s": round(t1 - t0, 3), "handle_s": round(t3 - t2, 3), "first_fit_s": round(t4 - t3, 3), "first_evaluate_s": round(t5 - t4, 3),
       "second_fit_s": round(t6 - t5, 4)})
