import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import varpro_amd as vp
from varpro_amd import synth
B = 65536
d = synth.double_exp_batch(B, m=1024, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
for rep in range(3):
    t0 = time.perf_counter(); bp = vp.BatchProblem(mdl, d["Y"], x=d["x"]); t1 = time.perf_counter()
    a, c, r = bp.fit(d["tau_guess"]); t2 = time.perf_counter(); bp.close()
    print("create %.1f ms  fit+D2H %.1f ms  -> %.2f Mfits/s PCIe-inclusive" % ((t1-t0)*1e3, (t2-t1)*1e3, B/(t2-t0)/1e6))
