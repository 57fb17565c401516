"""Parity census of a whole batch on the GPU box: python tools/census_probe.py [B] [cfg]  (cfg: 1 = double-exp fp64, 4 = five-exp fp32)"""
import json
import sys
import time

import numpy as np

import varpro_amd as vp
from oracle import census as CS
from oracle import oracle as O
from varpro_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 1
thr = min(16, O.max_threads())
if cfg == 1:
    d = synth.double_exp_batch(B, m=1024, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    a, c, rep = bp.fit(d["tau_guess"])
    t0 = time.time()
    ao, co, ro, _s = O.fit_batch(mdl, d["x"], d["Y"], d["tau_guess"], n_threads=thr)
    print("oracle: %.1f s on %d threads" % (time.time() - t0, thr))
else:
    d = synth.multi_exp_batch(B, 5, 4096, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    a, c, rep = bp.fit(d["tau_guess"])
    mdl64 = vp.multi_exponential_model(d["x"].astype(np.float64), d["tau_guess"][0].astype(np.float64))
    t0 = time.time()
    e32 = float(np.finfo(np.float32).eps)
    ao, co, ro, _s = O.fit_batch(mdl64, d["x"].astype(np.float64), d["Y"].astype(np.float64), d["tau_guess"].astype(np.float64), n_threads=thr,
                                 opts=O.default_opts(ftol=30 * e32, xtol=30 * e32, gtol=30 * e32))
    print("oracle: %.1f s on %d threads" % (time.time() - t0, thr))
res = CS.census(rep, a, ro, ao)
print(json.dumps(res, indent=1))
