"""Length-agnostic fit kernels (vp_block.hpp): parity vs the oracle at a small batch + throughput over m.
usage: PYTHONPATH=. python tools/blk_probe.py"""
import sys
import time

import numpy as np
import torch

import varpro_amd as vp
from oracle import census as CS
from oracle import oracle as O
from varpro_amd import synth

dev = torch.device("cuda:0")
for m in (200, 1000, 1001, 1024, 5000):
    B = 512
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"], stream_rows=True)
    a, c, rep = bp.fit(d["tau_guess"])
    ao, co, ro, _ = O.fit_batch(mdl, d["x"], d["Y"], d["tau_guess"], n_threads=16)
    res = CS.census(rep, a, ro, ao, max_listed=5)
    ok = ro["termination"] > 0
    dc = np.abs(c - co)[ok].max(1) / np.abs(co)[ok].max(1)
    print("m=%d" % m, {k: res[k] for k in ("same_success_class", "failed_device", "failed_oracle", "objective_rel_diff_median_common_successes",
                                          "objective_rel_diff_max_common_successes", "share_evals_within_3", "sum_evals_device", "sum_evals_oracle")},
          "max rel dc %.2e" % dc.max())
    for x in res["disagreements"]:
        print("   ", x)
    bp.close()
for (m, B) in ((1024, 65536), (2048, 32768), (5000, 16384), (10000, 8192), (100000, 1024)):
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    Y = torch.from_numpy(d["Y"]).to(dev)
    x = torch.from_numpy(d["x"]).to(dev)
    g = torch.from_numpy(d["tau_guess"]).to(dev)
    for stream in (True, False):
        bp = vp.BatchProblem(mdl, Y, x=x, stream_rows=stream)
        bp.set_timing(True)
        ts = []
        for _ in range(4):
            a, c, rep = bp.fit(g, want_coefficients=False)
            ts.append(bp.last_kernel_ms(2))
        r = bp.report_to_numpy(rep)
        print("m=%6d B=%6d %-9s %8.3f ms  %8.3f M fits/s  evals/fit %.2f failed %d" % (m, B, "streamed" if stream else "default", min(ts), B / min(ts) / 1e3,
                                                                                      r["n_evals"].mean(), (r["termination"] <= 0).sum()))
        bp.close()
