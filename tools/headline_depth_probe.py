"""Headline workload (B = 65 536, m = 1024, fp64 double exponential) with 1 ... 4 batches in flight: `depth` device-pointer
handles, each on its own HIP stream, fitted round-robin (the schedule of bench.py's `value` at depth 2).
Usage: PYTHONPATH=. python tools/headline_depth_probe.py [out.json]"""
import json
import sys
import time

import torch

import varpro_amd as vp
from varpro_amd import synth

dev = torch.device("cuda:0")
B, m = 65536, 1024
d = synth.double_exp_batch(B, m=m, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
Y = torch.from_numpy(d["Y"]).to(dev)
x = torch.from_numpy(d["x"]).to(dev)
g = torch.from_numpy(d["tau_guess"]).to(dev)
out = {}
for depth in (1, 2, 3, 4):
    strs = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    hs = []
    for s in strs:
        with torch.cuda.stream(s):
            hs.append(vp.BatchProblem(mdl, Y, x=x))
    torch.cuda.synchronize()

    def run(n):
        for _ in range(n):
            for i in range(depth):
                with torch.cuda.stream(strs[i]):
                    hs[i].fit(g, want_coefficients=False)
    run(2)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        rounds = 24 // depth
        t0 = time.perf_counter()
        run(rounds)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3 / (rounds * depth))
    out[str(depth)] = {"ms_per_batch": best, "fits_per_s": B / (best * 1e-3)}
    print(depth, out[str(depth)], flush=True)
    for h in hs:
        h.close()
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
