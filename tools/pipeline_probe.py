"""Throughput of back-to-back batches on ONE stream vs alternating TWO handles on two streams (the straggler tail of
batch k overlaps the bulk of batch k+1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth
B, K = 65536, 40
d = synth.double_exp_batch(B, m=1024, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda", 0)
Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
for ns in (1, 2, 3):
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    hs, reds = [], []
    for s in streams:
        with torch.cuda.stream(s):
            hs.append(vp.BatchProblem(mdl, Y, x=x)); reds.append(torch.zeros(4, dtype=torch.float64, device=dev))
    torch.cuda.synchronize()
    def step(k):
        i = k % ns
        with torch.cuda.stream(streams[i]):
            hs[i].fit(g, want_coefficients=False); hs[i].summary_device(reds[i])
    for k in range(4): step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K): step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d stream(s): %.3f ms/step  %.2f Mfits/s   sums %s" % (ns, dt / K * 1e3, B * K / dt / 1e6, [float(r[3]) for r in reds]))
    for h in hs: h.close()
