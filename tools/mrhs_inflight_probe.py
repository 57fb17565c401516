"""BASELINE configs[2] global fits in flight: D handles on D HIP streams, one host thread each (vp_fit of a handle with S > 1 waits
on the host for the fit's active count).  usage: python tools/mrhs_inflight_probe.py"""
import sys, time, threading
import numpy as np, torch
sys.path.insert(0, ".")
import varpro_amd as vp
from varpro_amd import synth
dev = torch.device("cuda", 0)
S2, m2 = 16384, 2048
d2 = synth.mrhs_triple_exp(S=S2, m=m2)
mdl2 = vp.multi_exponential_model(d2["x"], d2["tau_guess"], offset=True)
Y2 = torch.from_numpy(d2["Y"][None]).to(dev)
x2 = torch.from_numpy(d2["x"]).to(dev)
g2 = torch.from_numpy(d2["tau_guess"][None]).to(dev)
for depth in (1, 2, 3, 4):
    strs = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    hs = []
    for st in strs:
        with torch.cuda.stream(st):
            hs.append(vp.BatchProblem(mdl2, Y2.clone(), x=x2))
    torch.cuda.synchronize()
    def run_one(i, n):
        with torch.cuda.stream(strs[i]):
            for _ in range(n):
                hs[i].fit(g2, want_coefficients=False)
    def run(n):
        th = [threading.Thread(target=run_one, args=(i, n)) for i in range(depth)]
        for t in th: t.start()
        for t in th: t.join()
    run(14); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(20); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("depth %d: %.3f ms per fit" % (depth, dt * 1e3 / (20 * depth)))
    for h in hs: h.close()
