"""what do the two (normally empty) re-fit launches behind every vp_fit cost on the headline workload?  Two batches in flight as in
bench.py, with and without the second launches (vp_debug_set_refit), alternating.  usage: python tools/refit_cost_probe.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import varpro_amd as vp
from varpro_amd import synth
B, m, K = 65536, 1024, 60
dev = torch.device("cuda", 0)
hs, gs = [], []
streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)]
for i in range(2):
    d = synth.double_exp_batch(B, m=m, first_problem=i * B, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    with torch.cuda.stream(streams[i]):
        hs.append(vp.BatchProblem(mdl, torch.from_numpy(d["Y"]).to(dev), x=torch.from_numpy(d["x"]).to(dev)))
    gs.append(torch.from_numpy(d["tau_guess"]).to(dev))
red = [torch.zeros(4, dtype=torch.float64, device=dev) for _ in range(2)]
def run(nslots, n):
    for k in range(n):
        i = k % nslots
        with torch.cuda.stream(streams[i]):
            hs[i].fit(gs[i], want_coefficients=False)
            hs[i].summary_device(red[i])
def timed(nslots):
    run(nslots, 6); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(nslots, K); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for rep in range(3):
    for on in (True, False):
        for h in hs: h.set_refit(on)
        print("refit %-5s  one at a time %.4f ms   two in flight %.4f ms" % (on, timed(1), timed(2)))
