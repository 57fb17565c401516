"""vp_evaluate (r and J out) of the double exponential + offset over the variants that pick different kernels: full / shorter
lengths, weights, longer sets.  usage: PYTHONPATH=. python tools/eval_variants_probe.py [B]   (VARPRO_HIP_LIBRARY selects the build)"""
import sys

import numpy as np
import torch

import varpro_amd as vp
from varpro_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda:0")
for m, weighted in ((1024, False), (1024, True), (1000, False), (1000, True), (900, False), (1536, False), (1536, True),
                    (2048, False), (2048, True)):
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    Y = torch.from_numpy(d["Y"]).to(dev)
    g = torch.from_numpy(d["tau_guess"]).to(dev)
    w = torch.from_numpy(0.5 + np.random.default_rng(1).random(m)).to(dev) if weighted else None
    bp = vp.BatchProblem(mdl, Y, x=torch.from_numpy(d["x"]).to(dev), weights=w)
    for _ in range(3):
        ev = bp.evaluate(g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ev = bp.evaluate(g)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = B * 8 * (4 * m + 5) / 1e9
    chk = float(ev["r"].abs().sum()) + float(ev["J"].abs().sum())
    print("m %5d %-10s %.3f ms  %.2f TB/s  checksum %.12e" % (m, "weighted" if weighted else "unit", ms, gb / ms, chk))
    bp.close()
    del Y, ev
