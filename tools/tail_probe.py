"""How much of the fit kernel's time is the straggler tail?  Re-run the same batch ordered longest-fit-first
(using the evaluation counts of a first run) and compare kernel times; plus the PCIe-inclusive host-pointer rate."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d = synth.double_exp_batch(B, m=1024, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda", 0)
x = torch.from_numpy(d["x"]).to(dev)

def run(order, tag):
    Y = torch.from_numpy(d["Y"][order]).to(dev); g = torch.from_numpy(d["tau_guess"][order]).to(dev)
    bp = vp.BatchProblem(mdl, Y, x=x); bp.set_timing(True)
    ts = []
    for _ in range(5):
        a, c, rep = bp.fit(g); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
    r = bp.report_to_numpy(rep)
    print("%-28s %.3f ms (min of 5)  evals total %d" % (tag, min(ts), r["n_evals"].sum()))
    bp.close()
    return r["n_evals"]

ne = run(np.arange(B), "natural order")
run(np.argsort(-ne, kind="stable"), "longest first (oracle order)")
run(np.argsort(ne, kind="stable"), "shortest first (worst)")
# PCIe-inclusive: host numpy buffers handed over, library stages to HBM, fits, returns results to host
t0 = time.perf_counter()
bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
t1 = time.perf_counter()
a, c, rep = bp.fit(d["tau_guess"])
t2 = time.perf_counter()
bp.close()
print("host-pointer mode: create (H2D of Y) %.1f ms, fit incl. result D2H %.1f ms -> %.2f Mfits/s PCIe-inclusive" %
      ((t1 - t0) * 1e3, (t2 - t1) * 1e3, B / (t2 - t0) / 1e6))
