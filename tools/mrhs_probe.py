"""BASELINE configs[2] at full size: 1 alpha shared by S=16384 right-hand sides, m=2048, triple-exp + offset."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
d = synth.mrhs_triple_exp(S=S, m=m)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"], offset=True)
dev = torch.device("cuda", 0)
Y = torch.from_numpy(d["Y"][None]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=torch.from_numpy(d["x"]).to(dev))
g = torch.from_numpy(d["tau_guess"][None]).to(dev)
bp.set_timing(True)
T = 8
for name, wr, wj in (("evaluate (C,cost)", False, False), ("evaluate (+R)", True, False), ("evaluate (+R,J)", True, True)):
    ts = []
    for _ in range(4):
        bp.evaluate(g, want_residuals=wr, want_jacobian=wj); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_EVALUATE))
    byts = T * m * S * (1 + (1 if wr else 0) + (3 if wj else 0))
    print("%-20s %8.3f ms  %7.1f GB/s algorithmic" % (name, min(ts), byts / min(ts) / 1e6))
ts, te = [], []
for _ in range(8):
    t0 = time.perf_counter(); a, C, rep = bp.fit(g); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    te.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
r = bp.report_to_numpy(rep)
print("global fit            %8.3f ms wall, %.3f ms by the library's events  (%d evaluations, termination %d, objective %.3e)" % (min(ts), min(te), r["n_evals"][0], r["termination"][0], r["objective"][0]))
print("   per evaluation %.3f ms -> %.1f GB/s of Y streamed" % (min(ts) / r["n_evals"][0], T * m * S / (min(ts) / r["n_evals"][0]) / 1e6))
print("tau", a.cpu().numpy()[0], "true", d["tau_true"], " max|C-C_true|/max", float(np.abs(C.cpu().numpy()[0] - d["C_true"]).max() / np.abs(d["C_true"]).max()))
