"""global fit (double exponential + offset, one alpha shared by S right-hand sides) across problem lengths: which kernel set
serves a length, and what a pass costs there"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, varpro_amd as vp
from varpro_amd import _lib
dev = torch.device("cuda", 0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(0)
for m in (256, 500, 512, 600, 768, 1000, 1024, 1100, 1280, 1536, 1700, 1792, 2000, 2048):
    x = np.linspace(0.0, 12.5, m)
    Cm = rng.uniform(1, 50, (S, 3))
    Y = Cm[:, 0:1] * np.exp(-x / 1.0) + Cm[:, 1:2] * np.exp(-x / 3.0) + Cm[:, 2:3]
    mdl = vp.multi_exponential_model(x, [1.3, 3.6], offset=True)
    bp = vp.BatchProblem(mdl, torch.from_numpy(Y[None]).to(dev), x=torch.from_numpy(x).to(dev)); bp.set_timing(True)
    g = torch.tensor([[1.3, 3.6]], dtype=torch.float64, device=dev)
    te = []
    for _ in range(4):
        bp.evaluate(g, want_residuals=True, want_jacobian=True); te.append(bp.last_kernel_ms(_lib.VP_KERNEL_EVALUATE))
    tf = []
    for _ in range(4):
        t0 = time.perf_counter(); a, C, rep = bp.fit(g, want_coefficients=False); torch.cuda.synchronize(); tf.append((time.perf_counter() - t0) * 1e3)
    r = bp.report_to_numpy(rep)
    byt = 8 * m * S
    print("m %5d: evaluate(r,J) %.3f ms = %.2f TB/s | fit %.3f ms (%d evaluations) = %.2f TB/s of Y over the fit" % (m, min(te), byt * 4 / min(te) / 1e9, min(tf), r["n_evals"][0], byt * r["n_evals"][0] / min(tf) / 1e9), flush=True)
    bp.close()
