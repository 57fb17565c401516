import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import varpro_amd as vp
from oracle import oracle as O
import test_gpu_gram_parity as T
np.set_printoptions(linewidth=200, precision=6)
d, w = T._problem(64, 4096, "uniform", True)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
bp = vp.BatchProblem(mdl, d["Y"], x=d["x"], weights=w)
alpha, C, rep = bp.fit(d["tau_guess"])
yw = np.asarray(bp.weighted_data()).astype(np.float64)
bp.close()
grid64 = T._lattice(d["x"]); w64 = w.astype(np.float64)
for b in range(64):
    a64 = alpha[b].astype(np.float64)
    ref = O.evaluate_batch(mdl, grid64, (yw[b] / w64)[None], a64[None], w=w64)
    err = np.abs(C[b].astype(np.float64) - ref["C"][0]).max() / np.abs(ref["C"][0]).max()
    Phi = np.concatenate([np.exp(-grid64[None] / a64[:, None]), np.ones((1, grid64.size))]) * w64
    print(b, rep["termination"][b], rep["n_evals"][b], "obj %.6e" % rep["objective"][b], "ref %.6e" % ref["cost"][0], "cerr %.2e" % err, "cond %.2e" % np.linalg.cond(Phi.T), "alpha", alpha[b], "C", C[b] if err > 1e-2 else "", "Cref", ref["C"][0] if err > 1e-2 else "")
