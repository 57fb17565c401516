"""debug: Gram fit kernel vs fp64 oracle on a few cfg4 problems (host mode, traces)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, collections
import varpro_amd as vp
from varpro_amd import synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
B = 32
d = synth.multi_exp_batch(B, 5, 4096, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
a, C, rep, tr = bp.fit_trace(d["tau_guess"], max_rows=64)
print("terminations", collections.Counter(rep["termination"].tolist()), "evals", rep["n_evals"][:16])
np.set_printoptions(precision=5, linewidth=200, suppress=False)
for b in range(3):
    print("problem", b, "term", rep["termination"][b], "nfev", rep["n_evals"][b], "objective", rep["objective"][b])
    for r in range(min(8, rep["n_evals"][b] + 1)):
        print("   ", tr[b, r])
try:
    import oracle as orc
    mo = orc.multi_exp_model(5, True) if hasattr(orc, "multi_exp_model") else None
except Exception as e:
    print("oracle import failed", e)
