"""An ill-conditioned 3-exponential + offset evaluation: GPU result dumped to gpurun_out/illcond.npz so that it can
be compared with the 50-digit mpmath evaluation (tests/golden/make_golden.py machinery) next to the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import varpro_amd as vp
rng = np.random.default_rng(77)
m, B = 129, 6
x = np.linspace(0.0, 12.0, m)
taus = np.array([[1.0, 1.05, 1.1], [0.8, 0.82, 5.0], [2.0, 2.1, 2.2], [0.5, 3.0, 3.1], [1.0, 1.02, 1.04], [3.0, 3.3, 3.6]])
c = rng.uniform(1, 50, (B, 4))
Y = sum(c[:, j:j + 1] * np.exp(-x / taus[:, j:j + 1]) for j in range(3)) + c[:, 3:]
Y = Y + 1e-4 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal((B, m))
guess = taus * rng.uniform(0.9, 1.1, (B, 3))
mdl = vp.multi_exponential_model(x, guess[0], offset=True)
out = dict(x=x, Y=Y, guess=guess)
if vp.device_count() > 0:
    bp = vp.BatchProblem(mdl, Y, x=x)
    ev = bp.evaluate(guess)
    out.update(C_gpu=ev["C"], r_gpu=ev["r"], J_gpu=ev["J"])
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez("gpurun_out/illcond.npz", **out)
    print("saved")
