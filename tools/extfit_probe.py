"""Census probe of the stepped fit of caller-evaluated models (tests/test_gpu_extfit.py): prints the problems whose success
class differs from the oracle's, with both reports and the trial-point traces side by side."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import varpro_amd as vp  # noqa: E402
from test_gpu_external import peaks_model, peaks_data, oracle_problem  # noqa: E402
from test_gpu_extfit import host_model, oracle_fits  # noqa: E402

rng = np.random.default_rng(2024)
m, B = 512, 4096
x = np.linspace(0.0, 10.0, m)
cm = peaks_model(x)
_truth, _c, Y, guess = peaks_data(rng, B, x, noise=1e-2)
guess = guess * (1 + rng.uniform(-0.15, 0.15, guess.shape))
ref = oracle_fits(cm, Y, guess)
bp = vp.BatchProblem(cm.shape(), Y)
a, C, rep, steps = bp.fit_with_model(host_model(cm), guess)
bad = np.nonzero((rep["termination"] > 0) != (ref[1] > 0))[0]
print("steps", steps, "disagreements", bad)
print("oracle terminations", np.bincount(ref[1] + 10), "device", np.bincount(rep["termination"] + 10))
dev = np.abs(rep["n_evals"] - ref[2])
print("evals equal %.3f within3 %.3f" % ((dev == 0).mean(), (dev <= 3).mean()))
only_device = os.environ.get("EXTFIT_PROBE_DEVICE_ONLY") == "1"
for b in bad:
    print("problem", b, "oracle term/nfev/obj", ref[1][b], ref[2][b], ref[3][b], "device", rep[b], "alpha", a[b], "oracle alpha", ref[0][b])
    p = oracle_problem(cm, Y[b])
    p.set_params(guess[b])
    r, tr = p.fit_trace()
    # device trace of this problem alone: step by step
    bp1 = vp.BatchProblem(cm.shape(), Y[b:b + 1])
    bp1.fit_begin(guess[b:b + 1])
    al = guess[b:b + 1].copy()
    rows = []
    for it in range(300):
        al, want, nact = bp1.fit_step_with_basis(cm.eval_batch(al), cm.derivs_batch(al))
        al = np.array(al)
        rows.append((al[0].copy(), float(np.asarray(bp1.lib and 0))))
        if nact == 0:
            break
    for i in range(0 if only_device else max(len(tr), len(rows))):
        o = tr[i] if i < len(tr) else None
        d = rows[i - 1][0] if 0 < i <= len(rows) else (guess[b] if i == 0 else None)
        print("  eval %2d oracle x=%s |r|=%s ratio=%s | device x=%s" % (i, None if o is None else o[:4], None if o is None else o[4], None if o is None else o[5], d))
