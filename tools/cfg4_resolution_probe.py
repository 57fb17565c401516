"""configs[4] (fp32 five exponentials, Gram fit kernel): how many fits END on a point where the Gram evaluation has no
resolution left -- the objective the kernel reports against the fp64 oracle's cost at the returned parameters.
PYTHONPATH=. python tools/cfg4_resolution_probe.py [B]"""
import sys, os, collections, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import varpro_amd as vp
from oracle import oracle as O
from varpro_amd import synth, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = synth.multi_exp_batch(B, 5, 4096, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
bp.set_timing(True)
ts = []
for _ in range(3):
    alpha, C, rep = bp.fit(d["tau_guess"])
    ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
bp.close()
m = d["x"].shape[-1]
t0 = float(d["x"][0]); dt = (float(d["x"][-1]) - t0) / (m - 1)
grid64 = t0 + np.arange(m) * dt
ok = rep["termination"] > 0
ref = O.evaluate_batch(vp.multi_exponential_model(grid64, d["tau_guess"][0].astype(np.float64)), grid64, d["Y"][ok].astype(np.float64),
                       alpha[ok].astype(np.float64), n_threads=min(16, O.max_threads()), want_jac=False)
rel = np.abs(rep["objective"][ok] - ref["cost"]) / ref["cost"]
out = {"B": B, "fit_ms": min(ts), "mean_evals": float(rep["n_evals"].mean()), "max_evals": int(rep["n_evals"].max()),
       "terminations": dict(collections.Counter(int(t) for t in rep["termination"])), "failed": float((~ok).mean()),
       "objective_vs_true_cost_at_the_returned_point": {"median": float(np.median(rel)), "p99": float(np.percentile(rel, 99)),
                                                        "share_above_1e-3": float((rel > 1e-3).mean()), "share_above_1e-2": float((rel > 1e-2).mean()),
                                                        "share_above_1e-1": float((rel > 1e-1).mean()), "max": float(rel.max())},
       "true_cost_mean_over_successes": float(ref["cost"].mean())}
print(json.dumps(out))
