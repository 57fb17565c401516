"""The one success-class disagreement of the configs[3] census (shard 3, problem 29433): device and oracle traces side by side.
usage: PYTHONPATH=. python tools/census_disagreement_probe.py"""
import numpy as np
import varpro_amd as vp
from oracle import oracle as O
from varpro_amd import synth
np.set_printoptions(precision=12, linewidth=200)
first = 3*65536 + 29433
d = synth.double_exp_batch(4, m=1024, first_problem=first, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
print("guess", d["tau_guess"][0], "true", d["tau_true"][0] if "tau_true" in d else None)
bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
a, c, rep, tr = bp.fit_trace(d["tau_guess"], max_rows=100)
print("device rep", rep[0], "alpha", a[0])
print("device trace:\n", tr[0][:8])
p = O.Problem(mdl, d["x"], d["Y"][0])
p.set_params(d["tau_guess"][0])
r, tro = p.fit_trace(max_rows=100)
print("oracle rep", r.termination, r.n_evals, r.objective)
print("oracle trace:\n", tro[:8])
# evaluate at the device's 2nd/3rd trial points with both
for k in range(1, 4):
    al = tr[0][k][:2]
    if not np.isfinite(al).all(): break
    ev = bp.evaluate(np.tile(al, (4, 1)))
    ref = O.evaluate_batch(mdl, d["x"], d["Y"][:1], al[None, :])
    print("trial", k, al, "device status/cost", ev["status"][0], ev["cost"][0], "oracle status/cost", ref["status"][0], ref["cost"][0])
