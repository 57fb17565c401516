"""The headline double-exponential problems fitted three ways -- vp_fit, the stepped external fit (columns from vp_basis) and
the oracle -- evaluation counts per problem side by side."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import varpro_amd as vp  # noqa: E402
from varpro_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

B, m = 4096, 1024
d = synth.double_exp_batch(B, m=m, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda:0")
Y = torch.from_numpy(d["Y"]).to(dev)
x = torch.from_numpy(d["x"]).to(dev)
guess = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=x)
a1, _c, r1 = bp.fit(guess)
r1 = bp.report_to_numpy(r1)
bpx = vp.BatchProblem(vp.ExternalModel(3, 2, [(0, 0), (1, 1)]), Y)
phi = torch.empty((B, 3, m), dtype=torch.float64, device=dev)
dphi = torch.empty((B, 2, m), dtype=torch.float64, device=dev)


def model(alpha, want):
    bp.basis(alpha, skip_invariant=False, out_phi=phi, out_dphi=dphi)
    return phi, dphi


a2, _c2, r2, steps = bpx.fit_with_model(model, guess)
r2 = bpx.report_to_numpy(r2)
xt = x[None, None, :]


def model_torch(alpha, want):
    a_ = alpha[:, :, None]
    e_ = torch.exp(-xt / a_)
    phi[:, 0:2] = e_
    phi[:, 2] = 1.0
    dphi[:] = e_ * xt / (a_ * a_)
    return phi, dphi


a4, _c4, r4, _s4 = bpx.fit_with_model(model_torch, guess)
r4 = bpx.report_to_numpy(r4)
bpn = vp.BatchProblem(mdl, Y, x=x, grid_recurrence=False)


def model_norec(alpha, want):
    bpn.basis(alpha, skip_invariant=False, out_phi=phi, out_dphi=dphi)
    return phi, dphi


a5, _c5, r5, _s5 = bpx.fit_with_model(model_norec, guess)
r5 = bpx.report_to_numpy(r5)
bpc = vp.BatchProblem(vp.ExternalModel(3, 2, [(1, 0), (2, 1)]), Y)  # the constant column FIRST


def model_const_first(alpha, want):
    a_ = alpha[:, :, None]
    e_ = torch.exp(-xt / a_)
    phi[:, 1:3] = e_
    phi[:, 0] = 1.0
    dphi[:] = e_ * xt / (a_ * a_)
    return phi, dphi


a6, _c6, r6, _s6 = bpc.fit_with_model(model_const_first, guess)
r6 = bpc.report_to_numpy(r6)
a3, _c3, r3, _s = O.fit_batch(mdl, d["x"], d["Y"], d["tau_guess"], n_threads=8)
print("sum evals: vp_fit %d, stepped external %d, oracle %d" % (r1["n_evals"].sum(), r2["n_evals"].sum(), r3["n_evals"].sum()))
for name, r in (("vp_fit", r1), ("stepped", r2), ("stepped, torch.exp columns", r4), ("stepped, vp_basis without recurrence", r5),
                ("stepped, torch.exp columns, constant column first", r6)):
    dv = r["n_evals"] - r3["n_evals"]
    print(name, "vs oracle: equal %.3f within3 %.3f mean diff %.3f" % ((dv == 0).mean(), (np.abs(dv) <= 3).mean(), dv.mean()),
          "hist", np.bincount(np.clip(dv, -5, 5) + 5))
bad = np.nonzero(0 * np.abs(r2["n_evals"] - r3["n_evals"]) > 0)[0][:3]
for b in bad:
    p = O.Problem(mdl, d["x"], d["Y"][b])
    p.set_params(d["tau_guess"][b])
    rep, tr = p.fit_trace()
    print("problem", b, "oracle evals", rep.n_evals, "stepped", r2[b], "vp_fit", r1[b])
    bp1 = vp.BatchProblem(vp.ExternalModel(3, 2, [(0, 0), (1, 1)]), d["Y"][b:b + 1])
    bp1.fit_begin(d["tau_guess"][b:b + 1])
    al = d["tau_guess"][b:b + 1].copy()
    mh = vp.BatchProblem(mdl, d["Y"][b:b + 1], x=d["x"])
    for it in range(40):
        ph, dp = mh.basis(al)
        al, want, nact = bp1.fit_step_with_basis(ph, dp)
        al = np.array(al)
        o = tr[it + 1] if it + 1 < len(tr) else None
        print("   eval %2d stepped next x=%s | oracle x=%s ratio %s" % (it + 1, al[0], None if o is None else o[:2], None if o is None else o[3]))
        if nact == 0:
            break
