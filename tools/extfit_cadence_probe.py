"""The bench's `external_fit` leg (65 536 headline problems as a caller-evaluated model, columns by vp_basis of a descriptor handle
while most problems run, a torch expression over the compacted active set in the tail) with the CALLER's two free choices varied:
how often it reads the active count back (a host synchronisation) and how it writes the tail expression.
usage: PYTHONPATH=. python tools/extfit_cadence_probe.py"""
import time
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth
dev = torch.device("cuda", 0)
B, m = 65536, 1024
d = synth.double_exp_batch(B, m=m, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev)
guess = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=x)
bpx = vp.BatchProblem(vp.ExternalModel(3, 2, [(0, 0), (1, 1)]), Y)
phi_x = torch.ones((B, 3, m), dtype=torch.float64, device=dev)
dphi_x = torch.empty((B, 2, m), dtype=torch.float64, device=dev)
xt_row = x[None, None, :]
act_idx = torch.empty((B,), dtype=torch.int32, device=dev); act_cnt = torch.empty((1,), dtype=torch.int32, device=dev)

def tail_a(alpha, n_active):
    bpx.fit_active_set(act_idx, act_cnt)
    idx = act_idx[:n_active].long()
    a_ = alpha[idx][:, :, None]
    e_ = torch.exp(-xt_row / a_)
    phi_x[idx, 0:2] = e_
    dphi_x[idx] = e_ * xt_row / (a_ * a_)

def tail_b(alpha, n_active):  # fewer launches: one reciprocal, fused scale factors
    bpx.fit_active_set(act_idx, act_cnt)
    idx = act_idx[:n_active].long()
    nr = torch.reciprocal(alpha[idx]).neg_()[:, :, None]
    e_ = torch.exp(xt_row * nr)
    phi_x[idx, 0:2] = e_
    dphi_x[idx] = e_.mul_(xt_row).mul_(nr * nr)

def stepped(K, small_every, tail):
    bpx.fit_begin(guess)
    alpha_, want_, nact, steps_ = guess, None, B, 0
    while nact > 0 and steps_ < 400:
        if want_ is None or nact * 4 >= B: bp.basis(alpha_, skip_invariant=False, out_phi=phi_x, out_dphi=dphi_x)
        else: tail(alpha_, nact)
        look_ = (steps_ + 1) % K == 0 or (nact < 64 and (steps_ + 1) % small_every == 0)
        alpha_, want_, na_ = bpx.fit_step_with_basis(phi_x, dphi_x, want_count=look_)
        nact = na_ if look_ else nact
        steps_ += 1
    return bpx.fit_end(want_coefficients=False) + (steps_,)

for K, se, tail, name in ((4, 1, tail_a, "bench as is"), (4, 4, tail_a, "count every 4th step throughout"), (8, 8, tail_a, "count every 8th step"),
                          (4, 1, tail_b, "slimmer tail expression"), (4, 4, tail_b, "both"), (8, 8, tail_b, "both, every 8th")):
    stepped(K, se, tail); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); a, c, rep, st = stepped(K, se, tail); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    r = bpx.report_to_numpy(rep)
    print("%-36s %7.2f ms = %.3f M fits/s  steps %d  evals %d  failed %d" % (name, min(ts) * 1e3, B / min(ts) / 1e6, st, int(r["n_evals"].sum()), int((r["termination"] <= 0).sum())))
