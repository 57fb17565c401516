"""per-step wall clock of the headline problems fitted as a caller-evaluated model (bench.py `external_fit`): where do the ~26 ms go?
usage (GPU box): python tools/extfit_timeline.py [check_every]"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
import varpro_amd as vp
from varpro_amd import synth
B, m = 65536, 1024
check_every = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
d = synth.double_exp_batch(B, m=m, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); guess = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=x)
bpx = vp.BatchProblem(vp.ExternalModel(3, 2, [(0, 0), (1, 1)]), Y)
phi_x = torch.empty((B, 3, m), dtype=torch.float64, device=dev); dphi_x = torch.empty((B, 2, m), dtype=torch.float64, device=dev)
bp.basis(guess, skip_invariant=False, out_phi=phi_x, out_dphi=dphi_x)
act_idx = torch.empty((B,), dtype=torch.int32, device=dev); act_cnt = torch.empty((1,), dtype=torch.int32, device=dev)
xt_row = x[None, None, :]
def caller_model(alpha, first, n_active):
    if first or n_active * 4 >= B:
        bp.basis(alpha, skip_invariant=False, out_phi=phi_x, out_dphi=dphi_x)
    else:
        bpx.fit_active_set(act_idx, act_cnt)
        idx = act_idx[:n_active].long()
        a_ = alpha[idx][:, :, None]
        e_ = torch.exp(-xt_row / a_)
        phi_x[idx, 0:2] = e_
        dphi_x[idx] = e_ * xt_row / (a_ * a_)
def run(record):
    bpx.fit_begin(guess)
    alpha, nact, steps = guess, B, 0
    torch.cuda.synchronize(); t0 = time.perf_counter(); tl = []
    while nact > 0 and steps < 400:
        ta = time.perf_counter()
        caller_model(alpha, steps == 0, nact)
        if record: torch.cuda.synchronize()
        tb = time.perf_counter()
        look = (steps + 1) % check_every == 0 or nact < 64
        alpha, want, na = bpx.fit_step_with_basis(phi_x, dphi_x, want_count=look)
        if look: nact = na
        if record: torch.cuda.synchronize()
        tc = time.perf_counter()
        tl.append((steps, nact, (tb - ta) * 1e3, (tc - tb) * 1e3))
        steps += 1
    bpx.fit_end(want_coefficients=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, steps, tl
run(False)
tot, steps, _ = run(False)
print("total ms %.2f steps %d -> %.2f M fits/s (check_every %d)" % (tot, steps, B / tot / 1e3, check_every))
tot2, steps2, tl = run(True)
print("with per-phase syncs: %.2f ms" % tot2)
for s_, n_, a_, b_ in tl[:16] + tl[16::8]:
    print("step %3d active %6d caller %.3f ms step %.3f ms" % (s_, n_, a_, b_))
print("sum caller %.2f sum step %.2f" % (sum(t[2] for t in tl), sum(t[3] for t in tl)))
