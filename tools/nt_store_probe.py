"""Streaming-output kernels of a library build, 25 launches each (min / median ms): vp_basis (Phi, dPhi out), vp_evaluate (r, J out),
vp_residuals-only, vp_evaluate_with_basis (Phi, dPhi in; r, J out) at the headline shape -- the A/B of VP_NT_STORES.
usage: VARPRO_HIP_LIBRARY=lib.so PYTHONPATH=. python tools/nt_store_probe.py"""
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib
B, m = 65536, 1024
d = synth.double_exp_batch(B, m=m, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda", 0)
Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=x)
def timed(fn, n=25):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[3:]); return ts[0], ts[len(ts) // 2]
ph = torch.empty((B, 2, m), dtype=torch.float64, device=dev); dp = torch.empty((B, 2, m), dtype=torch.float64, device=dev)
print("basis            min %.4f median %.4f ms" % timed(lambda: bp.basis(g, skip_invariant=True, out_phi=ph, out_dphi=dp)))
del ph, dp
r = torch.empty((B, m), dtype=torch.float64, device=dev); J = torch.empty((B, 2, m), dtype=torch.float64, device=dev)
C = torch.empty((B, 3), dtype=torch.float64, device=dev); cost = torch.empty((B,), dtype=torch.float64, device=dev); st = torch.empty((B,), dtype=torch.int32, device=dev)
vptr = lambda t: None if t is None else t.data_ptr()
print("evaluate r+J     min %.4f median %.4f ms" % timed(lambda: _lib.check(bp.lib.vp_evaluate(bp._h, vptr(g), vptr(r), vptr(J), vptr(C), vptr(cost), vptr(st)))))
print("evaluate r       min %.4f median %.4f ms" % timed(lambda: _lib.check(bp.lib.vp_evaluate(bp._h, vptr(g), vptr(r), None, vptr(C), vptr(cost), vptr(st)))))
phi = torch.ones((B, 3, m), dtype=torch.float64, device=dev); dphi = torch.empty((B, 2, m), dtype=torch.float64, device=dev)
bp.basis(g, skip_invariant=False, out_phi=phi, out_dphi=dphi)
bx = vp.BatchProblem(vp.ExternalModel(3, 2, [(0, 0), (1, 1)]), Y)
print("ext evaluate r+J min %.4f median %.4f ms" % timed(lambda: _lib.check(bx.lib.vp_evaluate_with_basis(bx._h, vptr(g), vptr(phi), vptr(dphi), vptr(r), vptr(J), vptr(C), vptr(cost), vptr(st)))))
