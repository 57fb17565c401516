"""A/B timing of library builds on ONE box: python tools/ab_probe.py libA.so libB.so [B]  (each in a subprocess,
alternating, 3 rounds) -- kernel times from the library's own HIP events."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib
B = int(sys.argv[1])
d = synth.double_exp_batch(B, m=1024, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda", 0)
Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=x); bp.set_timing(True)
out = []
for wr, wj in ((False, False), (True, False), (True, True)):
    ts = []
    for _ in range(6):
        bp.evaluate(g, want_residuals=wr, want_jacobian=wj); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_EVALUATE))
    out.append(min(ts))
ts = []
for _ in range(8):
    a, c, rep = bp.fit(g, want_coefficients=False); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
r = bp.report_to_numpy(rep)
_ph = torch.empty((B, 2, 1024), dtype=torch.float64, device=dev); _dp = torch.empty((B, 2, 1024), dtype=torch.float64, device=dev)
tb = []
for _ in range(6):
    bp.basis(g, skip_invariant=True, out_phi=_ph, out_dphi=_dp); tb.append(bp.last_kernel_ms(_lib.VP_KERNEL_BASIS))
print("ev0 %%.3f ev1 %%.3f ev2 %%.3f basis %%.3f fit %%.3f ms (median %%.3f)  evals %%d  cost %%.9e" %% (out[0], out[1], out[2], min(tb), min(ts), sorted(ts)[len(ts)//2], r["n_evals"].sum(), np.nansum(r["objective"])))
''' % ROOT
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
B = rest[0] if rest else "65536"
envs = [a for a in rest[1:] if "=" in a] or [""]   # e.g. VP_FIT_CAP=0 VP_FIT_CAP=16 : each lib x each env
for rnd in range(3):
  for ev in envs:
    for lib in libs:
        env = dict(os.environ, VARPRO_HIP_LIBRARY=lib)
        if ev: env[ev.split("=")[0]] = ev.split("=")[1]
        o = subprocess.run([sys.executable, "-c", CHILD, B], env=env, capture_output=True, text=True)
        print("%-30s %-14s %s" % (os.path.basename(lib), ev, (o.stdout.strip().split("\n") or [""])[-1] or o.stderr[-300:]))
