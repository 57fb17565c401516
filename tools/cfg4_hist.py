"""cfg4: distribution of evaluations per fit, terminations, and launch time (for the tail analysis)"""
import sys, os, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = synth.multi_exp_batch(B, 5, 4096, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
dev = torch.device("cuda", 0)
Y = torch.from_numpy(d["Y"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=torch.from_numpy(d["x"]).to(dev))
bp.set_timing(True)
ts = []
for _ in range(3):
    a, C, rep = bp.fit(g); ts.append(bp.last_kernel_ms(_lib.VP_KERNEL_FIT))
r = bp.report_to_numpy(rep)
ne = r["n_evals"]
out = dict(B=B, ms=min(ts), mean=float(ne.mean()), max=int(ne.max()),
           pct={str(p): float(np.percentile(ne, p)) for p in (50, 90, 99, 99.9)},
           hist=np.bincount(np.minimum(ne, 700) // 10).tolist(),
           term=dict(collections.Counter(int(x) for x in r["termination"])))
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/nfev_cfg4.npy", ne)
