import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth
B=65536
d = synth.double_exp_batch(B, m=1024, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda", 0)
Y = torch.from_numpy(d["Y"]).to(dev); x = torch.from_numpy(d["x"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=x)
a,c,rep = bp.fit(g); r = bp.report_to_numpy(rep)
ne = r["n_evals"]
print("n_evals mean %.2f  percentiles 50/90/99/99.9/max:" % ne.mean(), np.percentile(ne,[50,90,99,99.9]), ne.max())
print("hist:", np.bincount(np.minimum(ne, 200)//10))
print("termination counts:", np.unique(r["termination"], return_counts=True))
