import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth
B=65536
d = synth.double_exp_batch(B, m=1024, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
dev = torch.device("cuda", 0)
bp = vp.BatchProblem(mdl, torch.from_numpy(d["Y"]).to(dev), x=torch.from_numpy(d["x"]).to(dev))
a,c,rep = bp.fit(torch.from_numpy(d["tau_guess"]).to(dev)); r = bp.report_to_numpy(rep)
np.save("/root/repo/gpurun_out/nfev_b65536.npy", r["n_evals"].astype(np.int32))
print("saved", r["n_evals"].sum())
