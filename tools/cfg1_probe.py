import sys; sys.path.insert(0, ".")
import numpy as np, torch, varpro_amd as vp
from varpro_amd import synth
dev=torch.device("cuda:0")
for B in (1024, 4096):
    d=synth.double_exp_batch(B,m=1024,noise=1e-3)
    mdl=vp.multi_exponential_model(d["x"],d["tau_guess"][0])
    Y=torch.from_numpy(d["Y"]).to(dev); x=torch.from_numpy(d["x"]).to(dev); g=torch.from_numpy(d["tau_guess"]).to(dev)
    for stream in (False, True):
        bp=vp.BatchProblem(mdl,Y,x=x,stream_rows=stream); bp.set_timing(True)
        ts=[]
        for _ in range(5):
            a,c,rep=bp.fit(g,want_coefficients=False); ts.append(bp.last_kernel_ms(2))
        r=bp.report_to_numpy(rep)
        print("B=%d %s %.3f ms %.2f M fits/s max evals %d" % (B, "streamed" if stream else "resident", min(ts), B/min(ts)/1e3, r["n_evals"].max()))
        bp.close()
