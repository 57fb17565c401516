"""vp_evaluate (r and J out) of descriptor models beyond the resident sets (blk_evaluate_kernel, vp_block.hpp): min / median ms of 15
launches.  usage: VARPRO_HIP_LIBRARY=lib.so PYTHONPATH=. python tools/blk_eval_probe.py"""
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib
dev = torch.device("cuda", 0)
for B, m in ((8192, 10000), (1024, 100000)):
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    bp = vp.BatchProblem(mdl, torch.from_numpy(d["Y"]).to(dev), x=torch.from_numpy(d["x"]).to(dev))
    g = torch.from_numpy(d["tau_guess"]).to(dev)
    r = torch.empty((B, m), dtype=torch.float64, device=dev); J = torch.empty((B, 2, m), dtype=torch.float64, device=dev)
    C = torch.empty((B, 3), dtype=torch.float64, device=dev); cost = torch.empty((B,), dtype=torch.float64, device=dev); st = torch.empty((B,), dtype=torch.int32, device=dev)
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(bp.lib.vp_evaluate(bp._h, g.data_ptr(), r.data_ptr(), J.data_ptr(), C.data_ptr(), cost.data_ptr(), st.data_ptr())); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[3:]); by = 8.0 * B * m * 4
    print("B=%d m=%d: min %.4f median %.4f ms -> %.3f of 8 TB/s (y in, r + 2 J out)" % (B, m, ts[0], ts[len(ts) // 2], by / ts[len(ts) // 2] / 8e9))
    bp.close(); del r, J
