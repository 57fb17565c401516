"""long caller-evaluated problems (bench.py external_model.long_problems f64_m10000): time per launch of vp_evaluate_with_basis as a
function of the number of resident waves (capped through the kernel's LDS footprint, VP_EXT_STREAM_LDS_KB) -- does the backward
pass's re-read of the columns come out of the 256 MB Infinity Cache when fewer problems are in flight?
usage: VP_EXT_STREAM_LDS_KB=<kb> python tools/ext_stream_cap_probe.py [m] [B]"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
import varpro_amd as vp
from varpro_amd import _lib
ml = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
Bl = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
xl = torch.linspace(0.0, 12.5, ml, dtype=torch.float64, device=dev)
t1 = 0.5 + 1.5 * torch.rand((Bl, 1), generator=g, dtype=torch.float64, device=dev)
t2 = 2.5 + 5.5 * torch.rand((Bl, 1), generator=g, dtype=torch.float64, device=dev)
cl = 1.0 + 99.0 * torch.rand((Bl, 3), generator=g, dtype=torch.float64, device=dev)
phl = torch.empty((Bl, 3, ml), dtype=torch.float64, device=dev); dpl = torch.empty((Bl, 2, ml), dtype=torch.float64, device=dev)
e1, e2 = torch.exp(-xl[None] / t1), torch.exp(-xl[None] / t2)
phl[:, 0], phl[:, 1], phl[:, 2] = e1, e2, 1.0
dpl[:, 0], dpl[:, 1] = e1 * xl[None] / (t1 * t1), e2 * xl[None] / (t2 * t2)
Yl = cl[:, 0:1] * e1 + cl[:, 1:2] * e2 + cl[:, 2:3]
al = torch.cat([t1, t2], 1) * 1.1
bp = vp.BatchProblem(vp.ExternalModel(3, 2, [(0, 0), (1, 1)]), Yl)
rl, Jl = torch.empty((Bl, ml), dtype=torch.float64, device=dev), torch.empty((Bl, 2, ml), dtype=torch.float64, device=dev)
Cl, costl, stl = torch.empty((Bl, 3), dtype=torch.float64, device=dev), torch.empty((Bl,), dtype=torch.float64, device=dev), torch.empty((Bl,), dtype=torch.int32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
def call():
    _lib.check(bp.lib.vp_evaluate_with_basis(bp._h, p(al), p(phl), p(dpl), p(rl), p(Jl), p(Cl), p(costl), p(stl)))
for _ in range(3): call()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
for a_, b_ in ev:
    a_.record(); call(); b_.record()
torch.cuda.synchronize()
ms = sorted(a_.elapsed_time(b_) for a_, b_ in ev)[len(ev) // 2]
by = Bl * 8 * ml * 9
print("LDS_KB=%s m=%d B=%d: %.3f ms  %.0f GB/s algorithmic  frac %.3f  ok %.3f" % (os.environ.get("VP_EXT_STREAM_LDS_KB", "-"), ml, Bl, ms, by / ms / 1e6, by / ms / 1e6 / 8000, float((stl == 0).double().mean())))
