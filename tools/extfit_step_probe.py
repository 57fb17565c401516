"""Times ONE step of the batched external fit (ext_fit_step_kernel, every problem active: the first step after vp_fit_begin)
for a few shapes: ms per launch (HIP events around the launch) and the input stream in GB/s against the HBM peak.
usage: python tools/extfit_step_probe.py [B]   (VARPRO_HIP_LIBRARY selects an A/B build)"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import varpro_amd as vp  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for (n, p, q, m, dt) in ((3, 2, 2, 1024, np.float64), (3, 4, 4, 1024, np.float64), (3, 4, 4, 512, np.float64), (3, 4, 4, 256, np.float64),
                         (2, 2, 2, 1024, np.float64), (3, 2, 2, 1024, np.float32), (3, 4, 4, 1024, np.float32)):
    tdt = torch.float64 if dt == np.float64 else torch.float32
    Bx = B if m * (n + p + 1) * B * (8 if dt == np.float64 else 4) < 24e9 else B // 2
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.linspace(0.0, 1.0, m, device=dev, dtype=tdt)
    Phi = torch.stack([torch.cos((j + 1) * 3.0 * x) for j in range(n)], 0)[None].repeat(Bx, 1, 1).contiguous()
    Phi += 0.01 * torch.randn(Phi.shape, device=dev, dtype=tdt, generator=g)
    dPhi = torch.randn((Bx, p, m), device=dev, dtype=tdt, generator=g)
    Y = Phi.sum(1) + 0.1 * torch.randn((Bx, m), device=dev, dtype=tdt, generator=g)
    pairs = [(j % n, j % q) for j in range(p)] if p != q else [(j % n, j) for j in range(q)]
    bp = vp.BatchProblem(vp.ExternalModel(n, q, pairs, dtype=dt), Y)
    alpha = torch.ones((Bx, q), device=dev, dtype=tdt)
    ms = []
    for rep in range(6):
        bp.fit_begin(alpha)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        bp.fit_step_with_basis(Phi, dPhi, want_count=False)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    t = float(np.median(ms[1:]))
    nbytes = Bx * m * (n + p + 1) * (8 if dt == np.float64 else 4)
    print("n=%d p=%d q=%d m=%4d %s B=%6d  %.3f ms  %.0f GB/s  frac %.3f" % (n, p, q, m, dt.__name__, Bx, t, nbytes / t / 1e6, nbytes / t / 1e6 / 8000.0))
    bp.close()
    del Phi, dPhi, Y
