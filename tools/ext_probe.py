"""Caller-evaluated models (vp_*_with_basis): time the evaluation at B problems of m rows on device-resident columns.
usage: python tools/ext_probe.py [B] [m]"""
import sys
import time

import numpy as np
import torch

import varpro_amd as vp

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
for (n, q, pairs, name) in ((3, 2, [(0, 0), (1, 1)], "double-exp shape n=3 q=2 p=2"),
                            (3, 4, [(0, 0), (0, 1), (1, 2), (1, 3)], "gauss+lorentz shape n=3 q=4 p=4")):
    p = len(pairs)
    g = torch.Generator(device=dev).manual_seed(1)
    Phi = torch.rand((B, n, m), dtype=torch.float64, device=dev, generator=g)
    dPhi = torch.rand((B, p, m), dtype=torch.float64, device=dev, generator=g)
    Y = torch.rand((B, m), dtype=torch.float64, device=dev, generator=g)
    alpha = torch.rand((B, q), dtype=torch.float64, device=dev, generator=g)
    bp = vp.BatchProblem(vp.ExternalModel(n, q, pairs), Y)
    lib, h = bp.lib, bp._h
    import ctypes as C
    r = torch.empty((B, m), dtype=torch.float64, device=dev)
    J = torch.empty((B, q, m), dtype=torch.float64, device=dev)
    Cc = torch.empty((B, n), dtype=torch.float64, device=dev)
    cost = torch.empty((B,), dtype=torch.float64, device=dev)
    st = torch.empty((B,), dtype=torch.int32, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_sp = timeit(lambda: lib.vp_set_params_with_basis(h, P(alpha), P(Phi), None))
    t_c = timeit(lambda: lib.vp_evaluate_with_basis(h, P(alpha), P(Phi), None, None, None, P(Cc), P(cost), P(st)))
    t_rj = timeit(lambda: lib.vp_evaluate_with_basis(h, P(alpha), P(Phi), P(dPhi), P(r), P(J), P(Cc), P(cost), P(st)))
    b_in = 8 * m * (n + 1)
    print("%s  B=%d m=%d" % (name, B, m))
    print("  set_params_with_basis (Phi, y in; r cached):  %.3f ms  %.2f TB/s of %d B" % (t_sp, B * (b_in + 8 * m) / t_sp / 1e9, b_in + 8 * m))
    print("  evaluate (Phi, y in; c, cost out):            %.3f ms  %.2f TB/s of %d B" % (t_c, B * b_in / t_c / 1e9, b_in))
    full = 8 * m * (n + p + 1) + 8 * m * (1 + q)
    print("  evaluate (Phi, dPhi, y in; r, J out):         %.3f ms  %.2f TB/s of %d B (in %d)" % (t_rj, B * full / t_rj / 1e9, full, 8 * m * (n + p + 1)))
    bp.close()
