"""Between the 16-rows-per-lane sets and 2048 rows: the register-resident sets (12 .. 32 rows per lane, some of them spilling at
512 VGPRs) against the streamed kernels (vp_block.hpp) on the same problems.  PYTHONPATH=. python tools/stream_vs_resident.py"""
import sys
import numpy as np
import torch

import varpro_amd as vp

dev = torch.device("cuda:0")
rng = np.random.default_rng(1)


def run(name, mdl, Y, x, g, B):
    Yd, xd, gd = torch.from_numpy(Y).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(g).to(dev)
    res = {}
    for stream in (False, True):
        bp = vp.BatchProblem(mdl, Yd, x=xd, stream_rows=stream)
        bp.set_timing(True)
        ts = []
        for _ in range(3):
            a, c, rep = bp.fit(gd, want_coefficients=False)
            ts.append(bp.last_kernel_ms(2))
        te = []
        for _ in range(3):
            bp.evaluate(gd, want_residuals=True, want_jacobian=True)
            te.append(bp.last_kernel_ms(1))
        r = bp.report_to_numpy(rep)
        res[stream] = (B / min(ts) / 1e3, min(te), r["n_evals"].mean(), int((r["termination"] <= 0).sum()))
        bp.close()
    print("%-22s resident %7.2f M fits/s (eval %.3f ms)  streamed %7.2f M fits/s (eval %.3f ms)  ratio %.2f  evals %.2f / %.2f failed %d / %d" % (
        name, res[False][0], res[False][1], res[True][0], res[True][1], res[True][0] / res[False][0], res[False][2], res[True][2],
        res[False][3], res[True][3]))


B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for nexp, base, ms in ((2, [1.0, 4.0], (600, 768, 1100, 1280, 1536, 1792, 2048)), (3, [0.7, 2.0, 6.0], (512, 768, 1024, 1280, 1536, 2048)),
                       (1, [2.0], (1536, 2048))):
    for m in ms:
        x = np.linspace(0, 12.5, m)
        tau = np.stack([rng.uniform(0.9, 1.1, B) * t0 for t0 in base], 1)
        c = rng.uniform(5, 50, (B, nexp + 1))
        Y = sum(c[:, j:j + 1] * np.exp(-x / tau[:, j:j + 1]) for j in range(nexp)) + c[:, nexp:nexp + 1]
        Y += 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
        g = tau * rng.uniform(0.9, 1.1, tau.shape)
        run("me%d+offset m=%d" % (nexp, m), vp.multi_exponential_model(x, g[0]), Y, x, g, B)
