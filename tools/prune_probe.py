"""which resident kernel sets does the streamed kernel (vp_block.hpp) now match?  For every multi-exponential family and a
length inside every in-between resident set: fit and evaluate (r, J out) time of the default selection against the streamed
kernels (stream_rows=True) on the same problems.  PYTHONPATH=. python tools/prune_probe.py [out.json]"""
import json, sys
import numpy as np
import torch

import varpro_amd as vp

dev = torch.device("cuda:0")
rng = np.random.default_rng(7)
out = []


def problems(nexp, m, B):
    base = {1: [2.0], 2: [1.0, 4.0], 3: [0.7, 2.0, 6.0]}[nexp]
    x = np.linspace(0, 12.5, m)
    tau = np.stack([rng.uniform(0.9, 1.1, B) * t0 for t0 in base], 1)
    c = rng.uniform(5, 50, (B, nexp + 1))
    Y = sum(c[:, j:j + 1] * np.exp(-x / tau[:, j:j + 1]) for j in range(nexp)) + c[:, nexp:nexp + 1]
    Y += 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    return x, Y, tau * rng.uniform(0.9, 1.1, tau.shape)


def run(nexp, m, B, dtype):
    x, Y, g = problems(nexp, m, B)
    Yd, xd, gd = (torch.from_numpy(a.astype(dtype)).to(dev) for a in (Y, x, g))
    mdl = vp.multi_exponential_model(x.astype(dtype), g[0].astype(dtype))
    row = {"nexp": nexp, "m": m, "B": B, "dtype": np.dtype(dtype).name}
    for stream in (False, True):
        bp = vp.BatchProblem(mdl, Yd, x=xd, stream_rows=stream)
        bp.set_timing(True)
        tf, te = [], []
        for _ in range(3):
            a, c, rep = bp.fit(gd, want_coefficients=False)
            tf.append(bp.last_kernel_ms(2))
        for _ in range(3):
            bp.evaluate(gd)
            te.append(bp.last_kernel_ms(0))
        r = bp.report_to_numpy(rep)
        k = "streamed" if stream else "default"
        row[k] = {"fit_ms": min(tf), "evaluate_ms": min(te), "evals": int(r["n_evals"].sum()), "failed": int((r["termination"] <= 0).sum())}
        bp.close()
    row["fit_ratio_streamed_over_default"] = row["streamed"]["fit_ms"] / row["default"]["fit_ms"]
    row["evaluate_ratio_streamed_over_default"] = row["streamed"]["evaluate_ms"] / max(row["default"]["evaluate_ms"], 1e-9)
    print("me%d %s m=%5d B=%d: fit %7.3f / %7.3f ms (x%.2f)   evaluate %7.3f / %7.3f ms (x%.2f)   evals %d / %d" % (
        nexp, row["dtype"], m, B, row["default"]["fit_ms"], row["streamed"]["fit_ms"], row["fit_ratio_streamed_over_default"],
        row["default"]["evaluate_ms"], row["streamed"]["evaluate_ms"], row["evaluate_ratio_streamed_over_default"],
        row["default"]["evals"], row["streamed"]["evals"]), flush=True)
    out.append(row)


B = 16384
for nexp in (1, 2, 3):
    for m in (700, 1024, 1150, 1400, 1700, 2000, 3000, 4096, 6000):
        try:
            run(nexp, m, B, np.float64)
        except Exception as e:  # a probe: report and go on
            print("me%d m=%d: %r" % (nexp, m, e), flush=True)
for m in (1024, 1500, 2048, 4096):
    try:
        run(2, m, B, np.float32)
    except Exception as e:
        print("me2 f32 m=%d: %r" % (m, e), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
