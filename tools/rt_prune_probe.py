"""run-time-descriptor shapes: resident sets (default selection) against the streamed kernels (stream_rows=True) per shape and
length -- fit and evaluate (r, J out).  PYTHONPATH=. python tools/rt_prune_probe.py [out.json]"""
import json, sys
import numpy as np, torch
import varpro_amd as vp

dev = torch.device("cuda", 0)
rng = np.random.default_rng(11)
K = vp.basis


def f_of(kind, p, t):
    if kind == K.EXP_COS: return np.exp(-p[0][:, None] * t) * np.cos(p[1][:, None] * t)
    if kind == K.EXP_RATE: return np.exp(-p[0][:, None] * t)
    if kind == K.EXP_DECAY: return np.exp(-t / p[0][:, None])
    if kind == K.SIN_PHASE: return np.sin(p[0][:, None] * t + p[1][:, None])
    return np.ones((len(p[0]) if p else 1, len(t)))


SHAPES = {  # name: (parameter names, true values, [(kind, [names])], constant column)
    "rt<2,3,4> exp*cos pair, shared parameter": (["a1", "a2", "a3"], [1.0, 2.5, 4.0], [(K.EXP_COS, ["a2", "a3"]), (K.EXP_COS, ["a1", "a2"])], False),
    "rt<2,4,4> two exp*cos": (["a1", "a2", "a3", "a4"], [1.0, 3.0, 2.5, 7.0], [(K.EXP_COS, ["a1", "a2"]), (K.EXP_COS, ["a3", "a4"])], False),
    "rt<3,3,3> rate, rate, decay": (["a1", "a2", "a3"], [0.5, 2.0, 0.25], [(K.EXP_RATE, ["a1"]), (K.EXP_RATE, ["a2"]), (K.EXP_DECAY, ["a3"])], False),
    "rt<3,2,2> rate, rate + constant": (["a1", "a2"], [0.6, 2.5], [(K.EXP_RATE, ["a1"]), (K.EXP_RATE, ["a2"])], True),
    "rt<2,2,2> sin-phase + constant": (["a1", "a2"], [3.0, 0.4], [(K.SIN_PHASE, ["a1", "a2"])], True),
}
out = []
B = 16384
for name, (names, true, funcs, const) in SHAPES.items():
    for m in (128, 256, 512, 768, 1024):
        t = np.linspace(0.0, 1.5, m)
        par = {n: v * rng.uniform(0.95, 1.05, B) for n, v in zip(names, true)}
        Y = sum(rng.uniform(2, 8, (B, 1)) * f_of(k, [par[n] for n in ns], t) for k, ns in funcs) + (rng.uniform(1, 3, (B, 1)) if const else 0.0)
        Y = Y + 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
        g0 = np.stack([par[n] * rng.uniform(0.97, 1.03, B) for n in names], 1)
        b = vp.SeparableModelBuilder(names).initial_parameters(g0[0]).independent_variable(t)
        for k, ns in funcs:
            b = b.function(ns, k)
            for n in ns: b = b.partial_deriv(n)
        if const: b = b.invariant_function(K.CONST) if hasattr(b, "invariant_function") else b
        try:
            mdl = b.build()
            row = {"shape": name, "m": m, "B": B}
            for stream in (False, True):
                bp = vp.BatchProblem(mdl, torch.from_numpy(Y).to(dev), x=torch.from_numpy(t).to(dev), stream_rows=stream)
                bp.set_timing(True)
                g = torch.from_numpy(g0).to(dev)
                tf, te = [], []
                for _ in range(3):
                    a, c, rep = bp.fit(g, want_coefficients=False); tf.append(bp.last_kernel_ms(2))
                for _ in range(3):
                    bp.evaluate(g); te.append(bp.last_kernel_ms(0))
                r = bp.report_to_numpy(rep)
                row["streamed" if stream else "default"] = {"fit_ms": min(tf), "evaluate_ms": min(te), "evals": int(r["n_evals"].sum()), "failed": int((r["termination"] <= 0).sum())}
                bp.close()
            d, s = row["default"], row["streamed"]
            print("%-42s m=%5d: fit %7.3f / %7.3f ms (x%.2f)  evaluate %7.3f / %7.3f ms (x%.2f)  evals %d / %d  failed %d / %d" % (
                name, m, d["fit_ms"], s["fit_ms"], s["fit_ms"] / d["fit_ms"], d["evaluate_ms"], s["evaluate_ms"], s["evaluate_ms"] / d["evaluate_ms"],
                d["evals"], s["evals"], d["failed"], s["failed"]), flush=True)
            out.append(row)
        except Exception as e:
            print(name, m, repr(e)[:300], flush=True)
            break
if len(sys.argv) > 1: json.dump(out, open(sys.argv[1], "w"), indent=1)
