"""evaluation counts of the resident and the streamed kernels of run-time-descriptor shapes against the oracle's, problem by problem.
PYTHONPATH=. python tools/rt_evals_probe.py"""
import numpy as np
import varpro_amd as vp
from oracle import oracle as O
from oracle import census as CS
K = vp.basis
rng = np.random.default_rng(11)
B = 2048
for name, names, true, funcs, const in (("rt<3,2,2>", ["a1", "a2"], [0.6, 2.5], [(K.EXP_RATE, ["a1"]), (K.EXP_RATE, ["a2"])], True),
                                        ("rt<2,3,4>", ["a1", "a2", "a3"], [1.0, 2.5, 4.0], [(K.EXP_COS, ["a2", "a3"]), (K.EXP_COS, ["a1", "a2"])], False)):
    for m in (256, 1024):
        t = np.linspace(0.0, 1.5, m)
        par = {n: v * rng.uniform(0.95, 1.05, B) for n, v in zip(names, true)}
        def f_of(k, p):
            if k == K.EXP_COS: return np.exp(-p[0][:, None] * t) * np.cos(p[1][:, None] * t)
            return np.exp(-p[0][:, None] * t)
        Y = sum(rng.uniform(2, 8, (B, 1)) * f_of(k, [par[n] for n in ns]) for k, ns in funcs) + (rng.uniform(1, 3, (B, 1)) if const else 0.0)
        Y = Y + 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
        g0 = np.stack([par[n] * rng.uniform(0.97, 1.03, B) for n in names], 1)
        b = vp.SeparableModelBuilder(names).initial_parameters(g0[0]).independent_variable(t)
        for k, ns in funcs:
            b = b.function(ns, k)
            for n in ns: b = b.partial_deriv(n)
        if const: b = b.invariant_function(K.CONST)
        mdl = b.build()
        ao, _c, ro, _s = O.fit_batch(mdl, t, Y, g0, n_threads=8)
        for stream in (False, True):
            for which in (("auto", "wave") if not stream else ("auto",)):
                bp = vp.BatchProblem(mdl, Y, x=t, stream_rows=stream)
                bp.set_fit_kernel(which)
                a, _C, rep = bp.fit(g0)
                r = bp.report_to_numpy(rep)
                res = CS.census(rep, a, ro, ao, max_listed=0)
                print("%s m=%d %-8s %-5s: evals %d (oracle %d)  within 3: %.3f  equal: %.3f  same class %.4f  objective median %.1e max %.1e" % (
                    name, m, "streamed" if stream else "resident", which, r["n_evals"].sum(), ro["n_evals"].sum(), res["share_evals_within_3"], res["share_evals_equal"],
                    res["same_success_class"], res["objective_rel_diff_median_common_successes"], res["objective_rel_diff_max_common_successes"]), flush=True)
                bp.close()
