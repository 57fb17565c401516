"""Model constructors shared by the tests (mirrors of the reference's test models)."""
import numpy as np

import varpro_amd as vp
from varpro_amd import basis


def double_exp_builder_model(x, initial):
    """shared_test_code/src/lib.rs:119-135: columns [exp(-x/tau1), exp(-x/tau2), 1]"""
    return (vp.SeparableModelBuilder(["tau1", "tau2"]).initial_parameters(initial)
            .function(["tau1"], basis.EXP_DECAY).partial_deriv("tau1")
            .function(["tau2"], basis.EXP_DECAY).partial_deriv("tau2")
            .invariant_function(basis.CONST).independent_variable(x).build())


def double_exp_unit_test_model(x, initial):
    """src/test_helpers/mod.rs:56-72: NOTE the different column order [exp(-x/tau2), exp(-x/tau1), 1]"""
    return (vp.SeparableModelBuilder(["tau1", "tau2"])
            .function(["tau2"], basis.EXP_DECAY).partial_deriv("tau2")
            .function(["tau1"], basis.EXP_DECAY).partial_deriv("tau1")
            .invariant_function(basis.CONST).independent_variable(x).initial_parameters(initial).build())


def oleary_model(t, initial):
    """shared_test_code/src/models.rs:397-425: phi1 = exp(-a2 t) cos(a3 t), phi2 = exp(-a1 t) cos(a2 t)"""
    return (vp.SeparableModelBuilder(["alpha1", "alpha2", "alpha3"]).initial_parameters(initial)
            .independent_variable(t)
            .function(["alpha2", "alpha3"], basis.EXP_COS).partial_deriv("alpha2").partial_deriv("alpha3")
            .function(["alpha1", "alpha2"], basis.EXP_COS).partial_deriv("alpha1").partial_deriv("alpha2")
            .build())


def numpy_reference_eval(model, x, y, alpha, w=None):
    """independent numpy restatement of src/solvers/levmar/mod.rs:42-73,101-201 for ONE single-RHS problem
    (numpy.linalg.svd + the reference's formulas): returns c, r, J (q, m)"""
    kinds, pidx = model.kinds, model.param_indices
    n, q, m = len(kinds), model.n_params, x.size
    W = np.ones(m) if w is None else np.asarray(w, dtype=float)

    def f(kind, t, p):
        if kind == 0:
            return np.ones_like(t), []
        if kind == 1:
            e = np.exp(-t / p[0])
            return e, [e * t / (p[0] * p[0])]
        if kind == 2:
            e = np.exp(-p[0] * t)
            return e, [-t * e]
        if kind == 3:
            ex = np.exp(-p[0] * t)
            return ex * np.cos(p[1] * t), [-t * ex * np.cos(p[1] * t), -t * ex * np.sin(p[1] * t)]
        ph = p[0] * t + p[1]
        return np.sin(ph), [t * np.cos(ph), np.cos(ph)]

    Phi = np.zeros((m, n))
    D = [np.zeros((m, n)) for _ in range(q)]
    for j in range(n):
        p = [alpha[i] for i in pidx[j]]
        v, dv = f(kinds[j], x, p)
        Phi[:, j] = v
        for a, i in enumerate(pidx[j]):
            D[i][:, j] += dv[a]
    Pw = Phi * W[:, None]
    yw = y * W
    U, s, Vt = np.linalg.svd(Pw, full_matrices=False)
    c = Vt.T @ ((U.T @ yw) / s)
    r = yw - Pw @ c
    J = np.zeros((q, m))
    for k in range(q):
        T = (D[k] * W[:, None]) @ c
        J[k] = U @ (U.T @ T) - T
    return c, r, J
