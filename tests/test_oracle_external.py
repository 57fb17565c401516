"""CPU tier: the oracle's external-model hook (vpo_problem_set_external_model -- the reference's trait calls
`model.eval()` / `model.eval_partial_deriv(k)`, /root/reference/src/solvers/levmar/mod.rs:45, :141, on ANY
`SeparableNonlinearModel`, /root/reference/src/model/mod.rs:239-363) pinned against an independent numpy restatement
(numpy.linalg.svd + the reference's formulas) for a model outside the descriptor language, and the C ABI's refusal /
validation paths that need no device."""
import numpy as np

import varpro_amd as vp
from test_gpu_external import oracle_problem, peaks_data, peaks_model, voigt_model


def _numpy_eval(cm, y, alpha, w):
    m = cm.x.size
    W = np.ones(m) if w is None else w
    Phi = cm.eval_batch(alpha[None])[0].T * W[:, None]
    d = cm.derivs_batch(alpha[None])[0]
    U, s, Vt = np.linalg.svd(Phi, full_matrices=False)
    yw = y * W
    c = Vt.T @ ((U.T @ yw) / s)
    r = yw - Phi @ c
    sh = cm.shape()
    J = np.zeros((sh.n_params, m))
    for k in range(sh.n_params):
        D = np.zeros((m, sh.n_basis))
        for p, (j, kk) in enumerate(cm.pairs()):
            if kk == k:
                D[:, j] += d[p] * W
        T = D @ c
        J[k] = U @ (U.T @ T) - T
    return c, r, J


def test_oracle_with_callbacks_matches_numpy_for_a_gauss_lorentz_model():
    rng = np.random.default_rng(3)
    for m, weighted in ((50, False), (400, True)):
        x = np.linspace(0.0, 10.0, m)
        cm = peaks_model(x)
        _t, _c, Y, guess = peaks_data(rng, 2, x)
        w = (0.5 + rng.random(m)) if weighted else None
        for b in range(2):
            p = oracle_problem(cm, Y[b], w=w)
            p.set_params(guess[b])
            c, r, J = _numpy_eval(cm, Y[b], guess[b], w)
            assert np.abs(p.linear_coefficients() - c).max() <= 1e-12 * np.abs(c).max()
            assert np.abs(p.residuals() - r).max() <= 1e-12 * np.abs(Y[b]).max()
            for k in range(4):
                assert np.abs(p.jacobian()[k] - J[k]).max() <= 1e-11 * np.abs(J[k]).max()


def test_oracle_fits_models_outside_the_descriptor_language():
    rng = np.random.default_rng(5)
    x = np.linspace(0.0, 10.0, 300)
    cm = peaks_model(x)
    truth, _c, Y, guess = peaks_data(rng, 3, x, noise=0.0)
    for b in range(3):
        p = oracle_problem(cm, Y[b])
        p.set_params(guess[b])
        rep = p.fit()
        assert rep.termination > 0
        assert np.abs(p.params() - truth[b]).max() <= 1e-6
    # three parameters in one basis function
    cm = voigt_model(x)
    a = np.array([5.0, 0.8, 0.4])
    y = 20 * cm.eval_batch(a[None])[0][0] + 3 * x / 10.0 + 1.0
    p = oracle_problem(cm, y)
    p.set_params(a * np.array([1.02, 0.95, 1.1]))
    rep = p.fit()
    assert rep.termination > 0 and np.abs(p.params() - a).max() <= 1e-6


def test_closure_model_builder_errors():
    import pytest
    cm = vp.ClosureModel(["a", "b"], np.arange(4.0))
    cm.function(["a"], lambda x, a: x * a)
    with pytest.raises(vp.ModelBuildError):
        cm.shape()  # MissingDerivative
    cm.partial_deriv("a", lambda x, a: x)
    with pytest.raises(vp.ModelBuildError):
        cm.partial_deriv("a", lambda x, a: x)  # DuplicateDerivative
    with pytest.raises(vp.ModelBuildError):
        cm.partial_deriv("b", lambda x, a: x)  # InvalidDerivative: not a parameter of this function
    with pytest.raises(vp.ModelBuildError) as ei:
        cm.shape()  # UnusedParameter: 'b' appears in no function (src/model/builder/mod.rs:539-553)
    assert ei.value.variant == "UnusedParameter"
    cm.function(["b", "a"], lambda x, b, a: x * a + b).partial_deriv("b", lambda x, b, a: 1 + 0 * x).partial_deriv("a", lambda x, b, a: x)
    sh = cm.shape()
    assert (sh.n_basis, sh.n_params, sh.ext_pairs) == (2, 2, [(0, 0), (1, 1), (1, 0)])
    # the builder's checks on the lists themselves, by variant name
    for make, variant in [(lambda: vp.ClosureModel([], np.arange(4.0)), "EmptyParameters"),
                          (lambda: vp.ClosureModel(["a", "a"], np.arange(4.0)), "DuplicateParameterNames"),
                          (lambda: vp.ClosureModel(["a,b"], np.arange(4.0)), "CommaInParameterNameNotAllowed"),
                          (lambda: vp.ClosureModel(["a"], np.arange(4.0)).shape(), "EmptyModel"),
                          (lambda: vp.ClosureModel(["a"], np.arange(4.0)).function([], lambda x: x), "EmptyParameters"),
                          (lambda: vp.ClosureModel(["a"], np.arange(4.0)).function(["c"], lambda x, c: x), "FunctionParameterNotInModel"),
                          (lambda: vp.ClosureModel(["a"], np.arange(4.0)).function(["a", "a"], lambda x, a, b: x), "DuplicateParameterNames"),
                          (lambda: vp.ClosureModel(["a"], np.arange(4.0)).partial_deriv("a", lambda x: x), "InvalidDerivative"),
                          (lambda: vp.ClosureModel(["a"], np.arange(4.0)).invariant_function(lambda x: x).partial_deriv("a", lambda x: x),
                           "InvalidDerivative")]:
        with pytest.raises(vp.ModelBuildError) as ei:
            make()
        assert ei.value.variant == variant, (ei.value.variant, variant)
