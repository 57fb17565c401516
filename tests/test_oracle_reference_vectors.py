"""Pin the CPU oracle against every known-answer vector the reference's own tests hold for the hot
path.  CPU only; these run in the `-m "not gpu"` tier."""
import numpy as np
import pytest

import refdata as rd
from models import double_exp_builder_model, double_exp_unit_test_model, numpy_reference_eval, oleary_model
from oracle import oracle as O
from varpro_amd import synth


def test_residuals_at_truth_are_small():
    # src/solvers/levmar/test.rs:111-140
    mdl = double_exp_unit_test_model(rd.T11, [2., 4.])
    p = O.Problem(mdl, rd.T11, rd.Y11)
    p.set_params([2., 4.])
    r = p.residuals()
    assert np.abs(r).max() < 1e-4
    assert (r ** 2).sum() < 1e-8


def test_residuals_unweighted_match_octave():
    # src/solvers/levmar/test.rs:141-162 (epsilon = 1e-4)
    mdl = double_exp_unit_test_model(rd.T11, [2., 4.])
    p = O.Problem(mdl, rd.T11, rd.Y11)
    p.set_params([0.5, 6.5])
    assert np.abs(p.residuals() - rd.RES_UNWEIGHTED_05_65).max() < 1e-4


def test_residuals_weighted_match_octave():
    # src/solvers/levmar/test.rs:166-207 (epsilon = 1e-3)
    w = np.sqrt(rd.Y11) + 2 * np.sin(rd.Y11)
    mdl = double_exp_unit_test_model(rd.T11, [0.5, 6.5])
    p = O.Problem(mdl, rd.T11, rd.Y11, w=w)
    p.set_params([0.5, 6.5])
    assert np.abs(p.residuals() - rd.RES_WEIGHTED_05_65).max() < 1e-3


def _fd_jacobian(p, alpha, h=1e-6):
    alpha = np.asarray(alpha, dtype=float)
    J = []
    for k in range(alpha.size):
        ap, am = alpha.copy(), alpha.copy()
        ap[k] += h
        am[k] -= h
        p.set_params(ap)
        rp = p.residuals()
        p.set_params(am)
        rm = p.residuals()
        J.append((rp - rm) / (2 * h))
    p.set_params(alpha)
    return np.array(J)


def test_jacobian_matches_finite_differences_at_truth():
    # src/solvers/levmar/test.rs:21-40: at the true parameters the Kaufman Jacobian equals the full one
    mdl = double_exp_unit_test_model(rd.T11, [2., 4.])
    p = O.Problem(mdl, rd.T11, rd.Y11)
    Jn = _fd_jacobian(p, [2., 4.])
    assert np.abs(Jn - p.jacobian()).max() < 1e-4


def test_gradient_identity_weighted():
    # src/solvers/levmar/test.rs:51-108: d||r||^2/dtau_k = 2 r^T J_k away from the optimum, weighted
    w = np.sqrt(rd.Y11) + np.sin(rd.Y11)
    mdl = double_exp_unit_test_model(rd.T11, [1., 2.])
    p = O.Problem(mdl, rd.T11, rd.Y11, w=w)
    a0 = np.array([0.5, 7.5])

    def ssq(a):
        p.set_params(a)
        return (p.residuals() ** 2).sum()

    h = 1e-5
    for k in range(2):
        e = np.zeros(2)
        e[k] = h
        num = (-ssq(a0 + 2 * e) + 8 * ssq(a0 + e) - 8 * ssq(a0 - e) + ssq(a0 - 2 * e)) / (12 * h)
        p.set_params(a0)
        calc = 2.0 * p.residuals().dot(p.jacobian()[k])
        assert abs(num - calc) < 1e-6


@pytest.mark.parametrize("builder", [double_exp_builder_model, double_exp_unit_test_model])
def test_oracle_vs_numpy_svd(builder):
    rng = np.random.default_rng(5)
    x = np.linspace(0, 10, 57)
    mdl = builder(x, [1.3, 4.1])
    y = 2 * np.exp(-x / 2) + np.exp(-x / 4) + 1 + 0.01 * rng.standard_normal(x.size)
    w = 0.5 + rng.random(x.size)
    for ww in (None, w):
        p = O.Problem(mdl, x, y, w=ww)
        p.set_params([1.3, 4.1])
        c, r, J = numpy_reference_eval(mdl, x, y, [1.3, 4.1], ww)
        assert np.abs(c - p.linear_coefficients()).max() <= 1e-12 * np.abs(c).max()
        assert np.abs(r - p.residuals()).max() <= 1e-12 * np.abs(y).max()
        assert np.abs(J - p.jacobian()).max() <= 1e-12 * np.abs(J).max()


def test_config0_fit_recovers_truth():
    # tests/integration_tests/main.rs:93-157, 160-227 (epsilon 1e-8) == BASELINE configs[0]
    c0 = synth.config0()
    mdl = double_exp_builder_model(c0["x"], c0["tau_guess"])
    p = O.Problem(mdl, c0["x"], c0["y"])
    p.set_params(c0["tau_guess"])
    rep = p.fit()
    assert rep.termination > 0
    tau, c = p.params(), p.linear_coefficients()
    assert np.abs(tau - c0["tau_true"]).max() < 1e-8
    assert np.abs(c - c0["c_true"]).max() < 1e-8
    assert np.abs(p.best_fit() - c0["y"]).max() < 1e-5


def test_mrhs_fits_recover_truth():
    # tests/integration_tests/main.rs:399-463 (S=2, branch S<=q) and :467-551 (S=3, branch S>q)
    x = synth.linspace_reference(0., 12.5, 20)
    coeffs = {2: [(2., 4., 0.2), (5., 1., 9.)], 3: [(2., 4., 0.2), (10., 12., 18.), (5., 1., 9.)]}
    for S, cs in coeffs.items():
        Y = np.stack([a * np.exp(-x / 1.) + b * np.exp(-x / 3.) + c for a, b, c in cs])  # (S, m)
        mdl = double_exp_builder_model(x, [2.5, 6.5])
        p = O.Problem(mdl, x, Y)
        p.set_params([2.5, 6.5])
        rep = p.fit()
        assert rep.termination > 0
        tau = p.params()
        i1, i2 = (0, 1) if tau[0] < tau[1] else (1, 0)
        C = p.linear_coefficients()
        assert abs(tau[i1] - 1.) < 1e-8 and abs(tau[i2] - 3.) < 1e-8
        for s, (a, b, c) in enumerate(cs):
            assert abs(C[s, i1] - a) < 1e-8 and abs(C[s, i2] - b) < 1e-8 and abs(C[s, 2] - c) < 1e-8
        assert np.abs(p.best_fit() - Y).max() < 1e-5


def test_oleary_example():
    # tests/integration_tests/main.rs:713-778 (MATLAB varpro output, epsilon 1e-5)
    mdl = oleary_model(rd.OLEARY_T, rd.OLEARY_GUESS)
    p = O.Problem(mdl, rd.OLEARY_T, rd.OLEARY_Y, w=rd.OLEARY_W)
    p.set_params(rd.OLEARY_GUESS)
    rep = p.fit()
    assert rep.termination > 0
    assert np.abs(p.params() - rd.OLEARY_ALPHA).max() < 1e-5
    assert np.abs(p.linear_coefficients() - rd.OLEARY_C).max() < 1e-5
    assert np.abs(p.residuals() - rd.OLEARY_WRES).max() < 1e-5
    assert np.abs(p.best_fit() - rd.OLEARY_Y).max() < 1e-2


@pytest.mark.parametrize("weighted", [False, True])
def test_lmfit_fixtures(weighted):
    # tests/integration_tests/main.rs:554-598 and :616-668 (python lmfit, epsilon 1e-5)
    pre = "weighted_multiexp" if weighted else "multiexp"
    x = rd.read_raw_f64(pre + "_xdata_1000_64bit.raw")
    y = rd.read_raw_f64(pre + "_ydata_1000_64bit.raw")
    assert x.size == 1000 and y.size == 1000
    w = 1.0 / np.sqrt(y) if weighted else None
    exp = rd.LMFIT_WEIGHTED if weighted else rd.LMFIT_UNWEIGHTED
    mdl = double_exp_builder_model(x, [1., 7.])
    p = O.Problem(mdl, x, y, w=w)
    p.set_params([1., 7.])
    rep = p.fit()
    assert rep.termination > 0
    assert np.abs(p.params() - exp["tau"]).max() < 1e-5
    assert np.abs(p.linear_coefficients() - exp["c"]).max() < 1e-5


def test_wide_basis_matrix_takes_the_minimum_norm_solution():
    """m < n (fewer samples than basis functions): nalgebra's `svd(true, true).solve(b, eps)` (src/solvers/levmar/mod.rs:51-54)
    returns the minimum-norm solution of the underdetermined system.  The reference holds no vector for this branch; the
    oracle's is pinned here against numpy's minimum-norm `lstsq` (LAPACK gelsd), for unit and general weights."""
    import varpro_amd as vp
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    for m, S, weighted in ((2, 1, False), (1, 1, False), (3, 2, True), (2, 3, True)):
        x = np.linspace(0.0, 1.5, m) if m > 1 else np.array([0.4])
        Y = rng.uniform(1.0, 5.0, (S, m))
        w = rng.uniform(0.5, 2.0, m) if weighted else None
        alpha = np.array([1.0, 2.5, 6.0])
        mdl = vp.multi_exponential_model(x, alpha, offset=True)                  # n = 4 > m
        ref = O.Problem(mdl, x, Y, w=w)
        assert ref.set_params(alpha) is not False
        ww = np.ones(m) if w is None else w
        Phi = np.stack([np.exp(-x / a) for a in alpha] + [np.ones(m)], axis=1) * ww[:, None]   # m x n
        Cn = np.linalg.lstsq(Phi, (Y * ww).T, rcond=None)[0].T                                # S x n, minimum norm
        assert np.abs(ref.linear_coefficients() - Cn).max() <= 1e-12 * np.abs(Cn).max()
        assert np.abs(ref.residuals()).max() <= 1e-13 * np.abs(Y * ww).max()
        assert np.abs(np.asarray(ref.jacobian())).max() <= 1e-12 * np.abs(Y * ww).max()
