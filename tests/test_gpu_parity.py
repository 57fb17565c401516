"""GPU parity tests proper: the HIP path (through the C ABI, libvarpro_hip.so) against the CPU oracle on
the same seeded inputs.  fp64 tolerances follow BASELINE.json / SURVEY.md H3:
    |dc| <= 1e-10 * max|c|,  |dr| <= 1e-10 * max|y_w|,  |dJ_k| <= 1e-10 * max|J_k|."""
import numpy as np
import pytest

import refdata as rd
import varpro_amd as vp
from models import double_exp_builder_model, double_exp_unit_test_model, oleary_model
from oracle import oracle as O
from varpro_amd import synth

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _rel(a, b, scale):
    return np.abs(a - b).max() / scale


def _check_eval(mdl, x, Y, alpha, w=None, tol=TOL):
    bp = vp.BatchProblem(mdl, Y, x=x, weights=w)
    got = bp.evaluate(alpha)
    ref = O.evaluate_batch(mdl, x, Y, alpha, w=w, n_threads=4)
    assert (got["status"] == 0).all() and (ref["status"] == 0).all()
    B = Y.shape[0]
    yw = Y if w is None else Y * w
    for b in range(B):
        assert _rel(got["C"][b], ref["C"][b], np.abs(ref["C"][b]).max()) <= tol, "C of problem %d" % b
        assert _rel(got["r"][b], ref["r"][b], np.abs(yw[b]).max()) <= tol, "r of problem %d" % b
        for k in range(mdl.n_params):
            assert _rel(got["J"][b, k], ref["J"][b, k], np.abs(ref["J"][b, k]).max()) <= tol, "J[%d] of %d" % (k, b)
        # cost is an absolute quantity of size ||y_w||^2 * eps at a perfect fit
        assert abs(got["cost"][b] - ref["cost"][b]) <= tol * max(ref["cost"][b], (yw[b] ** 2).sum() * 1e-6)
    # the trait-level calls return the same numbers as the fused call
    bp.set_params(alpha)
    assert np.array_equal(np.asarray(bp.residuals()), got["r"])
    assert np.array_equal(np.asarray(bp.jacobian()), got["J"])
    assert np.array_equal(np.asarray(bp.linear_coefficients()), got["C"])
    assert np.array_equal(np.asarray(bp.cost()), got["cost"])
    bp.close()


@pytest.mark.parametrize("m", [3, 11, 64, 100, 128, 129, 1000, 1023, 1024])
@pytest.mark.parametrize("weighted", [False, True])
def test_evaluate_double_exp_matches_oracle(m, weighted):
    rng = np.random.default_rng(100 + m)
    B = 9
    x = np.sort(rng.random(m)) * 10.0 if m % 2 else 12.5 * np.arange(m) / max(m - 1, 1)
    tau = np.stack([rng.uniform(0.5, 2, B), rng.uniform(2.5, 8, B)], 1)
    c = rng.uniform(0, 100, (B, 3))
    Y = c[:, 0:1] * np.exp(-x / tau[:, 0:1]) + c[:, 1:2] * np.exp(-x / tau[:, 1:2]) + c[:, 2:3]
    Y = Y + 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    alpha = tau * (1 + rng.uniform(-0.3, 0.3, tau.shape))
    w = (0.5 + rng.random(m)) if weighted else None
    mdl = double_exp_builder_model(x, alpha[0])
    _check_eval(mdl, x, Y, alpha, w)


def test_evaluate_quirk_grid_config0():
    # BASELINE configs[0] inputs: cond(Phi) ~ 5e2..9e4 on the reference's (quirky) grid
    c0 = synth.config0()
    mdl = double_exp_builder_model(c0["x"], c0["tau_guess"])
    Y = np.stack([c0["y"], c0["y"]])
    alpha = np.stack([c0["tau_guess"], c0["tau_true"] * 1.05])
    _check_eval(mdl, c0["x"], Y, alpha)


def test_known_answers_octave_through_gpu():
    # src/solvers/levmar/test.rs:111-208 through the runtime-descriptor kernels
    mdl = double_exp_unit_test_model(rd.T11, [2., 4.])
    prob = vp.SeparableProblemBuilder(mdl).observations(rd.Y11).build()
    prob.set_params([2., 4.])
    r = prob.residuals()
    assert np.abs(r).max() < 1e-4 and (r ** 2).sum() < 1e-8
    prob.set_params([0.5, 6.5])
    assert np.abs(prob.residuals() - rd.RES_UNWEIGHTED_05_65).max() < 1e-4
    w = np.sqrt(rd.Y11) + 2 * np.sin(rd.Y11)
    mdl = double_exp_unit_test_model(rd.T11, [0.5, 6.5])
    prob = vp.SeparableProblemBuilder(mdl).observations(rd.Y11).weights(w).build()
    assert np.abs(prob.residuals() - rd.RES_WEIGHTED_05_65).max() < 1e-3
    J = prob.jacobian()
    assert J.shape == (11, 2)
    p = O.Problem(mdl, rd.T11, rd.Y11, w=w)
    p.set_params([0.5, 6.5])
    assert _rel(J.T, p.jacobian(), np.abs(p.jacobian()).max()) <= TOL


def test_evaluate_oleary_model_matches_oracle():
    mdl = oleary_model(rd.OLEARY_T, rd.OLEARY_GUESS)
    Y = np.stack([rd.OLEARY_Y, rd.OLEARY_Y * 1.1])
    alpha = np.stack([rd.OLEARY_GUESS, rd.OLEARY_ALPHA])
    _check_eval(mdl, rd.OLEARY_T, Y, alpha, rd.OLEARY_W)


@pytest.mark.parametrize("m", [11, 1000, 1024])
def test_basis_kernel_matches_oracle(m):
    rng = np.random.default_rng(7)
    x = 12.5 * np.arange(m) / (m - 1)
    B = 5
    alpha = np.stack([rng.uniform(0.5, 2, B), rng.uniform(2.5, 8, B)], 1)
    mdl = double_exp_builder_model(x, alpha[0])
    bp = vp.BatchProblem(mdl, np.zeros((B, m)), x=x)
    phi, dphi = bp.basis(alpha)
    assert phi.shape == (B, 3, m) and dphi.shape == (B, 2, m)
    for b in range(B):
        ref = O.eval_phi(mdl, x, alpha[b])
        assert np.abs(phi[b] - ref).max() <= 4e-16 * 1.0 + 1e-15 * np.abs(ref).max()
        for k in range(2):
            refd = O.eval_dphi(mdl, x, alpha[b], k)[k]
            assert np.abs(dphi[b, k] - refd).max() <= 1e-14 * np.abs(refd).max()
    phi2, _ = bp.basis(alpha, skip_invariant=True)
    assert phi2.shape == (B, 2, m) and np.array_equal(phi2, phi[:, :2])
    bp.close()


def _check_fit(mdl, x, Y, guess, w=None, tol_alpha=1e-8):
    bp = vp.BatchProblem(mdl, Y, x=x, weights=w)
    alpha, C, rep = bp.fit(guess)
    a_ref, C_ref, rep_ref, _ = O.fit_batch(mdl, x, Y, guess, w=w, n_threads=4)
    ok = rep_ref["termination"] > 0
    assert ok.mean() > 0.9
    assert ((rep["termination"] > 0) == ok).all()
    scale_a = np.abs(a_ref).max(1, keepdims=True)
    scale_c = np.abs(C_ref).max(1, keepdims=True)
    assert (np.abs(alpha - a_ref)[ok] <= tol_alpha * np.broadcast_to(scale_a, alpha.shape)[ok]).all()
    assert (np.abs(C - C_ref)[ok] <= 10 * tol_alpha * np.broadcast_to(scale_c, C.shape)[ok]).all()
    assert np.abs(rep["objective"][ok] - rep_ref["objective"][ok]).max() <= 1e-8 * max(rep_ref["objective"][ok].max(), 1e-300)
    # same decision sequence as the oracle: identical number of evaluations for (nearly) all problems
    assert (rep["n_evals"][ok] == rep_ref["n_evals"][ok]).mean() > 0.9
    # handle state after fit == state at the final parameters
    assert np.array_equal(np.asarray(bp.params()), alpha)
    r = bp.residuals()
    assert np.abs(0.5 * (r ** 2).sum(1)[ok] - rep["objective"][ok]).max() <= 1e-9 * max(rep["objective"][ok].max(), 1e-300)
    s = bp.summary()
    assert s[1] == (rep["termination"] > 0).sum() and s[2] == (rep["termination"] <= 0).sum()
    assert s[3] == rep["n_evals"].sum()
    bp.close()
    return alpha, C, rep


def test_fit_config0_recovers_truth():
    # BASELINE configs[0] / tests/integration_tests/main.rs:160-227: tau=(1,3), c=(4,2.5,1) to 1e-8
    c0 = synth.config0()
    mdl = double_exp_builder_model(c0["x"], c0["tau_guess"])
    prob = vp.SeparableProblemBuilder(mdl).observations(c0["y"]).build()
    res = vp.LevMarSolver.default().fit(prob)
    assert res.was_successful()
    assert np.abs(res.nonlinear_parameters() - c0["tau_true"]).max() < 1e-8
    assert np.abs(res.linear_coefficients() - c0["c_true"]).max() < 1e-8
    assert np.abs(res.best_fit() - c0["y"]).max() < 1e-5
    p = O.Problem(mdl, c0["x"], c0["y"])
    p.set_params(c0["tau_guess"])
    rep = p.fit()
    assert res.minimization_report.number_of_evaluations == rep.n_evals
    assert res.minimization_report.termination.code == rep.termination


@pytest.mark.parametrize("m,noise", [(1024, 1e-3), (1024, 0.0), (1000, 1e-3), (100, 1e-3)])
def test_fit_batch_matches_oracle(m, noise):
    d = synth.double_exp_batch(48, m=m, noise=noise)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    _check_fit(mdl, d["x"], d["Y"], d["tau_guess"])


def test_fit_weighted_lmfit_fixture():
    # tests/integration_tests/main.rs:616-668
    x = rd.read_raw_f64("weighted_multiexp_xdata_1000_64bit.raw")
    y = rd.read_raw_f64("weighted_multiexp_ydata_1000_64bit.raw")
    mdl = double_exp_builder_model(x, [1., 7.])
    prob = vp.SeparableProblemBuilder(mdl).observations(y).weights(1.0 / np.sqrt(y)).build()
    res = vp.LevMarSolver.default().fit(prob)
    assert np.abs(res.nonlinear_parameters() - rd.LMFIT_WEIGHTED["tau"]).max() < 1e-5
    assert np.abs(res.linear_coefficients() - rd.LMFIT_WEIGHTED["c"]).max() < 1e-5


def test_fit_oleary_example():
    # tests/integration_tests/main.rs:713-778
    mdl = oleary_model(rd.OLEARY_T, rd.OLEARY_GUESS)
    prob = vp.SeparableProblemBuilder(mdl).observations(rd.OLEARY_Y).weights(rd.OLEARY_W).build()
    res = vp.LevMarSolver.default().fit(prob)
    assert res.was_successful()
    assert np.abs(res.nonlinear_parameters() - rd.OLEARY_ALPHA).max() < 1e-5
    assert np.abs(res.linear_coefficients() - rd.OLEARY_C).max() < 1e-5
    assert np.abs(res.problem.residuals() - rd.OLEARY_WRES).max() < 1e-5
