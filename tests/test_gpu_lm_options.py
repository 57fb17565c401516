"""The LM option surface on the device against the oracle: the builder knobs of levenberg_marquardt::LevenbergMarquardt
that the reference reaches through LevMarSolver::with_solver (src/solvers/levmar/mod.rs:221-223; used by the
reference's own tests at tests/integration_tests/main.rs:283-286, 361-365) and the rarer termination reasons
(src/solvers/levmar/mod.rs:249-253 maps them to Ok / Err).  Asserted: identical termination CODES, identical
evaluation counts where the trajectory stays above the rounding floor, same minimum.  Both device kernels (one
wavefront per problem, persistent slots) are driven."""
import numpy as np
import pytest

import varpro_amd as vp
from models import double_exp_builder_model
from oracle import oracle as O
from varpro_amd import synth

pytestmark = pytest.mark.gpu

T = vp.solver.TERMINATION_NAMES


def _both(mdl, d, solver, oracle_kw, kernel, w=None):
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"], weights=w)
    bp.set_fit_kernel(kernel)
    a, c, rep = bp.fit(d["tau_guess"], solver=solver)
    bp.close()
    a_ref, c_ref, rep_ref, _ = O.fit_batch(mdl, d["x"], d["Y"], d["tau_guess"], w=w, opts=O.default_opts(**oracle_kw),
                                           n_threads=4)
    return (a, c, rep), (a_ref, c_ref, rep_ref)


def _same_minimum(rep, rep_ref, sel, rel=1e-6):
    o, r = rep["objective"][sel], rep_ref["objective"][sel]
    assert (np.abs(o - r) <= rel * np.maximum(r, 1e-300)).all()


@pytest.mark.parametrize("kernel", ["wave", "slots"])
def test_stepbound_one_and_long_patience(kernel):
    # tests/integration_tests/main.rs:283-286, 361-365: with_stepbound(1.) / with_patience(1000)
    d = synth.double_exp_batch(64, m=1024, noise=1e-3)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    solver = vp.LevenbergMarquardt().with_stepbound(1.0).with_patience(1000)
    (a, c, rep), (a_ref, c_ref, rep_ref) = _both(mdl, d, solver, dict(stepbound=1.0, patience=1000), kernel)
    assert np.array_equal(rep["termination"] > 0, rep_ref["termination"] > 0)
    ok = rep_ref["termination"] > 0
    assert ok.mean() > 0.9
    _same_minimum(rep, rep_ref, ok)
    assert abs(rep["n_evals"][ok].mean() - rep_ref["n_evals"][ok].mean()) <= 0.1 * rep_ref["n_evals"][ok].mean()
    assert (np.abs(rep["n_evals"][ok].astype(int) - rep_ref["n_evals"][ok]) <= 3).mean() >= 0.5


@pytest.mark.parametrize("kernel", ["wave", "slots"])
@pytest.mark.parametrize("patience", [1, 2])
def test_lost_patience_same_code_and_same_evaluation_count(kernel, patience):
    # max evaluations = patience * (q + 1): both sides stop at exactly that count with LostPatience (an Err in
    # LevMarSolver::fit), unless a fit converged earlier.  A fit whose convergence test fires AT the last allowed
    # evaluation on one side and one evaluation later on the other (rounding-floor dithering, see test_gpu_parity) is
    # the only disagreement possible: at most a few per cent of the batch
    d = synth.double_exp_batch(96, m=1024, noise=1e-3)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    solver = vp.LevenbergMarquardt().with_patience(patience)
    (a, c, rep), (a_ref, c_ref, rep_ref) = _both(mdl, d, solver, dict(patience=patience), kernel)
    both = (rep_ref["termination"] == -4) & (rep["termination"] == -4)
    assert both.mean() > 0.5
    # (with patience = 2 the budget of 6 evaluations sits right where the bulk of this batch converges: more ties)
    assert ((rep["termination"] == -4) == (rep_ref["termination"] == -4)).mean() >= 0.8
    assert (rep["termination"] == rep_ref["termination"]).mean() >= 0.8
    lost = both
    assert (rep["n_evals"][lost] == patience * 3).all() and (rep_ref["n_evals"][lost] == patience * 3).all()
    assert (rep["n_evals"] <= patience * 3).all() and (rep_ref["n_evals"] <= patience * 3).all()
    # the problem keeps the best point found so far: same objective on both sides
    _same_minimum(rep, rep_ref, lost, rel=1e-8)
    assert np.abs(a[lost] - a_ref[lost]).max() <= 1e-7 * np.abs(a_ref[lost]).max()


@pytest.mark.parametrize("kernel", ["wave", "slots"])
def test_scale_diag_off(kernel):
    d = synth.double_exp_batch(64, m=1024, noise=1e-3)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    solver = vp.LevenbergMarquardt().with_scale_diag(False)
    (a, c, rep), (a_ref, c_ref, rep_ref) = _both(mdl, d, solver, dict(scale_diag=0), kernel)
    assert np.array_equal(rep["termination"] > 0, rep_ref["termination"] > 0)
    ok = rep_ref["termination"] > 0
    assert ok.mean() > 0.8
    _same_minimum(rep, rep_ref, ok)
    assert (np.abs(rep["n_evals"][ok].astype(int) - rep_ref["n_evals"][ok]) <= 3).mean() >= 0.5


@pytest.mark.parametrize("kernel", ["wave", "slots"])
def test_large_gtol_terminates_orthogonal_at_the_same_evaluation(kernel):
    # gtol = 1e-2: the scaled gradient test fires long before ftol/xtol -> TerminationReason::Orthogonal (a success)
    d = synth.double_exp_batch(64, m=1024, noise=1e-3)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    solver = vp.LevenbergMarquardt().with_gtol(1e-2)
    (a, c, rep), (a_ref, c_ref, rep_ref) = _both(mdl, d, solver, dict(gtol=1e-2), kernel)
    assert (rep_ref["termination"] == 2).mean() > 0.9
    same = rep["termination"] == rep_ref["termination"]
    assert same.mean() >= 0.97  # (a degenerate fit crawling along a flat valley may trip ftol first on one side)
    same &= rep["n_evals"] == rep_ref["n_evals"]
    assert same.mean() >= 0.95  # the gradient test fires at the same evaluation
    assert np.abs(a[same] - a_ref[same]).max() <= 1e-8 * np.abs(a_ref[same]).max()


@pytest.mark.parametrize("kernel", ["wave", "slots"])
def test_loose_tolerances_terminate_converged_with_identical_codes(kernel):
    # ftol = xtol = 1e-6, far above the rounding floor: the convergence tests fire on the same evaluation with the
    # same reason (Converged{ftol} / {xtol} / both) on both sides
    d = synth.double_exp_batch(96, m=1024, noise=1e-3)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    solver = vp.LevenbergMarquardt().with_ftol(1e-6).with_xtol(1e-6)
    (a, c, rep), (a_ref, c_ref, rep_ref) = _both(mdl, d, solver, dict(ftol=1e-6, xtol=1e-6), kernel)
    assert np.isin(rep_ref["termination"], [3, 4, 5]).mean() > 0.9
    same = rep["termination"] == rep_ref["termination"]
    assert same.mean() >= 0.97, [(T[int(x)], T[int(y)]) for x, y in zip(rep["termination"][~same], rep_ref["termination"][~same])]
    assert (rep["n_evals"][same] == rep_ref["n_evals"][same]).mean() >= 0.97


@pytest.mark.parametrize("kernel", ["wave", "slots"])
def test_exact_data_from_the_true_parameters(kernel):
    # noise-free data, started AT the truth: the residual is at the rounding floor from the first evaluation.  The
    # reference's driver then ends with ResidualsZero / Orthogonal / Converged (success) or NoImprovementPossible
    # (failure) depending on which rounding-level test fires first; device and oracle must agree on success/failure
    # per problem and both report an objective at the rounding floor
    d = synth.double_exp_batch(64, m=1024, noise=0.0)
    d = dict(d, tau_guess=d["tau_true"].copy())
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    (a, c, rep), (a_ref, c_ref, rep_ref) = _both(mdl, d, vp.LevenbergMarquardt(), {}, kernel)
    scale = (d["Y"] ** 2).sum(1)
    assert (rep["objective"] <= 1e-24 * scale).all() and (rep_ref["objective"] <= 1e-24 * scale).all()
    assert np.isin(rep["termination"], [1, 2, 3, 4, 5, -3]).all() and np.isin(rep_ref["termination"], [1, 2, 3, 4, 5, -3]).all()
    assert np.abs(a - d["tau_true"]).max() <= 1e-9 and np.abs(a_ref - d["tau_true"]).max() <= 1e-9
    assert (rep["n_evals"] <= 12).all() and (rep_ref["n_evals"] <= 12).all()


def test_residuals_zero_and_wrong_dimensions_codes():
    # y == 0: ||r|| = 0 at the first evaluation -> ResidualsZero (success) with exactly one evaluation on both sides
    m = 1024
    x = 12.5 * np.arange(m) / (m - 1)
    mdl = double_exp_builder_model(x, [1.0, 4.0])
    Y = np.zeros((3, m))
    g = np.array([[1.0, 4.0], [0.7, 5.0], [1.5, 3.0]])
    for kernel in ("wave", "slots"):
        bp = vp.BatchProblem(mdl, Y, x=x)
        bp.set_fit_kernel(kernel)
        a, c, rep = bp.fit(g)
        bp.close()
        _a, _c, rep_ref, _ = O.fit_batch(mdl, x, Y, g)
        assert (rep_ref["termination"] == 1).all() and (rep["termination"] == 1).all()
        assert (rep["n_evals"] == 1).all() and (rep_ref["n_evals"] == 1).all()
        assert np.array_equal(a, g)
