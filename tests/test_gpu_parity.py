"""GPU parity tests proper: the HIP path (through the C ABI, libvarpro_hip.so) against the CPU oracle on
the same seeded inputs.  fp64 tolerances follow BASELINE.json / SURVEY.md H3:
    |dc| <= 1e-10 * max|c|,  |dr| <= 1e-10 * max|y_w|,  |dJ_k| <= 1e-10 * max|J_k|.
The contract on a FIT: identical success flags, the same minimum (alpha, c, objective to the tolerances in the tests) and
the same LM trajectory up to the point where rounding decides -- NOT identical evaluation counts: the last iterations of a
converged fit compare an actual reduction at rounding level with ftol, and the oracle itself changes its count by a few
evaluations when it forms r under `long double`.  The tests therefore ask for |delta n_evals| <= 3
on at least half of the fits, and for equal termination CLASSES everywhere."""
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
            # |dJ_k| <= tol*max|J_k| + rounding floor relative to the un-projected column W D_k c
            # (J_k = -P_perp (W D_k c) cancels; for m == n it is identically zero)
            dkc = (O.eval_dphi(mdl, x, alpha[b], k) * ref["C"][b][:, None]).sum(0) * (1.0 if w is None else w)
            bound = tol * np.abs(ref["J"][b, k]).max() + 1e-13 * np.abs(dkc).max()
            assert np.abs(got["J"][b, k] - ref["J"][b, k]).max() <= bound, "J[%d] of problem %d" % (k, b)
        # cost is an absolute quantity of size ||y_w||^2 * eps at a perfect fit
        assert abs(got["cost"][b] - ref["cost"][b]) <= tol * max(ref["cost"][b], (yw[b] ** 2).sum() * 1e-6)
    # the trait-level calls return the same numbers as the fused call: the Jacobian call IS the fused kernel
    # (bit-identical); residuals / coefficients / cost come from the lighter set_params kernel, which on a uniform
    # grid builds the exponentials by the recurrence -> equal to rounding, not bit for bit
    bp.set_params(alpha)
    ymax = np.abs(yw).max()
    assert np.abs(np.asarray(bp.residuals()) - got["r"]).max() <= 1e-13 * ymax
    assert np.array_equal(np.asarray(bp.jacobian()), got["J"])
    assert np.abs(np.asarray(bp.linear_coefficients()) - got["C"]).max() <= 1e-11 * np.abs(got["C"]).max()
    assert np.abs(np.asarray(bp.cost()) - got["cost"]).max() <= 1e-11 * max(got["cost"].max(), ymax ** 2 * 1e-6)
    bp.close()


@pytest.mark.parametrize("m", [3, 11, 64, 100, 128, 129, 200, 256, 257, 500, 512, 513, 1000, 1023, 1024])
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


def _check_fit(mdl, x, Y, guess, w=None, noise_free=False):
    """The device LM and the oracle LM are the same algorithm (MINPACK lmder semantics of the
    levenberg-marquardt crate) in different floating-point association orders.  They must (a) walk the
    same trajectory while the residual is far above the rounding floor, (b) agree on success/failure,
    (c) reach the same minimum: objective to ~1e-12 relative, parameters to the sqrt(ftol)-limited
    accuracy any ftol-terminated minimiser has (1e-7 relative; 1e-8 on noise-free data, the
    reference's own end-to-end tolerance, tests/integration_tests/main.rs:152-156)."""
    bp = vp.BatchProblem(mdl, Y, x=x, weights=w)
    alpha, C, rep, tr = bp.fit_trace(guess, max_rows=16)
    a_ref, C_ref, rep_ref, _ = O.fit_batch(mdl, x, Y, guess, w=w, n_threads=4)
    B, q = alpha.shape
    ok = rep_ref["termination"] > 0
    assert ok.mean() > 0.9
    assert ((rep["termination"] > 0) == ok).all()
    # (a) per-iteration parity of the first evaluations
    n_rows_checked = 0
    for b in range(B):
        p = O.Problem(mdl, x, Y[b], w=w)
        p.set_params(guess[b])
        _r, tr_ref = p.fit_trace(max_rows=16)
        f0 = tr_ref[0, q]
        for i in range(min(6, len(tr_ref), int(rep["n_evals"][b]))):
            if tr_ref[i, q] < 1e-6 * f0:
                break  # inside the rounding-noise regime of a (near) perfect fit
            g, o = tr[b, i], tr_ref[i]
            assert np.abs(g[:q] - o[:q]).max() <= 1e-8 * np.abs(o[:q]).max(), (b, i, g, o)
            assert abs(g[q] - o[q]) <= 1e-8 * abs(o[q]), (b, i, g, o)
            n_rows_checked += 1
    assert n_rows_checked >= 3 * B
    # (c) same minimum
    rel_a = (np.abs(alpha - a_ref) / np.abs(a_ref).max(1, keepdims=True)).max(1)
    rel_c = (np.abs(C - C_ref) / np.abs(C_ref).max(1, keepdims=True)).max(1)
    if noise_free:
        # (seeds whose guess sits in the basin of a degenerate fit, second decay -> infinity, excepted)
        assert (rel_a[ok] <= 1e-8).mean() >= 0.9 and (rel_c[ok] <= 1e-8).mean() >= 0.9
        assert np.median(rel_a[ok]) <= 1e-12
    else:
        rel_o = np.abs(rep["objective"] - rep_ref["objective"]) / rep_ref["objective"]
        assert (rel_o[ok] <= 1e-6).all() and np.median(rel_o[ok]) <= 1e-12
        # a few seeds are (near-)degenerate fits whose second decay runs off to infinity: the
        # objective is flat along that direction and the parameters are not determined
        assert (rel_a[ok] <= 1e-7).mean() >= 0.9 and (rel_c[ok] <= 1e-6).mean() >= 0.9
    # the number of evaluations differs only by how long each implementation dithers at the rounding
    # floor before a termination test fires (the trajectories above it are identical, (a))
    assert (np.abs(rep["n_evals"][ok].astype(int) - rep_ref["n_evals"][ok]) <= 3).mean() >= 0.5
    assert abs(rep["n_evals"][ok].mean() - rep_ref["n_evals"][ok].mean()) <= 0.25 * rep_ref["n_evals"][ok].mean()
    # handle state after fit == state at the final parameters
    assert np.array_equal(np.asarray(bp.params()), alpha)
    assert np.array_equal(np.asarray(bp.linear_coefficients()), C)
    r = bp.residuals()
    cost = 0.5 * (r ** 2).sum(1)
    # (the residual cache is filled by the trait-level evaluation kernel, whose sweep orders the columns differently
    # from the fit kernel's constant-first sweep: equal to rounding, which near-degenerate fits amplify)
    rel = (np.abs(cost - rep["objective"]) / np.maximum(rep["objective"], 1e-12 * (Y ** 2).sum(1)))[ok]
    assert np.median(rel) <= 1e-12 and np.quantile(rel, 0.9) <= 1e-9 and rel.max() <= 1e-5
    s = bp.summary()
    assert s[1] == (rep["termination"] > 0).sum() and s[2] == (rep["termination"] <= 0).sum()
    assert s[3] == rep["n_evals"].sum()
    assert abs(s[0] - rep["objective"].sum()) <= 1e-12 * rep["objective"].sum()
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
    # trajectory parity with the oracle down to the rounding floor (cond(Phi) up to 9e4 here)
    bp = vp.BatchProblem(mdl, c0["y"][None, :], x=c0["x"])
    _a, _c, rep, tr = bp.fit_trace(c0["tau_guess"][None, :], max_rows=40)
    p = O.Problem(mdl, c0["x"], c0["y"])
    p.set_params(c0["tau_guess"])
    rep_ref, tr_ref = p.fit_trace(max_rows=40)
    assert rep_ref.termination > 0 and rep["termination"][0] > 0
    for i in range(11):
        assert np.abs(tr[0, i, :2] - tr_ref[i, :2]).max() <= 1e-8 * np.abs(tr_ref[i, :2]).max()
        assert abs(tr[0, i, 2] - tr_ref[i, 2]) <= 1e-8 * tr_ref[i, 2]
    assert abs(int(rep["n_evals"][0]) - rep_ref.n_evals) <= 8
    bp.close()


@pytest.mark.parametrize("m,noise", [(1024, 1e-3), (1024, 0.0), (1000, 1e-3), (100, 1e-3), (256, 1e-3), (400, 1e-3), (512, 1e-3)])
def test_fit_batch_matches_oracle(m, noise):
    d = synth.double_exp_batch(48, m=m, noise=noise)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    _check_fit(mdl, d["x"], d["Y"], d["tau_guess"], noise_free=(noise == 0.0))


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
