"""m < n: fewer samples than basis functions.  The reference's linear solve is a truncated SVD (src/solvers/levmar/mod.rs:51-54:
`svd.solve(&y_w, eps)`), which returns the MINIMUM-NORM coefficients of the underdetermined system, a residual at rounding
level and -- through the Kaufman formula on that residual -- a Jacobian at rounding level.  The handle pads the problem to n rows
of which n - m carry zero weight (vp_api.hip: vp_batch_create) and strips them from everything that crosses the ABI."""
import numpy as np
import pytest

import varpro_amd as vp
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,S,weighted", [(2, 1, False), (2, 1, True), (1, 1, False), (2, 3, False), (3, 2, True)])
def test_evaluation_is_the_minimum_norm_solution_of_the_reference(m, S, weighted):
    rng = np.random.default_rng(10 * m + S)
    x = np.linspace(0.0, 1.5, m) if m > 1 else np.array([0.4])
    Y = rng.uniform(1.0, 5.0, (S, m))
    w = rng.uniform(0.5, 2.0, m) if weighted else None
    guess = np.array([1.0, 2.5, 6.0])
    mdl = vp.multi_exponential_model(x, guess, offset=True)          # n = 4 > m
    B = 3
    alphas = guess[None] * rng.uniform(0.8, 1.2, (B, 3))
    Yb = np.stack([Y * (1 + 0.1 * b) for b in range(B)])
    bp = vp.BatchProblem(mdl, Yb if S > 1 else Yb[:, 0], x=x, weights=w)
    ev = bp.evaluate(alphas)
    for b in range(B):
        ref = O.Problem(mdl, x, Yb[b], w=w)
        ref.set_params(alphas[b])
        Cr = ref.linear_coefficients()
        Cg = ev["C"][b] if S > 1 else ev["C"][b][None]
        scale = np.abs(Yb[b] * (1 if w is None else w)).max()
        assert ev["status"][b] == 0
        assert np.abs(Cg - Cr).max() <= 1e-10 * np.abs(Cr).max()
        assert ev["r"][b].shape == (S * m,) and np.abs(ev["r"][b] - ref.residuals()).max() <= 1e-10 * scale
        assert ev["J"][b].shape == (3, S * m) and np.abs(ev["J"][b]).max() <= 1e-10 * scale    # (the oracle's: rounding level too)
        assert np.abs(np.asarray(ref.jacobian())).max() <= 1e-10 * scale
        assert ev["cost"][b] <= 1e-20 * scale ** 2
    # arrays with a row dimension keep the CALLER's m
    phi, dphi = bp.basis(alphas)
    assert phi.shape == (B, 4, m) and dphi.shape == (B, 3, m)
    xs = np.broadcast_to(x, (B, m))
    for j in range(3):
        assert np.abs(phi[:, j] - np.exp(-xs / alphas[:, j:j + 1])).max() <= 1e-14      # (the model's own, unweighted, matrix)
    yw = bp.weighted_data()
    assert np.abs(np.asarray(yw).reshape(B, S, m) - Yb * (1 if w is None else w)).max() <= 1e-15 * np.abs(Yb).max() * 2
    assert np.asarray(bp.residuals()).shape == (B, S * m)
    bf = np.asarray(bp.best_fit()).reshape(B, S, m)
    assert np.abs(bf - Yb).max() <= 1e-10 * np.abs(Yb).max()       # (unweighted model x coefficients:) an exact interpolation
    bp.close()


def test_fit_reports_follow_the_reference_driver():
    # m S >= q: the LM driver runs; the residual is zero at every point, so it stops at once with ResidualsZero / a converged code
    x = np.array([0.0, 1.0])
    Y = np.array([[3.0, 2.0]])
    mdl = vp.multi_exponential_model(x, [1.0, 2.5], offset=True)     # n = 3 > m = 2, q = 2 = m S
    bp = vp.BatchProblem(mdl, Y, x=x)
    a, C, rep = bp.fit(np.array([[1.0, 2.5]]))
    assert np.isfinite(rep["objective"][0]) and rep["objective"][0] <= 1e-25
    assert rep["n_evals"][0] >= 1
    ref = O.Problem(mdl, x, Y[0][None])
    ref.set_params([1.0, 2.5])
    assert np.abs(C[0] - ref.linear_coefficients()[0]).max() <= 1e-9 or rep["n_evals"][0] > 1
    bp.close()
    # m S < q: the driver evaluates once and reports WrongDimensions (levenberg-marquardt 0.14: `if n > m`)
    x1 = np.array([0.5])
    mdl3 = vp.multi_exponential_model(x1, [1.0, 2.5, 6.0], offset=True)   # q = 3 > m S = 1
    bp = vp.BatchProblem(mdl3, np.array([[2.0]]), x=x1)
    a, C, rep = bp.fit(np.array([[1.0, 2.5, 6.0]]))
    assert rep["termination"][0] == -7 and rep["n_evals"][0] == 1
    assert np.array_equal(a, [[1.0, 2.5, 6.0]])
    refw = O.Problem(mdl3, x1, np.array([[2.0]]))
    refw.set_params([1.0, 2.5, 6.0])
    rr = refw.fit()
    assert rr.termination == -7 and rr.n_evals == 1
    bp.set_observations(np.array([[4.0]]))
    ev = bp.evaluate(np.array([[1.0, 2.5, 6.0]]))
    assert abs(np.asarray(bp.best_fit()).ravel()[0] - 4.0) <= 1e-12
    bp.close()


def test_device_pointer_handle_with_fewer_samples_than_basis_functions():
    # the same through device pointers (torch tensors on cuda:0): padding happens inside vp_batch_create / vp_set_observations,
    # outputs arrive with the caller's m in the caller's device arrays
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    m, B = 2, 5
    x = np.array([0.2, 1.1])
    Y = rng.uniform(1.0, 5.0, (B, m))
    alphas = np.array([1.0, 2.5, 6.0])[None] * rng.uniform(0.8, 1.2, (B, 3))
    mdl = vp.multi_exponential_model(x, alphas[0], offset=True)
    bp = vp.BatchProblem(mdl, torch.from_numpy(Y).to(dev), x=torch.from_numpy(x).to(dev))
    ev = bp.evaluate(torch.from_numpy(alphas).to(dev))
    C = ev["C"].cpu().numpy()
    r = ev["r"].cpu().numpy()
    assert r.shape == (B, m) and ev["J"].shape == (B, 3, m)
    for b in range(B):
        ref = O.Problem(mdl, x, Y[b][None])
        ref.set_params(alphas[b])
        assert np.abs(C[b] - ref.linear_coefficients()[0]).max() <= 1e-10 * np.abs(ref.linear_coefficients()).max()
        assert np.abs(r[b]).max() <= 1e-10 * np.abs(Y[b]).max()
    Y2 = rng.uniform(1.0, 5.0, (B, m))
    bp.set_observations(torch.from_numpy(Y2).to(dev))
    ev2 = bp.evaluate(torch.from_numpy(alphas).to(dev))
    bf = bp.best_fit().cpu().numpy()
    assert bf.shape == (B, m) and np.abs(bf - Y2).max() <= 1e-10 * np.abs(Y2).max()
    bp.close()
