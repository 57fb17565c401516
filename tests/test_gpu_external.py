"""Models OUTSIDE the descriptor language: the reference's plugin boundary
is a trait -- any `SeparableNonlinearModel` (/root/reference/src/model/mod.rs:239-363), in particular the closure-based
`SeparableModel` (:441-512) -- so a Gaussian, a Lorentzian or a basis function of three parameters must be a drop-in too.
Here the CALLER evaluates Phi and the non-zero columns of dPhi/dalpha_k (numpy closures, `vp.ClosureModel`), the device
does everything downstream (weighting, factorisation / truncated solve, residual, Kaufman Jacobian: src/solvers/levmar/
mod.rs:42-73, 101-201) through vp_batch_create_external / vp_set_params_with_basis / vp_jacobian_with_derivatives /
vp_evaluate_with_basis.  The checker is the oracle given THE SAME closures (vpo_problem_set_external_model).
Tolerances: north_star's 1e-10 on c, r, J (J with the usual rounding floor relative to the un-projected column)."""
import numpy as np
import pytest

import varpro_amd as vp
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


# ---- the models: nothing here is expressible with the five descriptor kinds ------------------------------------------
def gauss(x, mu, sg):
    return np.exp(-0.5 * ((x - mu) / sg) ** 2)


def gauss_dmu(x, mu, sg):
    return gauss(x, mu, sg) * (x - mu) / sg ** 2


def gauss_dsg(x, mu, sg):
    return gauss(x, mu, sg) * (x - mu) ** 2 / sg ** 3


def lorentz(x, mu, ga):
    return ga ** 2 / ((x - mu) ** 2 + ga ** 2)


def lorentz_dmu(x, mu, ga):
    return 2 * ga ** 2 * (x - mu) / ((x - mu) ** 2 + ga ** 2) ** 2


def lorentz_dga(x, mu, ga):
    return 2 * ga * (x - mu) ** 2 / ((x - mu) ** 2 + ga ** 2) ** 2


def pvoigt(x, mu, w, eta):  # THREE parameters in one basis function (the descriptor language stops at two)
    return eta * lorentz(x, mu, w) + (1 - eta) * gauss(x, mu, w)


def pvoigt_dmu(x, mu, w, eta):
    return eta * lorentz_dmu(x, mu, w) + (1 - eta) * gauss_dmu(x, mu, w)


def pvoigt_dw(x, mu, w, eta):
    return eta * lorentz_dga(x, mu, w) + (1 - eta) * gauss_dsg(x, mu, w)


def pvoigt_deta(x, mu, w, eta):
    return lorentz(x, mu, w) - gauss(x, mu, w)


def peaks_model(x, dtype=np.float64):
    """c1 Gauss(mu1, s1) + c2 Lorentz(mu2, g2) + c3: n = 3, q = 4, p = 4"""
    return (vp.ClosureModel(["mu1", "s1", "mu2", "g2"], x, dtype=dtype)
            .function(["mu1", "s1"], gauss).partial_deriv("mu1", gauss_dmu).partial_deriv("s1", gauss_dsg)
            .function(["mu2", "g2"], lorentz).partial_deriv("mu2", lorentz_dmu).partial_deriv("g2", lorentz_dga)
            .invariant_function(lambda x: np.ones_like(x)))


def voigt_model(x):
    """c1 pVoigt(mu, w, eta) + c2 x + c3: one basis with three parameters, a linear background; n = 3, q = 3, p = 3"""
    return (vp.ClosureModel(["mu", "w", "eta"], x)
            .function(["mu", "w", "eta"], pvoigt).partial_deriv("mu", pvoigt_dmu).partial_deriv("w", pvoigt_dw)
            .partial_deriv("eta", pvoigt_deta)
            .invariant_function(lambda x: x / 10.0)
            .invariant_function(lambda x: np.ones_like(x)))


def oracle_problem(cm, Y, w=None, eps=-1.0):
    """the CPU oracle driven by the same closures (the reference's trait calls, src/solvers/levmar/mod.rs:45, :141)"""
    sh = cm.shape()
    prs = cm.pairs()

    def ev(a):
        return cm.eval_batch(a[None])[0]

    def dv(a, k):
        d = cm.derivs_batch(a[None])[0]
        out = np.zeros((sh.n_basis, cm.x.size))
        for p, (j, kk) in enumerate(prs):
            if kk == k:
                out[j] += d[p]
        return out

    return O.Problem(O.make_shape_desc(sh.n_basis, sh.n_params), None, Y, w=w, eps=eps, external=(ev, dv))


def peaks_data(rng, B, x, noise=1e-3):
    mu1 = rng.uniform(2.5, 3.5, B)
    s1 = rng.uniform(0.4, 0.9, B)
    mu2 = rng.uniform(6.0, 7.0, B)
    g2 = rng.uniform(0.5, 1.2, B)
    truth = np.stack([mu1, s1, mu2, g2], 1)
    c = np.stack([rng.uniform(5, 50, B), rng.uniform(5, 50, B), rng.uniform(0, 5, B)], 1)
    Y = (c[:, 0:1] * gauss(x, mu1[:, None], s1[:, None]) + c[:, 1:2] * lorentz(x, mu2[:, None], g2[:, None]) + c[:, 2:3])
    Y = Y + noise * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    guess = truth * (1 + rng.uniform(-0.1, 0.1, truth.shape))
    return truth, c, Y, guess


def _check_against_oracle(cm, Y, alpha, w=None, tol=TOL, rhs_major=False):
    """Y (B, m) or (B, S, m); returns the device results"""
    sh = cm.shape()
    B = Y.shape[0]
    bp = vp.BatchProblem(sh, Y, weights=w)
    Phi, dPhi = cm.eval_batch(alpha), cm.derivs_batch(alpha)
    got = bp.evaluate_with_basis(alpha, Phi, dPhi)
    assert (np.asarray(got["status"]) == 0).all()
    S = 1 if Y.ndim == 2 else Y.shape[1]
    m = Y.shape[-1]
    W = np.ones(m) if w is None else np.asarray(w)
    for b in range(B):
        p = oracle_problem(cm, Y[b], w=w)
        p.set_params(alpha[b])
        assert p.cached()
        c_ref, r_ref, J_ref = p.linear_coefficients(), p.residuals(), p.jacobian()
        yw = (Y[b] * W).reshape(-1)
        assert np.abs(np.asarray(got["C"][b]) - c_ref).max() <= tol * np.abs(c_ref).max(), "C of problem %d" % b
        assert np.abs(got["r"][b] - r_ref).max() <= tol * np.abs(yw).max(), "r of problem %d" % b
        cs = np.asarray(c_ref).reshape(S, sh.n_basis)
        for k in range(sh.n_params):
            # rounding floor relative to the un-projected column W D_k c (J_k = -P_perp W D_k c cancels)
            dk = np.zeros((sh.n_basis, m))
            for pi, (j, kk) in enumerate(cm.pairs()):
                if kk == k:
                    dk[j] += dPhi[b, pi]
            unproj = max(np.abs((dk * cs[s][:, None]).sum(0) * W).max() for s in range(S))
            bound = tol * np.abs(J_ref[k]).max() + 1e-13 * unproj
            assert np.abs(got["J"][b, k] - J_ref[k]).max() <= bound, "J[%d] of problem %d" % (k, b)
        cost_ref = 0.5 * (r_ref ** 2).sum()
        assert abs(got["cost"][b] - cost_ref) <= tol * max(cost_ref, (yw ** 2).sum() * 1e-6)
    # the trait-level sequence returns the same numbers as the fused call
    bp.set_params_with_basis(alpha, Phi)
    ymax = np.abs(Y * W).max()
    assert np.abs(np.asarray(bp.residuals()) - got["r"]).max() <= 1e-13 * ymax
    assert np.abs(np.asarray(bp.linear_coefficients()).reshape(np.asarray(got["C"]).shape) - got["C"]).max() <= 1e-12 * np.abs(got["C"]).max()
    with pytest.raises(vp.VarproHipError):  # no derivative columns at the current parameters yet
        bp.jacobian()
    J2 = bp.jacobian_with_derivatives(dPhi)
    assert np.abs(np.asarray(J2) - got["J"]).max() <= 1e-13 * np.abs(got["J"]).max()
    assert np.array_equal(np.asarray(bp.jacobian()), np.asarray(J2))  # dPhi is now the handle's
    assert np.array_equal(np.asarray(bp.params()), alpha)
    bp.close()
    return got


# (m = 3000, 4096: four waves per problem; 5000: beyond the kernels of this 8-column shape -> the generic kernels)
@pytest.mark.parametrize("m", [5, 50, 257, 1000, 1024, 3000, 4096, 5000])
@pytest.mark.parametrize("weighted", [False, True])
def test_gauss_lorentz_peaks_match_the_oracle(m, weighted):
    rng = np.random.default_rng(7 + m)
    x = np.linspace(0.0, 10.0, m)
    cm = peaks_model(x)
    _truth, _c, Y, guess = peaks_data(rng, 6, x)
    w = (0.5 + rng.random(m)) if weighted else None
    _check_against_oracle(cm, Y, guess, w)


def test_three_parameter_basis_function():
    rng = np.random.default_rng(11)
    m, B = 400, 5
    x = np.linspace(0.0, 10.0, m)
    cm = voigt_model(x)
    mu, wd, eta = rng.uniform(4, 6, B), rng.uniform(0.5, 1.0, B), rng.uniform(0.2, 0.8, B)
    Y = 20 * pvoigt(x, mu[:, None], wd[:, None], eta[:, None]) + 3 * x / 10.0 + 1.0
    Y = Y + 1e-3 * rng.standard_normal(Y.shape)
    alpha = np.stack([mu, wd, eta], 1) * (1 + rng.uniform(-0.05, 0.05, (B, 3)))
    _check_against_oracle(cm, Y, alpha)


@pytest.mark.parametrize("S", [2, 4, 7])
def test_multiple_right_hand_sides_both_jacobian_branches(S):
    # q = 4: S = 2, 4 take the branch S <= q (src/solvers/levmar/mod.rs:156-171), S = 7 the branch S > q (:172-186)
    rng = np.random.default_rng(13 + S)
    m, B = 300, 3
    x = np.linspace(0.0, 10.0, m)
    cm = peaks_model(x)
    truth, _c, _Y, guess = peaks_data(rng, B, x)
    Phi_t = cm.eval_batch(truth)  # (B, n, m)
    C = rng.uniform(1, 30, (B, S, 3))
    Y = np.einsum("bsn,bnm->bsm", C, Phi_t) + 1e-2 * rng.standard_normal((B, S, m))
    _check_against_oracle(cm, Y, guess, w=0.5 + rng.random(m))


def test_underdetermined_m_smaller_than_n():
    # m = 2 < n = 3: the reference's SVD solve returns the minimum-norm coefficients (src/solvers/levmar/mod.rs:51-54)
    rng = np.random.default_rng(17)
    x = np.array([3.0, 6.5])
    cm = peaks_model(x)
    alpha = np.array([[3.0, 0.7, 6.4, 0.9], [2.8, 0.5, 6.8, 1.1]])
    Y = rng.uniform(1, 5, (2, 2))
    sh = cm.shape()
    bp = vp.BatchProblem(sh, Y)
    got = bp.evaluate_with_basis(alpha, cm.eval_batch(alpha), cm.derivs_batch(alpha))
    for b in range(2):
        Phi = cm.eval_batch(alpha[b:b + 1])[0].T  # (m, n)
        c_ref = np.linalg.lstsq(Phi, Y[b], rcond=None)[0]
        assert np.abs(got["C"][b] - c_ref).max() <= 1e-10 * np.abs(c_ref).max()
        assert np.abs(got["r"][b]).max() <= 1e-10 * np.abs(Y[b]).max()
    bp.close()


def test_device_pointer_mode_keeps_the_callers_arrays():
    import torch
    rng = np.random.default_rng(19)
    m, B = 1024, 32
    x = np.linspace(0.0, 10.0, m)
    cm = peaks_model(x)
    _t, _c, Y, guess = peaks_data(rng, B, x)
    Phi, dPhi = cm.eval_batch(guess), cm.derivs_batch(guess)
    host = vp.BatchProblem(cm.shape(), Y)
    ref = host.evaluate_with_basis(guess, Phi, dPhi)
    dev = torch.device("cuda:0")
    bp = vp.BatchProblem(cm.shape(), torch.as_tensor(Y, device=dev))
    tPhi, tdPhi = torch.as_tensor(Phi, device=dev), torch.as_tensor(dPhi, device=dev)
    got = bp.evaluate_with_basis(torch.as_tensor(guess, device=dev), tPhi, tdPhi)
    for key in ("r", "J", "C", "cost"):
        assert np.array_equal(got[key].cpu().numpy(), np.asarray(ref[key])), key
    # trait-level sequence on the retained pointers
    bp.set_params_with_basis(torch.as_tensor(guess, device=dev), tPhi, tdPhi)
    assert np.abs(bp.residuals().cpu().numpy() - ref["r"]).max() <= 1e-13 * np.abs(Y).max()
    assert np.abs(bp.jacobian().cpu().numpy() - ref["J"]).max() <= 1e-13 * np.abs(ref["J"]).max()
    bp.close()
    host.close()


def test_fp32_handle():
    rng = np.random.default_rng(23)
    m, B = 600, 4
    x = np.linspace(0.0, 10.0, m)
    cm = peaks_model(x, dtype=np.float32)
    _t, _c, Y, guess = peaks_data(rng, B, x)
    Y32, g32 = Y.astype(np.float32), guess.astype(np.float32)
    Phi, dPhi = cm.eval_batch(g32), cm.derivs_batch(g32)
    bp = vp.BatchProblem(cm.shape(), Y32)
    got = bp.evaluate_with_basis(g32, Phi, dPhi)
    cm64 = peaks_model(x)
    for b in range(B):
        p = oracle_problem(cm64, Y32[b].astype(np.float64))
        p.set_params(g32[b].astype(np.float64))
        r_ref = p.residuals()
        assert np.abs(got["r"][b] - r_ref).max() <= 2e-4 * np.abs(Y32[b]).max()
        fit = Y32[b] - got["r"][b]
        assert np.abs(fit - (Y32[b] - r_ref)).max() <= 2e-4 * np.abs(Y32[b]).max()
    bp.close()


def decay_model(x, dtype=np.float64):
    """c1 exp(-x/t1) + c2 exp(-x/t2) + c3 as a CLOSURE model: n = 3, q = 2, p = 2 (the headline shape, caller-evaluated)"""
    return (vp.ClosureModel(["t1", "t2"], x, dtype=dtype)
            .function(["t1"], lambda x, t: np.exp(-x / t)).partial_deriv("t1", lambda x, t: np.exp(-x / t) * x / t ** 2)
            .function(["t2"], lambda x, t: np.exp(-x / t)).partial_deriv("t2", lambda x, t: np.exp(-x / t) * x / t ** 2)
            .invariant_function(lambda x: np.ones_like(x)))


# the register-resident kernels for LONG problems (round 5, vp_ext.hpp: W waves of a workgroup share the columns of one
# problem -- one pass over the caller's arrays): fp64 four waves to 4 096 rows, eight to 8 192; beyond: the generic kernels
@pytest.mark.parametrize("m", [2048, 2500, 4096, 5000, 8192, 10000])
@pytest.mark.parametrize("weighted", [False, True])
def test_long_problems_fp64(m, weighted):
    rng = np.random.default_rng(100 + m)
    B = 3
    x = np.linspace(0.0, 12.5, m)
    cm = decay_model(x)
    tau = np.stack([rng.uniform(0.5, 2.0, B), rng.uniform(2.5, 8.0, B)], 1)
    c = rng.uniform(1.0, 100.0, (B, 3))
    Y = c[:, 0:1] * np.exp(-x / tau[:, 0:1]) + c[:, 1:2] * np.exp(-x / tau[:, 1:2]) + c[:, 2:3]
    Y = Y + 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    w = (0.5 + rng.random(m)) if weighted else None
    _check_against_oracle(cm, Y, tau * (1 + rng.uniform(-0.2, 0.2, tau.shape)), w=w)


@pytest.mark.parametrize("m", [10000, 12345])
@pytest.mark.parametrize("weighted", [False, True])
def test_streamed_evaluation_keeps_the_householder_route_for_badly_scaled_columns(m, weighted):
    """round 6: beyond 8 192 rows a WELL-conditioned problem takes the direct route of the streamed kernel (lane-private pass 1,
    row-local pass 2, vp_blk_ext.hpp); a problem whose factor's diagonal spreads over more than 1e4 keeps the two exact
    Householder passes.  A basis function scaled by 1e-7 (its coefficient by 1e7: r and J are invariant, and so is the
    reference's result, src/solvers/levmar/mod.rs:42-73, 101-201) forces that route: same 1e-10 against the oracle."""
    rng = np.random.default_rng(300 + m)
    B = 3
    x = np.linspace(0.0, 12.5, m)
    sc = 1e-7
    cm = (vp.ClosureModel(["t1", "t2"], x)
          .function(["t1"], lambda x, t: sc * np.exp(-x / t)).partial_deriv("t1", lambda x, t: sc * np.exp(-x / t) * x / t ** 2)
          .function(["t2"], lambda x, t: np.exp(-x / t)).partial_deriv("t2", lambda x, t: np.exp(-x / t) * x / t ** 2)
          .invariant_function(lambda x: np.ones_like(x)))
    tau = np.stack([rng.uniform(0.5, 2.0, B), rng.uniform(2.5, 8.0, B)], 1)
    c = rng.uniform(1.0, 100.0, (B, 3))
    Y = c[:, 0:1] * np.exp(-x / tau[:, 0:1]) + c[:, 1:2] * np.exp(-x / tau[:, 1:2]) + c[:, 2:3]
    Y = Y + 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    w = (0.5 + rng.random(m)) if weighted else None
    _check_against_oracle(cm, Y, tau * (1 + rng.uniform(-0.2, 0.2, tau.shape)), w=w)


# fp32 handles had no resident kernels before round 5 (generic kernels at every shape): one wave to 1 024 rows, four to
# 4 096, eight to 16 384.  Checked against the fp64 oracle of the converted inputs at fp32 resolution x cond(Phi)
@pytest.mark.parametrize("m", [200, 1000, 4096, 5000, 16384, 20000])
def test_long_problems_fp32(m):
    rng = np.random.default_rng(200 + m)
    B = 3
    x = np.linspace(0.0, 12.5, m)
    cm32, cm64 = decay_model(x, dtype=np.float32), decay_model(x)
    tau = np.stack([rng.uniform(0.5, 1.0, B), rng.uniform(4.0, 8.0, B)], 1).astype(np.float32)
    c = rng.uniform(10.0, 100.0, (B, 3))
    Y = c[:, 0:1] * np.exp(-x / tau[:, 0:1]) + c[:, 1:2] * np.exp(-x / tau[:, 1:2]) + c[:, 2:3]
    Y32 = (Y + 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)).astype(np.float32)
    a32 = (tau * (1 + rng.uniform(-0.1, 0.1, tau.shape))).astype(np.float32)
    Phi, dPhi = cm32.eval_batch(a32), cm32.derivs_batch(a32)
    bp = vp.BatchProblem(cm32.shape(), Y32)
    got = bp.evaluate_with_basis(a32, Phi, dPhi)
    assert (np.asarray(got["status"]) == 0).all()
    for b in range(B):
        # the oracle gets the SAME fp32 columns (converted), so that only the solve differs
        ev = lambda a, b=b: Phi[b].astype(np.float64)
        def dv(a, k, b=b):
            out = np.zeros((3, m))
            out[k] = dPhi[b, k].astype(np.float64)
            return out
        p = O.Problem(O.make_shape_desc(3, 2), None, Y32[b].astype(np.float64), external=(ev, dv))
        p.set_params(a32[b].astype(np.float64))
        c_ref, r_ref, J_ref = p.linear_coefficients(), p.residuals(), p.jacobian()
        ymax = np.abs(Y32[b]).max()
        assert np.abs(got["r"][b] - r_ref).max() <= 3e-4 * ymax, (m, np.abs(got["r"][b] - r_ref).max() / ymax)
        assert np.abs(np.asarray(got["C"][b]) - c_ref).max() <= 3e-3 * np.abs(c_ref).max()
        for k in range(2):
            unproj = np.abs(dPhi[b, k].astype(np.float64) * c_ref[k]).max()
            assert np.abs(got["J"][b, k] - J_ref[k]).max() <= 1e-3 * unproj, (m, k)
    bp.close()


# lengths that are not a multiple of anything (odd m: element-wise accesses instead of 16-byte groups), just past a kernel's
# capacity, several right-hand sides per problem -- on the multi-wave kernels (1025 .. 8191) and the streamed one (beyond)
@pytest.mark.parametrize("m,S", [(1025, 1), (3001, 3), (4097, 1), (8191, 2), (9001, 1), (12345, 3)])
@pytest.mark.parametrize("weighted", [False, True])
def test_long_problems_ragged_lengths_and_right_hand_sides(m, S, weighted):
    rng = np.random.default_rng(300 + m + S)
    B = 2
    x = np.linspace(0.0, 12.5, m)
    cm = decay_model(x)
    tau = np.stack([rng.uniform(0.5, 2.0, B), rng.uniform(2.5, 8.0, B)], 1)
    c = rng.uniform(1.0, 100.0, (B, S, 3))
    Y = (c[..., 0:1] * np.exp(-x / tau[:, None, 0:1]) + c[..., 1:2] * np.exp(-x / tau[:, None, 1:2]) + c[..., 2:3])
    Y = Y + 1e-3 * np.abs(Y).max(-1, keepdims=True) * rng.standard_normal(Y.shape)
    if S == 1:
        Y = Y[:, 0]
    w = (0.5 + rng.random(m)) if weighted else None
    _check_against_oracle(cm, Y, tau * (1 + rng.uniform(-0.2, 0.2, tau.shape)), w=w)


def test_best_fit_and_statistics_against_the_oracle():
    rng = np.random.default_rng(29)
    m, B = 500, 4
    x = np.linspace(0.0, 10.0, m)
    cm = peaks_model(x)
    truth, _c, Y, _g = peaks_data(rng, B, x, noise=1e-2)
    w = 0.5 + rng.random(m)
    bp = vp.BatchProblem(cm.shape(), Y, weights=w)
    bp.set_params_with_basis(truth, cm.eval_batch(truth), cm.derivs_batch(truth))
    bf = np.asarray(bp.best_fit())
    st = bp.statistics()
    for b in range(B):
        p = oracle_problem(cm, Y[b], w=w)
        p.set_params(truth[b])
        assert np.abs(bf[b] - p.best_fit()).max() <= 1e-10 * np.abs(Y[b]).max()
        so = p.statistics()
        assert so is not None and st["status"][b] == 0
        assert np.abs(st["cov"][b] - so["cov"]).max() <= 1e-8 * np.abs(so["cov"]).max()
        assert abs(st["reduced_chi2"][b] - so["reduced_chi2"]) <= 1e-10 * so["reduced_chi2"]
        assert np.abs(st["conf_sigma"][b] - so["conf_sigma"]).max() <= 1e-8 * np.abs(so["conf_sigma"]).max()
    bp.close()


def test_entry_points_that_need_a_device_side_model_are_refused():
    x = np.linspace(0.0, 10.0, 64)
    cm = peaks_model(x)
    bp = vp.BatchProblem(cm.shape(), np.ones((2, 64)))
    a = np.tile([3.0, 0.7, 6.4, 0.9], (2, 1))
    r, st = bp.residuals(with_status=True)
    assert r is None and (np.asarray(st) == 2).all()  # residuals() before set_params(): None
    for call in (lambda: bp.set_params(a), lambda: bp.evaluate(a), lambda: bp.fit(a)):
        with pytest.raises(vp.VarproHipError) as e:
            call()
        assert e.value.code == -2  # VP_ERR_UNSUPPORTED
    bp.close()
    # a descriptor handle refuses the external entry points
    mdl = vp.multi_exponential_model(x, [1.0, 3.0])
    bp = vp.BatchProblem(mdl, np.ones((2, 64)), x=x)
    with pytest.raises(vp.VarproHipError):
        bp.set_params_with_basis(a[:, :2], np.ones((2, 3, 64)))
    bp.close()
    # builder validation
    with pytest.raises(vp.VarproHipError):
        vp.BatchProblem(vp.ExternalModel(3, 4, [(0, 0), (0, 0)]), np.ones((2, 64)))  # pair listed twice
    with pytest.raises(vp.VarproHipError):
        vp.BatchProblem(vp.ExternalModel(3, 4, [(3, 0)]), np.ones((2, 64)))  # basis index out of range


def test_nonfinite_basis_is_latched_as_status():
    # a model error / overflow (src/solvers/levmar/mod.rs:61-72: cached = None) for ONE problem of the batch
    x = np.linspace(0.0, 10.0, 200)
    cm = peaks_model(x)
    a = np.tile([3.0, 0.7, 6.4, 0.9], (3, 1))
    Y = cm.eval_batch(a).sum(1)
    Phi = cm.eval_batch(a)
    Phi[1, 0, 17] = np.inf
    bp = vp.BatchProblem(cm.shape(), Y)
    got = bp.evaluate_with_basis(a, Phi, cm.derivs_batch(a))
    assert list(np.asarray(got["status"])) == [0, 1, 0]
    bp.close()


@pytest.mark.parametrize("n,q,npairs", [(1, 1, 1), (2, 5, 7), (5, 2, 6), (6, 3, 5), (7, 3, 12), (8, 8, 16), (4, 6, 6)])
@pytest.mark.parametrize("m", [100, 700])
def test_any_shape_the_header_admits(n, q, npairs, m):
    """shapes inside and outside the table of resident kernels (vp_ext.hpp; the rest runs on the generic kernels reading
    the caller's columns): fixed smooth columns handed over as Phi / dPhi, the oracle gets them through its callbacks"""
    rng = np.random.default_rng(1000 * n + 10 * q + npairs + m)
    B = 3
    x = np.linspace(0.0, 1.0, m)
    allp = [(j, k) for j in range(n) for k in range(q)]
    rng.shuffle(allp)
    pairs = sorted(allp[:npairs], key=lambda jk: rng.random())  # any order
    # every parameter needs a pair for the reference's builder; not for the ABI (a column of J is then zero)
    Phi = np.stack([[np.cos((j + 1) * np.pi * (1.0 + 0.05 * b) * x) + 0.2 * np.sin(3 * (j + 1) * x) for j in range(n)]
                    for b in range(B)])  # nearly orthogonal columns: cond(Phi_w) < 1e2, parity at the plain 1e-10
    dPhi = np.stack([[np.sin((p + 2) * (1.0 + 0.2 * b) * x) * (1 + 0.1 * p) for p in range(npairs)] for b in range(B)])
    Y = rng.standard_normal((B, m)) + Phi.sum(1)
    w = 0.5 + rng.random(m)
    alpha = rng.random((B, q))
    bp = vp.BatchProblem(vp.ExternalModel(n, q, pairs), Y, weights=w)
    got = bp.evaluate_with_basis(alpha, Phi, dPhi)
    assert (np.asarray(got["status"]) == 0).all()
    for b in range(B):
        def dv(a, k, b=b):
            out = np.zeros((n, m))
            for pi, (j, kk) in enumerate(pairs):
                if kk == k:
                    out[j] += dPhi[b, pi]
            return out
        p = O.Problem(O.make_shape_desc(n, q), None, Y[b], w=w, external=(lambda a, b=b: Phi[b], dv))
        p.set_params(alpha[b])
        c_ref, r_ref, J_ref = p.linear_coefficients(), p.residuals(), p.jacobian()
        assert np.linalg.cond(Phi[b].T * w[:, None]) < 1e2
        tol = TOL
        assert np.abs(got["C"][b] - c_ref).max() <= tol * np.abs(c_ref).max()
        assert np.abs(got["r"][b] - r_ref).max() <= TOL * np.abs(Y[b] * w).max()
        for k in range(q):
            unproj = np.abs(dv(None, k) * np.asarray(c_ref)[:, None]).sum(0).max()
            assert np.abs(got["J"][b, k] - J_ref[k]).max() <= tol * np.abs(J_ref[k]).max() + 1e-13 * unproj
    bp.close()
