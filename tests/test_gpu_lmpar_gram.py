"""The LM step of the Gram fit kernel (vp_fitg.hpp) solves MINPACK's trust-region sub-problem on the Cholesky factor of
R^T R + par D^2 (lmpar_chol, vp_fit.hpp) where MINPACK's lmpar / qrsolv rotate [R; sqrt(par) D] with Givens rotations
(levenberg-marquardt 0.14, determine_lambda_and_parameter_update; call site src/solvers/levmar/mod.rs:247).  Same Newton
iteration on par, same bounds, same exits -- so on the same factor the two must return the same par and the same step up to
the conditioning of R^T R.  vp_debug_lmpar_gram runs the device routine once per record; the oracle exports its lmpar."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sl

from oracle import oracle as O
from varpro_amd import _lib

pytestmark = pytest.mark.gpu


def _device(R, ipvt, diag, qtb, delta, par):
    lib = _lib.load()
    B, q = diag.shape
    out = np.empty((B, q + 2))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    R, ipvt, diag, qtb = (np.ascontiguousarray(R, dtype=np.float64), np.ascontiguousarray(ipvt, dtype=np.int32),
                          np.ascontiguousarray(diag, dtype=np.float64), np.ascontiguousarray(qtb, dtype=np.float64))
    delta, par = np.ascontiguousarray(delta, dtype=np.float64), np.ascontiguousarray(par, dtype=np.float64)
    _lib.check(lib.vp_debug_lmpar_gram(B, q, p(R), p(ipvt), p(diag), p(qtb), p(delta), p(par), p(out)))
    return out[:, 0], out[:, 1], out[:, 2:]


def _cases(q, B, rng, cond_exp_max):
    """pivoted QR factors of random Jacobians with column scales and near-dependent columns (cond up to 10^cond_exp_max)"""
    R = np.zeros((B, q, q)); ip = np.zeros((B, q), dtype=np.int32); dg = np.zeros((B, q)); qb = np.zeros((B, q))
    delta = np.zeros(B); par = np.zeros(B); cond = np.zeros(B)
    for b in range(B):
        J = rng.standard_normal((8 * q, q)) * 10.0 ** rng.uniform(-2, 2, q)
        if q > 1 and b % 3 == 0:  # a nearly dependent pair of columns
            J[:, 1] = J[:, 0] * rng.uniform(0.5, 2.0) + 10.0 ** rng.uniform(-cond_exp_max, -1) * J[:, 1]
        f = rng.standard_normal(8 * q) * 10.0 ** rng.uniform(-1, 1)
        Q, Rb, P = sl.qr(J, mode="economic", pivoting=True)
        R[b], ip[b], qb[b] = np.triu(Rb), P, Q.T @ f
        dg[b] = np.linalg.norm(J, axis=0)
        cond[b] = np.linalg.cond(Rb / dg[b][P][None, :])
        gn = np.linalg.norm(dg[b] * np.linalg.lstsq(J, f, rcond=None)[0])
        delta[b] = gn * 10.0 ** rng.uniform(-3, 0.5)          # from deep inside the trust region to a full Gauss-Newton step
        par[b] = 0.0 if b % 2 == 0 else 10.0 ** rng.uniform(-4, 2)
    return R, ip, dg, qb, delta, par, cond


@pytest.mark.parametrize("q", [2, 3, 5])
def test_lmpar_on_cholesky_factors_matches_minpack_lmpar(q):
    rng = np.random.default_rng(100 + q)
    B = 600
    R, ip, dg, qb, delta, par, cond = _cases(q, B, rng, cond_exp_max=5)
    par_d, dx_d, step_d = _device(R, ip, dg, qb, delta, par)
    worst = 0.0
    n_active = n_checked = 0
    for b in range(B):
        par_o, step_o, dx_o = O.lmpar(R[b], ip[b], dg[b], qb[b], delta[b], par[b])
        # stated bound: the Cholesky route works on R^T R (conditioning squared, in the diag-scaled metric the step is measured in)
        tol = 1e-12 + 50.0 * cond[b] ** 2 * np.finfo(np.float64).eps
        if tol > 1e-3:
            continue  # (cond > 3e5 in the scaled metric: the bound says nothing; cfg4's Jacobians are dropped-column-regularised below that)
        n_checked += 1
        n_active += par_o > 0
        assert (par_o == 0.0) == (par_d[b] == 0.0), (b, par_o, par_d[b])
        scale_s = np.abs(dg[b] * step_o).max()
        e_par = abs(par_d[b] - par_o) / max(par_o, 1e-300) if par_o > 0 else 0.0
        e_step = np.abs(dg[b] * (step_d[b] - step_o)).max() / scale_s
        e_dx = abs(dx_d[b] - dx_o) / dx_o
        worst = max(worst, max(e_par, e_step, e_dx) / tol)
        assert e_par <= tol and e_step <= tol and e_dx <= tol, (b, cond[b], e_par, e_step, e_dx, tol)
    assert n_checked >= 0.8 * B and n_active >= n_checked // 4 and n_active <= n_checked - n_checked // 10  # both branches exercised
    print("q", q, "records", n_checked, "of", B, "with par > 0:", n_active, "worst error / bound", worst, "max cond", cond.max())


def test_lmpar_on_cholesky_factors_rank_deficient_factor():
    # a factor whose trailing diagonal is exactly zero (what gram_to_qr leaves for a dropped column): MINPACK zeroes the
    # Gauss-Newton components beyond the numerical rank and takes parl = 0; the regularised factor is still positive definite
    rng = np.random.default_rng(7)
    q, B = 5, 200
    R, ip, dg, qb, delta, par, cond = _cases(q, B, rng, cond_exp_max=3)
    R[:, q - 1, q - 1] = 0.0
    R[::2, q - 2, q - 2:] = 0.0
    par_d, dx_d, step_d = _device(R, ip, dg, qb, delta, par)
    n_dwarf = 0
    for b in range(B):
        par_o, step_o, dx_o = O.lmpar(R[b], ip[b], dg[b], qb[b], delta[b], par[b])
        assert np.isfinite(par_d[b]) and np.isfinite(step_d[b]).all()
        assert (par_o == 0.0) == (par_d[b] == 0.0)
        tol = 1e-7  # (factors of cond <= 1e3 in the scaled metric: 50 cond^2 eps)
        # (a par at the dwarf end of MINPACK's bracket -- 1e-22 where R^T R is of order 1 -- regularises nothing: the system
        # is singular to working precision and its solution along the null space is whatever rounding makes it, in either
        # routine; the LM loop judges such a step by the residual it produces.  Compared: every record whose par is 0 or a
        # number that regularises.)
        par_scale = (np.abs(np.diag(R[b])).max() / dg[b].max()) ** 2
        if par_o > 0 and par_o <= 1e-12 * par_scale:
            n_dwarf += 1
            continue
        if par_o > 0:
            assert abs(par_d[b] - par_o) <= tol * par_o, (b, par_o, par_d[b])
        assert np.abs(dg[b] * (step_d[b] - step_o)).max() <= tol * max(np.abs(dg[b] * step_o).max(), 1e-300), b
    assert n_dwarf <= B // 10
