"""The generic fallback kernels (vp_generic.hpp): any model descriptor the header admits (n <= 8, q <= 8, p <= 16, any
mix of basis kinds, shared parameters) at any m is ACCEPTED -- never VP_ERR_UNSUPPORTED, never a CPU path -- and agrees
with the oracle.  The reference accepts any SeparableNonlinearModel (src/model/mod.rs:239-363); the register-resident
kernels exist only for the shapes / sizes of the registry."""
import itertools

import numpy as np
import pytest

import varpro_amd as vp
from oracle import oracle as O
from varpro_amd import basis
from varpro_amd.model import SeparableModel

pytestmark = pytest.mark.gpu


def make_model(n, q, x, seed):
    """a descriptor with n basis functions over q parameters: every parameter used at least once, some shared"""
    rng = np.random.default_rng(seed)
    # arities: as many two-parameter kinds as needed to reach q, the rest one-parameter / constant
    ar = [0] * n
    need = q
    for j in range(n):
        if need <= 0:
            break
        a = 2 if need >= 2 and (n - j) * 1 < need else 1
        a = min(a, need)
        ar[j] = a
        need -= a
    assert need == 0, (n, q)
    # leftover basis functions: one constant, the others re-use parameters (shared-parameter columns)
    kinds, params, alpha = [], [], np.zeros(q)
    nxt = 0
    const_used = False
    for j in range(n):
        if ar[j] == 2:
            k = basis.EXP_COS if j % 2 == 0 else basis.SIN_PHASE
            idx = (nxt, nxt + 1)
            alpha[nxt], alpha[nxt + 1] = (0.08 + 0.07 * j, 0.6 + 0.35 * j) if k == basis.EXP_COS else (0.45 + 0.2 * j, 0.3 * j)
            nxt += 2
        elif ar[j] == 1:
            k = basis.EXP_DECAY if j % 3 != 2 else basis.EXP_RATE
            idx = (nxt,)
            alpha[nxt] = 0.6 * 1.8 ** j if k == basis.EXP_DECAY else 0.05 + 0.12 * j
            nxt += 1
        elif not const_used:
            k, idx, const_used = basis.CONST, (), True
        else:  # shared parameters: a second kind over already-used parameters
            i = int(rng.integers(0, q))
            k, idx = (basis.EXP_RATE, (i,)) if j % 2 else (basis.SIN_PHASE, (i, (i + 1) % q))
        kinds.append(k)
        params.append(idx)
    names = ["a%d" % i for i in range(q)]
    return SeparableModel(names, kinds, params, x, alpha), alpha


SHAPES = [(n, q) for n in range(1, 9) for q in range(1, 9) if q <= 2 * n]


@pytest.mark.parametrize("m", [100, 3000])
def test_every_descriptor_shape_is_accepted_and_matches_the_oracle(m):
    x = 10.0 * np.arange(m) / (m - 1) + 0.01
    rng = np.random.default_rng(m)
    n_checked = 0
    for n, q in SHAPES:
        mdl, alpha0 = make_model(n, q, x, seed=17 * n + q)
        p = len(mdl.pairs)
        assert p <= 16
        B = 3
        alpha = alpha0[None, :] * (1.0 + 0.05 * rng.uniform(-1, 1, (B, q)))
        Phi = O.eval_phi(mdl, x, alpha0).T
        c_true = rng.uniform(1, 5, (B, n))
        Y = c_true @ Phi.T + 1e-3 * rng.standard_normal((B, m))
        w = 0.5 + rng.random(m) if (n + q) % 2 else None
        bp = vp.BatchProblem(mdl, Y, x=x, weights=w)          # must not raise VP_ERR_UNSUPPORTED
        got = bp.evaluate(alpha)
        ref = O.evaluate_batch(mdl, x, Y, alpha, w=w, n_threads=4)
        assert np.array_equal(got["status"] == 0, ref["status"] == 0), (n, q)
        yw = Y if w is None else Y * w
        for b in range(B):
            if ref["status"][b] != 0:
                continue
            Pw = O.eval_phi(mdl, x, alpha[b]).T * (1.0 if w is None else w[:, None])
            cond = np.linalg.cond(Pw)
            if cond > 1e9:
                continue  # numerically rank-deficient: the truncation decision is implementation-defined
            tolc = 1e-10 + 3e-14 * cond
            assert np.abs(got["C"][b] - ref["C"][b]).max() <= tolc * np.abs(ref["C"][b]).max(), (n, q, b, cond)
            assert np.abs(got["r"][b] - ref["r"][b]).max() <= 1e-10 * np.abs(yw[b]).max() * max(1.0, cond * 1e-6), (n, q, b)
            for k in range(q):
                dkc = (O.eval_dphi(mdl, x, alpha[b], k) * ref["C"][b][:, None]).sum(0) * (1.0 if w is None else w)
                # J_k = -P_perp (W D_k c) cancels: rounding floor relative to the un-projected column, growing with cond(Phi)
                # (the oracle's U (U^T T) - T form and the device's Q-coordinate form round differently there)
                bound = tolc * np.abs(ref["J"][b, k]).max() + 1e-13 * max(10.0, cond) * np.abs(dkc).max()
                assert np.abs(got["J"][b, k] - ref["J"][b, k]).max() <= bound, (n, q, b, k, cond)
            assert abs(got["cost"][b] - ref["cost"][b]) <= 1e-9 * max(ref["cost"][b], (yw[b] ** 2).sum() * 1e-6)
            n_checked += 1
        # the model surface (eval / eval_partial_deriv) runs the generic Phi kernel
        phi, dphi = bp.basis(alpha)
        assert np.abs(phi[0] - O.eval_phi(mdl, x, alpha[0])).max() <= 1e-13 * max(1.0, np.abs(phi[0]).max())
        bp.close()
    assert n_checked >= 2 * len(SHAPES)


@pytest.mark.parametrize("m", [100, 3000])
@pytest.mark.parametrize("weighted", [False, True])
def test_generic_fit_matches_the_oracle(m, weighted):
    # q = 6 over five basis functions of four different kinds + a constant: no specialised kernel has this shape
    x = 10.0 * np.arange(m) / (m - 1) + 0.01
    kinds = [basis.EXP_DECAY, basis.EXP_DECAY, basis.EXP_COS, basis.SIN_PHASE, basis.CONST]
    params = [(0,), (1,), (2, 3), (4, 5), ()]
    truth = np.array([0.9, 4.0, 0.15, 1.3, 0.8, 0.4])
    mdl = SeparableModel(["t1", "t2", "g", "w1", "w2", "ph"], kinds, params, x, truth)
    rng = np.random.default_rng(5 + m)
    B = 12
    Phi = O.eval_phi(mdl, x, truth).T
    c = rng.uniform(2, 6, (B, 5))
    Y = c @ Phi.T + 2e-3 * rng.standard_normal((B, m))
    guess = truth[None, :] * (1.0 + 0.06 * rng.uniform(-1, 1, (B, 6)))
    w = (0.5 + rng.random(m)) if weighted else None
    bp = vp.BatchProblem(mdl, Y, x=x, weights=w)
    alpha, C, rep, tr = bp.fit_trace(guess, max_rows=8)
    a_ref, C_ref, rep_ref, _ = O.fit_batch(mdl, x, Y, guess, w=w, n_threads=4)
    ok = rep_ref["termination"] > 0
    assert ok.mean() >= 0.8
    assert np.array_equal(rep["termination"] > 0, ok)
    # same trajectory for the leading evaluations, same minimum
    for b in range(B):
        pr = O.Problem(mdl, x, Y[b], w=w)
        pr.set_params(guess[b])
        _r, tr_ref = pr.fit_trace(max_rows=8)
        for i in range(min(4, len(tr_ref), int(rep["n_evals"][b]))):
            if tr_ref[i, 6] < 1e-6 * tr_ref[0, 6]:
                break
            assert np.abs(tr[b, i, :6] - tr_ref[i, :6]).max() <= 1e-7 * np.abs(tr_ref[i, :6]).max(), (b, i)
            assert abs(tr[b, i, 6] - tr_ref[i, 6]) <= 1e-8 * tr_ref[i, 6], (b, i)
    rel_o = np.abs(rep["objective"] - rep_ref["objective"])[ok] / rep_ref["objective"][ok]
    assert rel_o.max() <= 1e-6 and np.median(rel_o) <= 1e-10
    # (a few fits are degenerate -- a decay time runs off to ~1e7, where exp(-t/tau) duplicates the constant column and
    # the objective is flat in tau: both sides reach the same objective there, the parameter itself is not determined)
    sane = ok & (np.abs(a_ref).max(1) < 1e3)
    assert sane.mean() >= 0.6
    # (an ftol-terminated minimiser pins the parameters to ~sqrt(ftol) x conditioning of the 6-parameter problem)
    assert (np.abs(alpha - a_ref)[sane].max(1) <= 1e-5 * np.abs(a_ref)[sane].max(1)).all()
    assert abs(rep["n_evals"][ok].mean() - rep_ref["n_evals"][ok].mean()) <= 0.25 * rep_ref["n_evals"][ok].mean()
    # handle state after the fit + best fit
    assert np.array_equal(np.asarray(bp.params()), alpha)
    r = bp.residuals()
    cost = 0.5 * (r ** 2).sum(1)
    assert (np.abs(cost - rep["objective"])[ok] <= 1e-8 * rep["objective"][ok]).all()
    bf = bp.best_fit()
    assert np.abs(bf[ok] - (C[ok, None, :] * np.stack([O.eval_phi(mdl, x, a).T for a in alpha[ok]])).sum(2)).max() <= 1e-9 * np.abs(Y).max()
    # fit statistics (FitStatistics::try_calculate) through the generic kernel vs the oracle at the device's parameters
    st = bp.statistics()
    for b in np.flatnonzero(sane)[:3]:
        pr = O.Problem(mdl, x, Y[b], w=w)
        pr.set_params(alpha[b])
        ref = pr.statistics()
        assert st["status"][b] == 0 and ref is not None
        assert abs(st["reduced_chi2"][b] - ref["reduced_chi2"]) <= 1e-8 * ref["reduced_chi2"]
        scale = np.sqrt(np.outer(np.diag(ref["cov"]), np.diag(ref["cov"])))
        assert np.abs(st["cov"][b] - ref["cov"]).max() <= 1e-6 * scale.max() and (np.abs(st["cov"][b] - ref["cov"]) <= 1e-5 * scale).all()
        assert np.abs(st["conf_sigma"][b] - ref["conf_sigma"]).max() <= 1e-6 * np.abs(ref["conf_sigma"]).max()
    bp.close()


def test_generic_multiple_right_hand_sides_evaluate_and_fp32():
    # S > 1 on a shape without MRHS kernels: trait-level evaluation (every RHS as its own column), global fit; fp32 handle
    m, S = 700, 5
    x = 10.0 * np.arange(m) / (m - 1) + 0.01
    kinds = [basis.EXP_COS, basis.EXP_RATE, basis.SIN_PHASE, basis.CONST, basis.EXP_DECAY]
    params = [(0, 1), (2,), (3, 4), (), (5,)]
    alpha = np.array([0.2, 1.1, 0.3, 0.7, 0.2, 2.5])
    mdl = SeparableModel(["a", "b", "c", "d", "e", "f"], kinds, params, x, alpha)
    rng = np.random.default_rng(3)
    Phi = O.eval_phi(mdl, x, alpha).T
    Y = rng.uniform(1, 4, (S, 5)) @ Phi.T + 1e-3 * rng.standard_normal((S, m))
    bp = vp.BatchProblem(mdl, Y[None], x=x)
    ev = bp.evaluate(alpha[None] * 1.02)
    ref = O.Problem(mdl, x, Y)
    ref.set_params(alpha * 1.02)
    assert np.abs(ev["r"][0] - ref.residuals()).max() <= 1e-10 * np.abs(Y).max()
    Jr = ref.jacobian()
    for k in range(6):
        assert np.abs(ev["J"][0, k] - Jr[k]).max() <= 1e-9 * np.abs(Jr[k]).max()
    # the global fit of a shape without MRHS kernels: one workgroup per problem walks the S columns (gen_mrhs_fit_kernel)
    a_fit, C_fit, rep, tr = bp.fit_trace(alpha[None] * 1.02, max_rows=10)
    rr, tr_ref = ref.fit_trace(max_rows=10)
    assert rep["termination"][0] > 0 and rr.termination > 0
    for i in range(min(4, len(tr_ref), int(rep["n_evals"][0]))):   # same trajectory for the leading evaluations
        if tr_ref[i, 6] < 1e-6 * tr_ref[0, 6]:
            break
        assert np.abs(tr[0, i, :6] - tr_ref[i, :6]).max() <= 1e-6 * np.abs(tr_ref[i, :6]).max(), i
        assert abs(tr[0, i, 6] - tr_ref[i, 6]) <= 1e-6 * tr_ref[i, 6], i
    assert abs(rep["objective"][0] - rr.objective) <= 1e-6 * rr.objective
    assert np.abs(a_fit[0] - ref.params()).max() <= 1e-4 * np.abs(ref.params()).max()
    assert C_fit.shape == (1, S, 5)
    assert np.abs(C_fit[0] - ref.linear_coefficients()).max() <= 1e-4 * np.abs(C_fit).max()
    r = bp.residuals()
    assert abs(0.5 * (np.asarray(r) ** 2).sum() - rep["objective"][0]) <= 1e-8 * rep["objective"][0]
    bp.close()
    # a batch of two independent global fits, no trace
    bp2 = vp.BatchProblem(mdl, np.stack([Y, Y[::-1]]), x=x)
    a2, C2, rep2 = bp2.fit(np.stack([alpha, alpha]) * 1.02)
    assert (rep2["termination"] > 0).all()
    assert np.abs(a2[0] - a_fit[0]).max() <= 1e-9 and np.abs(a2[1] - a_fit[0]).max() <= 1e-6 * np.abs(a_fit).max()
    bp2.close()
    mdl32 = SeparableModel(["a", "b", "c", "d", "e", "f"], kinds, params, x.astype(np.float32), alpha.astype(np.float32),
                           dtype=np.float32)
    bp32 = vp.BatchProblem(mdl32, Y[:1].astype(np.float32), x=x.astype(np.float32))
    ev32 = bp32.evaluate(alpha[None].astype(np.float32) * 1.02)
    ref1 = O.Problem(mdl, x, Y[0])
    ref1.set_params(alpha * 1.02)
    assert np.abs(ev32["r"][0] - ref1.residuals()).max() <= 5e-4 * np.abs(Y).max()
    a32, c32, rep32 = bp32.fit(alpha[None].astype(np.float32) * 1.02)
    assert rep32["termination"][0] > 0 or rep32["objective"][0] <= 1e-4 * 0.5 * (Y[0] ** 2).sum()
    bp32.close()


def test_generic_global_fit_weighted_and_fp32():
    # the generic one-launch global fit with weights (fp64, against the oracle) and on an fp32 handle (recovers the truth
    # to what fp32 resolves)
    m, S = 600, 7
    x = 8.0 * np.arange(m) / (m - 1) + 0.05
    kinds = [basis.EXP_DECAY, basis.EXP_RATE, basis.EXP_COS, basis.CONST]
    params = [(0,), (1,), (2, 3), ()]
    alpha = np.array([1.7, 0.9, 0.25, 1.3])
    mdl = SeparableModel(["a", "b", "c", "d"], kinds, params, x, alpha)
    rng = np.random.default_rng(21)
    Phi = O.eval_phi(mdl, x, alpha).T
    Ctrue = rng.uniform(1, 5, (S, 4))
    Y = Ctrue @ Phi.T + 1e-3 * rng.standard_normal((S, m))
    w = np.linspace(0.5, 2.0, m)
    guess = alpha * np.array([1.04, 0.97, 1.03, 0.98])
    bp = vp.BatchProblem(mdl, Y[None], x=x, weights=w)
    a_fit, C_fit, rep = bp.fit(guess[None])
    ref = O.Problem(mdl, x, Y, w=w)
    ref.set_params(guess)
    rr = ref.fit()
    assert rep["termination"][0] > 0 and rr.termination > 0
    assert abs(rep["objective"][0] - rr.objective) <= 1e-6 * rr.objective
    assert np.abs(a_fit[0] - ref.params()).max() <= 1e-5 * np.abs(ref.params()).max()
    assert np.abs(C_fit[0] - ref.linear_coefficients()).max() <= 1e-4 * np.abs(C_fit).max()
    bp.close()
    mdl32 = SeparableModel(["a", "b", "c", "d"], kinds, params, x.astype(np.float32), alpha.astype(np.float32), dtype=np.float32)
    Yc = Ctrue @ Phi.T                                            # noise-free: the truth is the minimum
    bp32 = vp.BatchProblem(mdl32, Yc[None].astype(np.float32), x=x.astype(np.float32))
    a32, C32, rep32 = bp32.fit(guess[None].astype(np.float32))
    assert rep32["termination"][0] > 0 or rep32["objective"][0] <= 1e-8 * 0.5 * (Yc ** 2).sum()
    assert np.abs(a32[0] - alpha).max() <= 2e-2 * np.abs(alpha).max()
    bp32.close()
