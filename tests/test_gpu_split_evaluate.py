"""The split evaluate kernel (vp_kernels.hpp, eval2_split: full-length unweighted problems of the multi-exponential + offset
families on a uniform grid, m == 1024) against the oracle and against the kernels it replaces there.

Phase 1 factors [exp | y] with the constant column implicit (c, cost, r); phase 2 rebuilds the derivative columns and
carries them through Q^T / Q (J).  set_params alone and set_params + residuals run phase 1 only.  Tolerances:
north_star's 1e-10 relative to max|c|, max|y|, max|J_k| (src/solvers/levmar/mod.rs:42-73, 101-201)."""
import numpy as np
import pytest

import varpro_amd as vp
from oracle import oracle as O
from varpro_amd import synth

pytestmark = pytest.mark.gpu

TOL = 1e-10
M = 1024
TAUS = [1.0, 3.0, 7.5]


def _batch(n_exp, B=24, noise=1e-3, seed_off=0):
    d = synth.multi_exp_batch(B, n_exp, M, TAUS[:n_exp], noise=noise, spread=0.1, guess_spread=0.1)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], offset=True)
    return d, mdl


def _assert_eval_matches(ev, ref, d, n_exp, want_r=True, want_J=True):
    B = d["Y"].shape[0]
    assert (np.asarray(ev["status"]) == 0).all() and (ref["status"] == 0).all()
    for b in range(B):
        ymax = np.abs(d["Y"][b]).max()
        assert np.abs(ev["C"][b] - ref["C"][b]).max() <= TOL * np.abs(ref["C"][b]).max(), "c of problem %d" % b
        assert abs(ev["cost"][b] - ref["cost"][b]) <= TOL * max(ref["cost"][b], (d["Y"][b] ** 2).sum() * 1e-6)
        if want_r:
            assert np.abs(ev["r"][b] - ref["r"][b]).max() <= TOL * ymax, "r of problem %d" % b
        if want_J:
            for k in range(n_exp):
                assert np.abs(ev["J"][b, k] - ref["J"][b, k]).max() <= TOL * np.abs(ref["J"][b, k]).max(), \
                    "J[%d] of problem %d" % (k, b)


@pytest.mark.parametrize("n_exp", [1, 2, 3])
def test_split_kernel_matches_oracle_in_every_output_mode(n_exp):
    d, mdl = _batch(n_exp)
    ref = O.evaluate_batch(mdl, d["x"], d["Y"], d["tau_guess"], n_threads=4)
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    _assert_eval_matches(bp.evaluate(d["tau_guess"]), ref, d, n_exp)                                   # r + J (both phases)
    _assert_eval_matches(bp.evaluate(d["tau_guess"], want_jacobian=False), ref, d, n_exp, want_J=False)  # phase 1, r out
    ev0 = bp.evaluate(d["tau_guess"], want_residuals=False, want_jacobian=False)                       # phase 1, c / cost only
    _assert_eval_matches(ev0, ref, d, n_exp, want_r=False, want_J=False)
    # the trait-level sequence set_params -> residuals -> jacobian returns the same numbers
    bp.set_params(d["tau_guess"])
    assert np.abs(np.asarray(bp.residuals()) - ref["r"]).max() <= TOL * np.abs(d["Y"]).max()
    J = np.asarray(bp.jacobian())
    for k in range(n_exp):
        assert np.abs(J[:, k] - ref["J"][:, k]).max() <= TOL * np.abs(ref["J"][:, k]).max()
    bp.close()


@pytest.mark.parametrize("n_exp", [1, 2, 3])
def test_split_kernel_equals_the_unsplit_kernel_to_rounding(n_exp):
    # grid_recurrence=False keeps the handle off the uniform-grid kernels: per-row exponentials, all columns in one sweep
    d, mdl = _batch(n_exp, B=16)
    a = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    b = vp.BatchProblem(mdl, d["Y"], x=d["x"], grid_recurrence=False)
    ea, eb = a.evaluate(d["tau_guess"]), b.evaluate(d["tau_guess"])
    ymax = np.abs(d["Y"]).max()
    assert np.abs(ea["r"] - eb["r"]).max() <= 1e-12 * ymax
    assert np.abs(ea["C"] - eb["C"]).max() <= 1e-11 * np.abs(eb["C"]).max()
    for k in range(n_exp):
        assert np.abs(ea["J"][:, k] - eb["J"][:, k]).max() <= 1e-11 * np.abs(eb["J"][:, k]).max()
    a.close()
    b.close()


def test_split_kernel_rank_deficient_basis_takes_the_truncated_branch():
    # two equal decay times: Phi has rank 2 of 3, the reference's SVD drops the singular value below epsilon and returns
    # the minimum-norm coefficients; phase 1 must hand the range(Q)-part of the residual (e) on to the back-application
    d, mdl = _batch(2, B=8)
    alpha = d["tau_guess"].copy()
    alpha[:, 1] = alpha[:, 0]
    # (an explicit epsilon: with the default, machine epsilon, the rounding-level singular value of two equal columns is
    # kept and the solve is ill-posed in every implementation)
    ref = O.evaluate_batch(mdl, d["x"], d["Y"], alpha, eps=1e-8, n_threads=2)
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"], epsilon=1e-8)
    ev = bp.evaluate(alpha)
    assert (np.asarray(ev["status"]) == ref["status"]).all()
    ok = ref["status"] == 0
    assert ok.any()
    for b in np.nonzero(ok)[0]:
        assert np.abs(ev["r"][b] - ref["r"][b]).max() <= 1e-9 * np.abs(d["Y"][b]).max()
        assert np.abs(ev["C"][b] - ref["C"][b]).max() <= 1e-8 * np.abs(ref["C"][b]).max()
        assert abs(ev["cost"][b] - ref["cost"][b]) <= 1e-9 * ref["cost"][b]
    bp.close()


def test_split_kernel_flags_non_finite_parameters_per_problem():
    d, mdl = _batch(2, B=6)
    alpha = d["tau_guess"].copy()
    alpha[1, 0] = 0.0       # exp(-t/0): NaN column -> residuals() == None for this problem only
    alpha[4, 1] = np.nan
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    ev = bp.evaluate(alpha)
    st = np.asarray(ev["status"])
    assert st[1] != 0 and st[4] != 0
    good = [0, 2, 3, 5]
    assert (st[good] == 0).all()
    ref = O.evaluate_batch(mdl, d["x"], d["Y"][good], alpha[good], n_threads=2)
    for i, b in enumerate(good):
        assert np.abs(ev["r"][b] - ref["r"][i]).max() <= TOL * np.abs(d["Y"][b]).max()
    bp.close()


def test_weighted_and_shorter_problems_keep_their_kernels_and_agree():
    # the dispatch: weights or m < 1024 -> the one-sweep kernel; same oracle, same tolerances
    d, mdl = _batch(2, B=8)
    w = 0.5 + np.random.default_rng(3).random(M)
    ref = O.evaluate_batch(mdl, d["x"], d["Y"], d["tau_guess"], w=w, n_threads=2)
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"], weights=w)
    ev = bp.evaluate(d["tau_guess"])
    for b in range(8):
        assert np.abs(ev["r"][b] - ref["r"][b]).max() <= TOL * np.abs(d["Y"][b] * w).max()
        for k in range(2):
            assert np.abs(ev["J"][b, k] - ref["J"][b, k]).max() <= TOL * np.abs(ref["J"][b, k]).max()
    bp.close()
    m2 = 1000
    x2 = d["x"][:m2]
    mdl2 = vp.multi_exponential_model(x2, d["tau_guess"][0], offset=True)
    ref2 = O.evaluate_batch(mdl2, x2, d["Y"][:, :m2], d["tau_guess"], n_threads=2)
    bp2 = vp.BatchProblem(mdl2, d["Y"][:, :m2].copy(), x=x2)
    ev2 = bp2.evaluate(d["tau_guess"])
    for b in range(8):
        assert np.abs(ev2["r"][b] - ref2["r"][b]).max() <= TOL * np.abs(d["Y"][b]).max()
    bp2.close()


def test_split_kernel_at_batch_scale_properties():
    # B = 65 536 (the bench workload): projector identities that do not need the oracle -- Phi^T r = 0 and J_k^T Phi = 0
    # (r and every Kaufman column lie in the orthogonal complement of range(Phi)), cost = 0.5 ||r||^2
    torch = pytest.importorskip("torch")
    B = 65536
    d = synth.double_exp_batch(B, m=M, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    dev = torch.device("cuda", 0)
    Y = torch.from_numpy(d["Y"]).to(dev)
    g = torch.from_numpy(d["tau_guess"]).to(dev)
    bp = vp.BatchProblem(mdl, Y, x=torch.from_numpy(d["x"]).to(dev))
    ev = bp.evaluate(g)
    phi, _ = bp.basis(g)
    ok = ev["status"] == 0
    assert int(ok.sum()) >= B - 8
    r, J = ev["r"][ok], ev["J"][ok]
    phi = phi[ok]
    ynorm = torch.linalg.vector_norm(Y[ok], dim=1)
    pnorm = torch.linalg.vector_norm(phi, dim=2)
    ptr = torch.einsum("bnm,bm->bn", phi, r).abs() / (pnorm * ynorm[:, None])
    assert float(ptr.max()) <= 1e-11
    jn = torch.linalg.vector_norm(J, dim=2)
    ptj = torch.einsum("bnm,bkm->bnk", phi, J).abs() / (pnorm[:, :, None] * jn[:, None, :])
    assert float(ptj.max()) <= 1e-9
    cost = 0.5 * (r * r).sum(dim=1)
    assert float(((cost - ev["cost"][ok]).abs() / cost).max()) <= 1e-11
    bp.close()
