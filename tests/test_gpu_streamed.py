"""The LENGTH-AGNOSTIC kernels (varpro_amd/csrc/vp_block.hpp; VERDICT round 3, "next" item 6): rows streamed in blocks
through a TSQR-style update of an (n + 1 + p)^2 triangle, any m -- the reference takes any `output_len()`
(/root/reference/src/model/mod.rs:263).  `stream_rows=True` (VP_FLAG_STREAM_ROWS) forces them where a register-resident
set would cover m, so that both run the same problems; beyond the largest resident set the library selects them itself.
  * fit (blk_fit_kernel): the oracle's fit of the same problems -- same success class, objective to 1e-12 median /
    1e-6 max, evaluation counts within 3 for the bulk, the leading trial points of the trajectory to 1e-8;
  * trait-level evaluation (blk_evaluate_kernel: forward pass + exact Householder back-application block by block): c, r,
    J against the oracle at north_star's 1e-10.
Model families: 1 / 2 / 3 exponentials (+ offset), the O'Leary exp*cos pair (run-time descriptor, shared parameter), fp32;
weighted / unweighted, per-problem grids, uniform and non-uniform grids, m not a multiple of anything (element-wise staging
instead of the LDS DMA), m = 100 000."""
import numpy as np
import pytest

import varpro_amd as vp
from models import oleary_model
from oracle import census as CS
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _multiexp(rng, B, m, nexp, offset, uniform=True, noise=1e-3):
    x = np.linspace(0.0, 12.5, m) if uniform else np.sort(rng.random(m)) * 12.5
    base = {1: [2.0], 2: [1.0, 3.0], 3: [0.7, 2.0, 6.0]}[nexp]
    tau = np.stack([rng.uniform(0.85, 1.15, B) * t0 for t0 in base], 1)
    c = rng.uniform(5, 50, (B, nexp + 1))
    Y = sum(c[:, j:j + 1] * np.exp(-x / tau[:, j:j + 1]) for j in range(nexp)) + (c[:, nexp:nexp + 1] if offset else 0.0)
    Y = Y + noise * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    guess = tau * rng.uniform(0.9, 1.12, tau.shape)
    return x, Y, guess


def _check_fit(mdl, x, Y, guess, w=None, min_ok=0.9):
    bp = vp.BatchProblem(mdl, Y, x=x, weights=w, stream_rows=True)
    a, _C, rep = bp.fit(guess)
    bp.close()
    ao, _Co, ro, _s = O.fit_batch(mdl, x if np.ndim(x) == 1 else x[0], Y, guess, w=w, n_threads=8) if np.ndim(x) == 1 else (None,) * 4
    if ao is None:  # per-problem grids: the oracle one problem at a time
        ao, ro = np.empty_like(a), np.zeros(len(Y), dtype=O.REPORT_DTYPE)
        for b in range(len(Y)):
            p = O.Problem(mdl, x[b], Y[b], w=None if w is None else (w[b] if np.ndim(w) == 2 else w))
            p.set_params(guess[b])
            r = p.fit()
            ao[b] = p.params()
            ro[b] = (r.termination, r.n_evals, r.objective)
    res = CS.census(rep, a, ro, ao, max_listed=20)
    assert res["success_class_disagreements"] == 0, res["disagreements"]
    assert (ro["termination"] > 0).mean() >= min_ok
    assert res["objective_rel_diff_median_common_successes"] <= 1e-12
    assert res["objective_rel_diff_max_common_successes"] <= 1e-6
    assert res["share_evals_within_3"] >= 0.7  # (the existing batch contract asks 0.5: counts differ in the rounding-noise tail)
    return res


@pytest.mark.parametrize("nexp,offset", [(1, True), (1, False), (2, True), (2, False), (3, True), (3, False)])
@pytest.mark.parametrize("m,weighted", [(100, False), (1000, True), (1001, False), (1024, False), (3000, True), (10000, False)])
def test_streamed_fit_multiexponential(nexp, offset, m, weighted):
    rng = np.random.default_rng(100 * nexp + m + int(offset))
    B = 48
    x, Y, guess = _multiexp(rng, B, m, nexp, offset)
    w = rng.uniform(0.5, 1.5, m) if weighted else None
    mdl = vp.multi_exponential_model(x, guess[0], offset=offset)
    _check_fit(mdl, x, Y, guess, w, min_ok=0.8 if nexp == 3 else 0.9)


# four exponentials + offset (round 5: a streamed set of its own; the generic kernels before).  cond(Phi) ~ 1e3-1e4 with these
# decay times: the trajectories of two correct drivers part sooner than for two or three exponentials, so the contract is the
# success class and the minimum; evaluation totals within a factor of two
@pytest.mark.parametrize("offset", [True, False])
@pytest.mark.parametrize("m,weighted", [(200, False), (1024, True), (5000, False)])
def test_streamed_fit_four_exponentials(offset, m, weighted):
    rng = np.random.default_rng(4000 + m + int(offset))
    B = 48
    x = np.linspace(0.0, 25.0, m)
    base = [0.4, 1.3, 4.0, 12.0]
    tau = np.stack([rng.uniform(0.9, 1.1, B) * t0 for t0 in base], 1)
    c = rng.uniform(10, 50, (B, 5))
    Y = sum(c[:, j:j + 1] * np.exp(-x / tau[:, j:j + 1]) for j in range(4)) + (c[:, 4:5] if offset else 0.0)
    Y = Y + 1e-4 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    guess = tau * rng.uniform(0.95, 1.05, tau.shape)
    w = (0.5 + rng.random(m)) if weighted else None
    mdl = vp.multi_exponential_model(x, guess[0], offset=offset)
    bp = vp.BatchProblem(mdl, Y, x=x, weights=w)
    a, _C, rep = bp.fit(guess)
    ev = bp.evaluate(guess)
    bp.close()
    ao, _Co, ro, _s = O.fit_batch(mdl, x, Y, guess, w=w, n_threads=8)
    res = CS.census(rep, a, ro, ao, max_listed=20)
    assert res["same_success_class"] >= 0.95, res["disagreements"]
    assert (ro["termination"] > 0).mean() >= 0.9
    assert res["objective_rel_diff_median_common_successes"] <= 1e-9
    assert res["objective_rel_diff_max_common_successes"] <= 1e-5
    # (evaluation counts: 503 vs 373 over 48 fits at m = 5000 -- near the minimum of a cond(Phi) ~ 1e4 problem the step
    # lengths of two correct drivers differ in the second digit; bounded, not matched)
    assert res["sum_evals_device"] <= 2 * res["sum_evals_oracle"] and res["sum_evals_oracle"] <= 2 * res["sum_evals_device"]
    # trait level at the guesses (blk_evaluate_kernel or the resident set, whichever the length selects)
    ref = O.evaluate_batch(mdl, x, Y, guess, w=w, n_threads=8)
    ok = (np.asarray(ev["status"]) == 0) & (ref["status"] == 0)
    assert ok.mean() == 1.0
    yw = np.abs(Y * (1.0 if w is None else w)).max(1)
    assert (np.abs(np.asarray(ev["r"]) - ref["r"]).max(1) / yw).max() <= 1e-9


# five exponentials (+ offset) in fp64 -- the shape of BASELINE configs[4] in double precision (end of round 5: a streamed set of
# its own at one wave per SIMD; the generic kernels before).  cond(Phi) ~ 1e5-1e6: contract as for four exponentials
@pytest.mark.parametrize("offset", [True, False])
@pytest.mark.parametrize("m,weighted", [(300, False), (1024, True), (4096, False)])
def test_streamed_fit_five_exponentials(offset, m, weighted):
    rng = np.random.default_rng(5000 + m + int(offset))
    B = 48
    x = np.linspace(0.0, 40.0, m)
    base = [0.5, 1.5, 4.0, 10.0, 25.0]
    tau = np.stack([rng.uniform(0.9, 1.1, B) * t0 for t0 in base], 1)
    c = rng.uniform(10, 50, (B, 6))
    Y = sum(c[:, j:j + 1] * np.exp(-x / tau[:, j:j + 1]) for j in range(5)) + (c[:, 5:6] if offset else 0.0)
    Y = Y + 1e-5 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    guess = tau * rng.uniform(0.97, 1.03, tau.shape)
    w = (0.5 + rng.random(m)) if weighted else None
    mdl = vp.multi_exponential_model(x, guess[0], offset=offset)
    bp = vp.BatchProblem(mdl, Y, x=x, weights=w)
    a, _C, rep = bp.fit(guess)
    ev = bp.evaluate(guess)
    bp.close()
    ao, _Co, ro, _s = O.fit_batch(mdl, x, Y, guess, w=w, n_threads=8)
    res = CS.census(rep, a, ro, ao, max_listed=20)
    print({k: v for k, v in res.items() if k != "disagreements"})
    # measured: same class on 48 / 48 in all six cases, objective 1e-13 median / 8e-13 max, evaluation totals within 4 %
    assert res["success_class_disagreements"] == 0, res["disagreements"]
    assert (ro["termination"] > 0).mean() >= 0.9
    assert res["objective_rel_diff_median_common_successes"] <= 1e-11
    assert res["objective_rel_diff_max_common_successes"] <= 1e-9
    assert abs(res["sum_evals_device"] - res["sum_evals_oracle"]) <= 0.1 * res["sum_evals_oracle"]
    # trait level at the guesses (blk_evaluate_kernel): north_star's 1e-10 on r and J (measured 8e-13 on J)
    ref = O.evaluate_batch(mdl, x, Y, guess, w=w, n_threads=8)
    ok = (np.asarray(ev["status"]) == 0) & (ref["status"] == 0)
    assert ok.mean() == 1.0
    yw = np.abs(Y * (1.0 if w is None else w)).max(1)
    assert (np.abs(np.asarray(ev["r"]) - ref["r"]).max(1) / yw).max() <= TOL
    jn = np.abs(ref["J"]).max((1, 2))[:, None, None]
    assert float((np.abs(np.asarray(ev["J"]) - ref["J"]) / jn).max()) <= TOL


@pytest.mark.parametrize("nexp,m", [(1, 4100), (2, 6000), (3, 5000)])
def test_streamed_fit_four_waves_per_problem_weighted(nexp, m):
    """launches smaller than the device with >= 2 blocks per wave run FOUR waves per problem (vp_block.hpp, blk_fit_kernel W = 4:
    per-wave rings and carries, merged through LDS by the same stacked QR) -- here with weights (the ring carries three streams)"""
    rng = np.random.default_rng(7 * nexp + m)
    B = 40
    x, Y, guess = _multiexp(rng, B, m, nexp, True)
    w = rng.uniform(0.5, 1.5, m)
    mdl = vp.multi_exponential_model(x, guess[0], offset=True)
    _check_fit(mdl, x, Y, guess, w, min_ok=0.8 if nexp == 3 else 0.9)


def test_streamed_fit_one_and_four_waves_per_problem_agree():
    """the same 40 problems alone (four waves each) and as the head of a batch larger than 16 x the CU count (one wave each):
    different TSQR orders, same fits"""
    rng = np.random.default_rng(21)
    m, B0 = 4608, 40
    x, Y, guess = _multiexp(rng, B0, m, 2, True)
    mdl = vp.multi_exponential_model(x, guess[0])
    bp = vp.BatchProblem(mdl, Y, x=x, stream_rows=True)
    a4, C4, r4 = bp.fit(guess)
    bp.close()
    reps = 16 * 304 // B0 + 2  # (> 16 workgroups' worth per CU on any current part)
    Yb, gb = np.tile(Y, (reps, 1)), np.tile(guess, (reps, 1))
    bp = vp.BatchProblem(mdl, Yb, x=x, stream_rows=True)
    a1, C1, r1 = bp.fit(gb)
    bp.close()
    assert ((r4["termination"] > 0) == (r1["termination"][:B0] > 0)).all() and (r4["termination"] > 0).mean() >= 0.9
    ok = r4["termination"] > 0
    assert (np.abs(r4["objective"][ok] - r1["objective"][:B0][ok]) <= 1e-10 * r4["objective"][ok]).all()
    assert (np.abs(a4[ok] - a1[:B0][ok]).max(1) <= 1e-6 * np.abs(a4[ok]).max(1)).all()
    assert np.abs(r4["n_evals"].astype(int) - r1["n_evals"][:B0].astype(int)).max() <= 8
    # every copy of a problem inside the big batch gets the same answer bit for bit (one wave each, same arithmetic)
    assert (a1.reshape(reps, B0, -1) == a1[:B0][None]).all()


def test_streamed_fit_nonuniform_and_per_problem_grids():
    rng = np.random.default_rng(5)
    B, m = 24, 2500
    x, Y, guess = _multiexp(rng, B, m, 2, True, uniform=False)
    mdl = vp.multi_exponential_model(x, guess[0])
    _check_fit(mdl, x, Y, guess)
    # per-problem grids (stretched copies of one grid) with per-problem weights
    xs = x[None, :] * rng.uniform(0.9, 1.1, (B, 1))
    Ys = np.stack([_multiexp(np.random.default_rng(50 + b), 1, m, 2, True)[1][0] for b in range(B)])
    ws = rng.uniform(0.5, 1.5, (B, m))
    mdl = vp.multi_exponential_model(xs[0], guess[0])
    _check_fit(mdl, xs, Ys, guess, ws)


def test_streamed_fit_trajectory_matches_the_oracle():
    rng = np.random.default_rng(9)
    B, m = 8, 5000
    x, Y, guess = _multiexp(rng, B, m, 2, True)
    mdl = vp.multi_exponential_model(x, guess[0])
    bp = vp.BatchProblem(mdl, Y, x=x, stream_rows=True)
    a, C, rep, tr = bp.fit_trace(guess, max_rows=64)
    bp.close()
    for b in range(B):
        p = O.Problem(mdl, x, Y[b])
        p.set_params(guess[b])
        ro, tro = p.fit_trace(max_rows=64)
        assert (ro.termination > 0) == (rep["termination"][b] > 0)
        rows_dev = int(np.isfinite(tr[b, :, 0]).sum())
        lead = min(6, len(tro), rows_dev)  # (the two may stop an evaluation apart in the rounding-noise tail)
        assert lead >= 3 and abs(rows_dev - len(tro)) <= 8  # (tests/c/test_trait_lm.c uses the same band)
        for i in range(lead):
            for k in range(3):  # alpha_trial (2), ||r||
                assert abs(tr[b, i, k] - tro[i, k]) <= 1e-8 * max(abs(tro[i, k]), 1e-300) + 1e-12, (b, i, k)
        assert np.abs(C[b] - p.linear_coefficients()).max() <= 1e-6 * np.abs(C[b]).max()


def test_streamed_fit_oleary_model():
    rng = np.random.default_rng(11)
    B, m = 32, 4000
    t = np.linspace(0.0, 1.5, m)
    at = np.stack([1.0 * rng.uniform(0.9, 1.1, B), 2.5 * rng.uniform(0.9, 1.1, B), 4.0 * rng.uniform(0.9, 1.1, B)], 1)
    c = np.stack([rng.uniform(4, 8, B), rng.uniform(0.5, 2, B)], 1)
    Y = (c[:, :1] * np.exp(-at[:, 1:2] * t) * np.cos(at[:, 2:3] * t) + c[:, 1:2] * np.exp(-at[:, 0:1] * t) * np.cos(at[:, 1:2] * t))
    Y = Y + 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    guess = at * rng.uniform(0.92, 1.08, at.shape)
    mdl = oleary_model(t, guess[0])
    _check_fit(mdl, t, Y, guess, min_ok=0.8)


def test_one_hundred_thousand_rows():
    rng = np.random.default_rng(13)
    B, m = 6, 100000
    x, Y, guess = _multiexp(rng, B, m, 2, True)
    mdl = vp.multi_exponential_model(x, guess[0])
    bp = vp.BatchProblem(mdl, Y, x=x)  # no flag: beyond every resident set the library streams by itself
    a, C, rep = bp.fit(guess)
    ev = bp.evaluate(a)
    bp.close()
    ao, Co, ro, _ = O.fit_batch(mdl, x, Y, guess, n_threads=6)
    assert ((rep["termination"] > 0) == (ro["termination"] > 0)).all() and (ro["termination"] > 0).all()
    assert (np.abs(rep["objective"] - ro["objective"]) <= 1e-10 * ro["objective"]).all()
    assert (np.abs(a - ao).max(1) <= 1e-7 * np.abs(ao).max(1)).all()
    ref = O.evaluate_batch(mdl, x, Y, a, n_threads=6)
    for b in range(B):
        assert np.abs(ev["r"][b] - ref["r"][b]).max() <= TOL * np.abs(Y[b]).max()
        for k in range(2):
            assert np.abs(ev["J"][b, k] - ref["J"][b, k]).max() <= TOL * np.abs(ref["J"][b, k]).max() + 1e-13 * np.abs(Y[b]).max()


@pytest.mark.parametrize("nexp,offset", [(1, True), (2, True), (2, False), (3, True)])
@pytest.mark.parametrize("m,weighted", [(64, False), (1000, True), (1001, False), (2048, False), (5000, True), (20000, False)])
def test_streamed_evaluation_matches_the_oracle(nexp, offset, m, weighted):
    rng = np.random.default_rng(7 * nexp + m)
    B = 6
    x, Y, guess = _multiexp(rng, B, m, nexp, offset)
    w = rng.uniform(0.5, 1.5, m) if weighted else None
    mdl = vp.multi_exponential_model(x, guess[0], offset=offset)
    bp = vp.BatchProblem(mdl, Y, x=x, weights=w, stream_rows=True)
    ev = bp.evaluate(guess)
    ref = O.evaluate_batch(mdl, x, Y, guess, w=w, n_threads=4)
    yw = Y if w is None else Y * w
    assert (np.asarray(ev["status"]) == 0).all()
    for b in range(B):
        assert np.abs(ev["C"][b] - ref["C"][b]).max() <= TOL * np.abs(ref["C"][b]).max()
        assert np.abs(ev["r"][b] - ref["r"][b]).max() <= TOL * np.abs(yw[b]).max()
        assert abs(ev["cost"][b] - ref["cost"][b]) <= TOL * max(ref["cost"][b], (yw[b] ** 2).sum() * 1e-6)
        for k in range(nexp):
            dkc = (O.eval_dphi(mdl, x, guess[b], k) * ref["C"][b][:, None]).sum(0) * (1.0 if w is None else w)
            bound = TOL * np.abs(ref["J"][b, k]).max() + 1e-13 * np.abs(dkc).max()
            assert np.abs(ev["J"][b, k] - ref["J"][b, k]).max() <= bound
    # trait-level calls: set_params caches r; jacobian recomputes
    bp.set_params(guess)
    assert np.abs(np.asarray(bp.residuals()) - ev["r"]).max() <= 1e-13 * np.abs(yw).max()
    assert np.abs(np.asarray(bp.jacobian()) - ev["J"]).max() <= 1e-13 * np.abs(ev["J"]).max()
    bp.close()


def test_streamed_evaluation_rank_deficient_basis():
    # two equal decay times: Phi has rank 2 of 3 -> the reference's truncated SVD solve (src/solvers/levmar/mod.rs:51-54,
    # pinned at epsilon 1e-8 as in the resident kernels' tests); the Jacobian keeps the full Q
    rng = np.random.default_rng(3)
    m, B = 3000, 3
    x = np.linspace(0.0, 10.0, m)
    Y = 5 * np.exp(-x / 2.0) + 1.0 + 1e-3 * rng.standard_normal((B, m))
    alpha = np.tile([2.0, 2.0], (B, 1))
    mdl = vp.multi_exponential_model(x, alpha[0])
    bp = vp.BatchProblem(mdl, Y, x=x, epsilon=1e-8, stream_rows=True)
    ev = bp.evaluate(alpha)
    bp.close()
    for b in range(B):
        p = O.Problem(mdl, x, Y[b], eps=1e-8)
        p.set_params(alpha[b])
        assert np.abs(ev["C"][b] - p.linear_coefficients()).max() <= 1e-9 * np.abs(p.linear_coefficients()).max()
        assert np.abs(ev["r"][b] - p.residuals()).max() <= 1e-9 * np.abs(Y[b]).max()


def test_fp32_streamed_fit_close_to_the_resident_kernel():
    rng = np.random.default_rng(21)
    B, m = 64, 2000
    x, Y, guess = _multiexp(rng, B, m, 2, True)
    x32, Y32, g32 = x.astype(np.float32), Y.astype(np.float32), guess.astype(np.float32)
    mdl = vp.multi_exponential_model(x32, g32[0], dtype=np.float32)
    res = []
    for stream in (True, False):
        bp = vp.BatchProblem(mdl, Y32, x=x32, stream_rows=stream)
        res.append(bp.fit(g32))
        bp.close()
    (a1, _c1, r1), (a2, _c2, r2) = res
    ok = (r1["termination"] > 0) & (r2["termination"] > 0)
    assert ok.mean() > 0.8
    assert np.median(np.abs(r1["objective"] - r2["objective"])[ok] / r2["objective"][ok]) <= 1e-3


def test_default_selection_streams_beyond_the_resident_sets():
    rng = np.random.default_rng(17)
    B, m = 16, 9000
    x, Y, guess = _multiexp(rng, B, m, 2, True)
    mdl = vp.multi_exponential_model(x, guess[0])
    out = []
    for stream in (False, True):
        bp = vp.BatchProblem(mdl, Y, x=x, stream_rows=stream)
        out.append(bp.fit(guess))
        bp.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][2]["n_evals"], out[1][2]["n_evals"])
