"""The in-between kernel sets of the double exponential + offset (fp64): 12 rows per lane (512 < m <= 768) and 20 / 24 / 28
rows per lane (1024 < m <= 1280 / 1536 / 1792), and the two-wave sets beyond 2048 rows (20 / 24 / 28 rows per lane on two
waves: m <= 2560 / 3072 / 3584): evaluation, fit and the multiple-right-hand-side path against the oracle -- a
length just above a set's capacity must not pay for twice the rows (or for the spilling 32-rows-per-lane set), and the
row-chunked MRHS streaming kernel must tile any even number of rows per lane (chunks of 4 where 8 does not divide it)."""
import numpy as np
import pytest

import varpro_amd as vp
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _data(rng, B, m, noise=1e-3):
    x = np.linspace(0.0, 12.5, m)
    tau = np.stack([rng.uniform(0.8, 1.3, B), rng.uniform(2.5, 3.6, B)], 1)
    c = rng.uniform(1, 50, (B, 3))
    Y = c[:, 0:1] * np.exp(-x / tau[:, 0:1]) + c[:, 1:2] * np.exp(-x / tau[:, 1:2]) + c[:, 2:3]
    return x, Y + noise * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape), tau


@pytest.mark.parametrize("m,weighted", [(513, False), (520, True), (700, False), (767, True), (768, False), (1026, False), (1100, True),
                                        (1280, False), (1290, True), (1536, False), (1537, False), (1700, True), (1792, False),
                                        (2100, False), (2560, True), (2562, False), (3000, True), (3584, False), (3600, False),
                                        (5000, False), (6144, True), (7000, False), (8192, False)])
def test_single_rhs_evaluation_and_fit(m, weighted):
    rng = np.random.default_rng(m)
    B = 40
    x, Y, tau = _data(rng, B, m)
    w = rng.uniform(0.3, 2.0, m) if weighted else None
    guess = tau * rng.uniform(0.85, 1.2, (B, 2))
    mdl = vp.multi_exponential_model(x, guess[0], offset=True)
    bp = vp.BatchProblem(mdl, Y, x=x, weights=w)
    ev = bp.evaluate(guess)
    ref = O.evaluate_batch(mdl, x, Y, guess, w=w, n_threads=4)
    yw = Y if w is None else Y * w
    for b in range(B):
        assert np.abs(ev["C"][b] - ref["C"][b]).max() <= TOL * np.abs(ref["C"][b]).max()
        assert np.abs(ev["r"][b] - ref["r"][b]).max() <= TOL * np.abs(yw[b]).max()
        for k in range(2):
            assert np.abs(ev["J"][b, k] - ref["J"][b, k]).max() <= TOL * np.abs(ref["J"][b, k]).max()
    a, C, rep = bp.fit(guess)
    ar, Cr, repr_, _secs = O.fit_batch(mdl, x, Y, guess, w=w, n_threads=4)
    ok = (rep["termination"] > 0) & (repr_["termination"] > 0)
    assert ((rep["termination"] > 0) == (repr_["termination"] > 0)).all() and ok.mean() > 0.9
    # parameters: both drivers stop on ftol, where the minimum is flat to sqrt(eps) -- and degenerate fits (a decay time run
    # off to where its exponential is a constant next to the offset) have no defined parameters at all: objective only there
    sane = ok & (np.abs(ar).max(1) < 50.0)
    assert sane.mean() > 0.85
    assert (np.abs(a[sane] - ar[sane]).max(1) <= 1e-5 * np.abs(ar[sane]).max(1)).all()
    assert (np.abs(rep["objective"][sane] - repr_["objective"][sane]) <= 1e-8 * repr_["objective"][sane]).all()
    assert (np.abs(rep["objective"][ok] - repr_["objective"][ok]) <= 1e-5 * repr_["objective"][ok]).all()   # (degenerate valleys)
    bp.close()


@pytest.mark.parametrize("S,m,weighted", [(6, 710, True), (33, 520, False), (7, 768, False), (5, 1200, True), (9, 1400, False), (4, 1790, False)])
def test_multiple_right_hand_sides(S, m, weighted):
    rng = np.random.default_rng(S * m)
    x = np.linspace(0.0, 12.5, m)
    Cm = rng.uniform(1, 50, (S, 3))
    Y = Cm[:, 0:1] * np.exp(-x / 1.0) + Cm[:, 1:2] * np.exp(-x / 3.0) + Cm[:, 2:3]
    Y = Y + 1e-3 * np.abs(Y).max() * rng.standard_normal(Y.shape)
    w = rng.uniform(0.3, 2.0, m) if weighted else None
    guess = np.array([1.3, 3.7])
    mdl = vp.multi_exponential_model(x, guess, offset=True)
    bp = vp.BatchProblem(mdl, Y[None], x=x, weights=w)
    ev = bp.evaluate(guess[None])
    ref = O.Problem(mdl, x, Y, w=w)
    ref.set_params(guess)
    yw = Y if w is None else Y * w
    assert np.abs(ev["C"][0] - ref.linear_coefficients()).max() <= TOL * np.abs(ref.linear_coefficients()).max()
    assert np.abs(ev["r"][0] - ref.residuals()).max() <= TOL * np.abs(yw).max()
    Jr = ref.jacobian()
    for k in range(2):
        assert np.abs(ev["J"][0, k] - Jr[k]).max() <= 1e-10 * np.abs(Jr[k]).max()
    a, C, rep = bp.fit(guess[None])
    rr = ref.fit()
    assert rep["termination"][0] > 0 and rr.termination > 0
    assert np.abs(a[0] - ref.params()).max() <= 1e-6 * np.abs(ref.params()).max()
    assert abs(rep["objective"][0] - rr.objective) <= 1e-8 * rr.objective
    bp.close()


@pytest.mark.parametrize("m,weighted", [(1100, False), (1536, True), (1537, False), (2048, False), (2100, True), (3072, False),
                                        (3500, False), (4096, True), (5000, False), (8192, True)])
def test_single_exponential_beyond_1024_rows(m, weighted):
    # one exponential + offset: 24 / 32 rows per lane up to 2048 rows, the same on two waves up to 4096 -- without these sets
    # the model dropped to the generic kernels above 1024 rows
    rng = np.random.default_rng(m + 1)
    B = 24
    x = np.linspace(0.0, 10.0, m)
    tau = rng.uniform(1.5, 2.5, (B, 1))
    c = rng.uniform(1, 50, (B, 2))
    Y = c[:, 0:1] * np.exp(-x / tau) + c[:, 1:2]
    Y = Y + 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    w = rng.uniform(0.3, 2.0, m) if weighted else None
    guess = tau * rng.uniform(0.8, 1.25, (B, 1))
    mdl = vp.multi_exponential_model(x, guess[0], offset=True)
    bp = vp.BatchProblem(mdl, Y, x=x, weights=w)
    ev = bp.evaluate(guess)
    ref = O.evaluate_batch(mdl, x, Y, guess, w=w, n_threads=4)
    yw = Y if w is None else Y * w
    for b in range(B):
        assert np.abs(ev["C"][b] - ref["C"][b]).max() <= TOL * np.abs(ref["C"][b]).max()
        assert np.abs(ev["r"][b] - ref["r"][b]).max() <= TOL * np.abs(yw[b]).max()
        assert np.abs(ev["J"][b, 0] - ref["J"][b, 0]).max() <= TOL * np.abs(ref["J"][b, 0]).max()
    a, C, rep = bp.fit(guess)
    ar, Cr, rr, _secs = O.fit_batch(mdl, x, Y, guess, w=w, n_threads=4)
    assert (rep["termination"] > 0).all() and (rr["termination"] > 0).all()
    assert (np.abs(a - ar).max(1) <= 1e-6 * np.abs(ar).max(1)).all()
    assert (np.abs(rep["objective"] - rr["objective"]) <= 1e-8 * rr["objective"]).all()
    bp.close()


@pytest.mark.parametrize("m,weighted", [(600, False), (1100, True), (1536, False), (2100, False), (3072, True), (4096, False)])
def test_triple_exponential_at_the_in_between_and_four_wave_sets(m, weighted):
    # three exponentials + offset: 12 / 20 / 24 rows per lane (m <= 768 / 1280 / 1536) and four waves beyond 2048 rows
    # (12 / 16 rows per lane: m <= 3072 / 4096)
    rng = np.random.default_rng(m + 3)
    B = 16
    x = np.linspace(0.0, 25.0, m)
    tau = np.stack([rng.uniform(0.9, 1.1, B), rng.uniform(2.8, 3.3, B), rng.uniform(8.0, 10.0, B)], 1)
    c = rng.uniform(5, 50, (B, 4))
    Y = sum(c[:, j:j + 1] * np.exp(-x / tau[:, j:j + 1]) for j in range(3)) + c[:, 3:4]
    Y = Y + 1e-4 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    w = rng.uniform(0.5, 1.5, m) if weighted else None
    guess = tau * rng.uniform(0.95, 1.05, (B, 3))
    mdl = vp.multi_exponential_model(x, guess[0], offset=True)
    bp = vp.BatchProblem(mdl, Y, x=x, weights=w)
    ev = bp.evaluate(guess)
    ref = O.evaluate_batch(mdl, x, Y, guess, w=w, n_threads=4)
    yw = Y if w is None else Y * w
    for b in range(B):
        assert np.abs(ev["C"][b] - ref["C"][b]).max() <= 1e-9 * np.abs(ref["C"][b]).max()
        assert np.abs(ev["r"][b] - ref["r"][b]).max() <= TOL * np.abs(yw[b]).max()
        for k in range(3):
            assert np.abs(ev["J"][b, k] - ref["J"][b, k]).max() <= 1e-9 * np.abs(ref["J"][b, k]).max()
    a, C, rep = bp.fit(guess)
    ar, Cr, rr, _secs = O.fit_batch(mdl, x, Y, guess, w=w, n_threads=4)
    assert ((rep["termination"] > 0) == (rr["termination"] > 0)).all()
    ok = (rep["termination"] > 0)
    assert ok.mean() > 0.8
    assert (np.abs(rep["objective"][ok] - rr["objective"][ok]) <= 1e-6 * rr["objective"][ok]).all()
    bp.close()
