"""Run-time-descriptor models at the in-between kernel sets added in round 3 -- 4 / 8 / 12 rows per lane (m <= 256 / 512 /
768) and the 4-wave set beyond 1024 rows (m <= 4096) -- for the two shapes the reference's own examples use: the builder-made
double exponential + offset (src/test_helpers/mod.rs:11-72 style, (n, q, p) = (3, 2, 2)) and the O'Leary exp*cos pair with a
shared parameter (shared_test_code/src/models.rs:397-425, (2, 3, 4)).  Evaluation to 1e-10 and the fit against the oracle."""
import numpy as np
import pytest

import varpro_amd as vp
from models import double_exp_builder_model, oleary_model
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _oleary_data(rng, B, m):
    t = np.linspace(0.0, 1.5, m)
    a = np.stack([1.0 * rng.uniform(0.9, 1.1, B), 2.5 * rng.uniform(0.9, 1.1, B), 4.0 * rng.uniform(0.9, 1.1, B)], 1)
    c = np.stack([rng.uniform(4, 8, B), rng.uniform(0.5, 2, B)], 1)
    Y = (c[:, :1] * np.exp(-a[:, 1:2] * t) * np.cos(a[:, 2:3] * t) + c[:, 1:2] * np.exp(-a[:, 0:1] * t) * np.cos(a[:, 1:2] * t))
    Y = Y + 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    return t, Y, a * rng.uniform(0.93, 1.07, (B, 3))


def _dexp_data(rng, B, m):
    x = np.linspace(0.0, 12.5, m)
    tau = np.stack([rng.uniform(0.8, 1.3, B), rng.uniform(2.5, 3.6, B)], 1)
    c = rng.uniform(1, 50, (B, 3))
    Y = c[:, 0:1] * np.exp(-x / tau[:, 0:1]) + c[:, 1:2] * np.exp(-x / tau[:, 1:2]) + c[:, 2:3]
    Y = Y + 1e-3 * np.abs(Y).max(1, keepdims=True) * rng.standard_normal(Y.shape)
    return x, Y, tau * rng.uniform(0.9, 1.15, (B, 2))


@pytest.mark.parametrize("which", ["oleary", "double_exp_builder"])
@pytest.mark.parametrize("m,weighted", [(129, False), (200, True), (256, False), (300, False), (512, True), (600, False), (768, True),
                                        (1100, False), (2000, True), (4096, False)])
def test_evaluation_and_fit_at_the_in_between_sizes(which, m, weighted):
    rng = np.random.default_rng(m + (0 if which == "oleary" else 7))
    B = 24
    if which == "oleary":
        x, Y, guess = _oleary_data(rng, B, m)
        mdl = oleary_model(x, guess[0])
    else:
        x, Y, guess = _dexp_data(rng, B, m)
        mdl = double_exp_builder_model(x, guess[0])
    q = guess.shape[1]
    w = rng.uniform(0.4, 1.8, m) if weighted else None
    bp = vp.BatchProblem(mdl, Y, x=x, weights=w)
    ev = bp.evaluate(guess)
    ref = O.evaluate_batch(mdl, x, Y, guess, w=w, n_threads=4)
    yw = Y if w is None else Y * w
    for b in range(B):
        assert np.abs(ev["C"][b] - ref["C"][b]).max() <= TOL * np.abs(ref["C"][b]).max()
        assert np.abs(ev["r"][b] - ref["r"][b]).max() <= TOL * np.abs(yw[b]).max()
        for k in range(q):
            assert np.abs(ev["J"][b, k] - ref["J"][b, k]).max() <= TOL * np.abs(ref["J"][b, k]).max()
    a, C, rep = bp.fit(guess)
    ar, Cr, rr, _secs = O.fit_batch(mdl, x, Y, guess, w=w, n_threads=4)
    assert ((rep["termination"] > 0) == (rr["termination"] > 0)).all()
    ok = (rep["termination"] > 0) & (np.abs(ar).max(1) < 50.0)
    assert ok.mean() > 0.85
    assert (np.abs(a[ok] - ar[ok]).max(1) <= 1e-5 * np.abs(ar[ok]).max(1)).all()
    assert (np.abs(rep["objective"][ok] - rr["objective"][ok]) <= 1e-8 * rr["objective"][ok]).all()
    bp.close()


@pytest.mark.parametrize("S,m", [(5, 300), (12, 700), (3, 1500)])
def test_oleary_global_fit_at_the_in_between_sizes(S, m):
    # S > 1: the 8 / 12-rows-per-lane sets carry MRHS kernels; beyond 1024 rows (a single-RHS 4-wave set) the global fit runs
    # on the generic kernels
    rng = np.random.default_rng(S * m)
    t, Y, guess = _oleary_data(rng, S, m)
    a0 = np.array([1.0, 2.5, 4.0])
    c = np.stack([rng.uniform(4, 8, S), rng.uniform(0.5, 2, S)], 1)
    Y = (c[:, :1] * np.exp(-a0[1] * t) * np.cos(a0[2] * t) + c[:, 1:2] * np.exp(-a0[0] * t) * np.cos(a0[1] * t))
    Y = Y + 1e-3 * np.abs(Y).max() * rng.standard_normal(Y.shape)
    g = a0 * np.array([1.05, 0.96, 1.04])
    mdl = oleary_model(t, g)
    bp = vp.BatchProblem(mdl, Y[None], x=t)
    ev = bp.evaluate(g[None])
    ref = O.Problem(mdl, t, Y)
    ref.set_params(g)
    assert np.abs(ev["C"][0] - ref.linear_coefficients()).max() <= TOL * np.abs(ref.linear_coefficients()).max()
    assert np.abs(ev["r"][0] - ref.residuals()).max() <= TOL * np.abs(Y).max()
    Jr = ref.jacobian()
    for k in range(3):
        assert np.abs(ev["J"][0, k] - Jr[k]).max() <= 1e-9 * np.abs(Jr[k]).max()
    a, C, rep = bp.fit(g[None])
    rr = ref.fit()
    assert rep["termination"][0] > 0 and rr.termination > 0
    assert np.abs(a[0] - ref.params()).max() <= 1e-6 * np.abs(ref.params()).max()
    assert abs(rep["objective"][0] - rr.objective) <= 1e-8 * rr.objective
    bp.close()
