"""Cases that used to return VP_ERR_UNSUPPORTED (VERDICT round 2, item 7 / ADVICE): more than 65535 problems with several
right-hand sides, a global fit on a shape whose specialised kernel set has no multiple-right-hand-side kernels, best_fit
of a generic global fit."""
import numpy as np
import pytest

import varpro_amd as vp
from oracle import oracle as O
from varpro_amd import synth

pytestmark = pytest.mark.gpu


def test_more_than_65535_problems_with_two_right_hand_sides():
    # B = 70000 problems x S = 2 (small m): the streaming kernels used to index problems by gridDim.y (<= 65535)
    B, S, m = 70000, 2, 32
    d = synth.double_exp_batch(B * S, m=m, noise=1e-3)
    Y = d["Y"].reshape(B, S, m)
    g = d["tau_guess"].reshape(B, S, 2)[:, 0, :].copy()
    mdl = vp.multi_exponential_model(d["x"], g[0])
    bp = vp.BatchProblem(mdl, Y, x=d["x"])
    ev = bp.evaluate(g, want_jacobian=False)
    sel = np.r_[0:8, 65530:65544, B - 8:B]
    for b in sel:
        ref = O.Problem(mdl, d["x"], Y[b])
        ref.set_params(g[b])
        assert np.abs(ev["C"][b] - ref.linear_coefficients()).max() <= 1e-9 * np.abs(ref.linear_coefficients()).max()
        assert np.abs(np.asarray(ev["r"][b]) - ref.residuals()).max() <= 1e-9 * np.abs(Y[b]).max()
    a, C, rep = bp.fit(g)
    assert (rep["termination"] != 0).all()
    for b in sel[::3]:
        ref = O.Problem(mdl, d["x"], Y[b])
        ref.set_params(g[b])
        rr = ref.fit()
        if rr.termination > 0 and rep["termination"][b] > 0:
            assert abs(rep["objective"][b] - rr.objective) <= 1e-6 * rr.objective + 1e-12
    bp.close()


@pytest.mark.parametrize("case", ["f64_m4096", "f32_me5_m1000"])
def test_global_fit_on_shapes_whose_specialised_set_has_no_mrhs_kernels(case):
    # double exponential at 2048 < m <= 4096 (four waves per problem) and the fp32 five-exponential Gram shape have no
    # multiple-right-hand-side kernels of their own: S > 1 handles run on the generic kernels instead of failing
    rng = np.random.default_rng(5)
    if case == "f64_m4096":
        m, S, dt, taus, guess = 4096, 5, np.float64, np.array([1.0, 3.0]), np.array([1.3, 3.9])
        tol = 1e-6
    else:
        m, S, dt, taus, guess = 1000, 4, np.float32, np.array([0.5, 1.5, 3.0, 6.0, 12.0]), None
        guess = taus * np.array([1.03, 0.98, 1.02, 0.97, 1.03])
        tol = 5e-2
    x = 12.5 * np.arange(m) / (m - 1)
    Cm = rng.uniform(1, 10, (S, taus.size + 1))
    Y = sum(Cm[:, k:k + 1] * np.exp(-x[None] / taus[k]) for k in range(taus.size)) + Cm[:, -1:]
    mdl = vp.multi_exponential_model(x.astype(dt), guess.astype(dt), dtype=dt)
    bp = vp.BatchProblem(mdl, Y[None].astype(dt), x=x.astype(dt))
    a, C, rep = bp.fit(guess[None].astype(dt))
    assert rep["termination"][0] > 0 or rep["objective"][0] <= 1e-6 * 0.5 * (Y ** 2).sum()
    assert np.abs(np.sort(a[0]) - taus).max() <= tol * taus.max()
    bf = np.asarray(bp.best_fit())                       # == FitResult::best_fit for multiple right-hand sides
    assert bf.shape == (1, S, m)
    assert np.abs(bf[0] - Y).max() <= (1e-6 if dt == np.float64 else 5e-3) * np.abs(Y).max()
    bp.close()
