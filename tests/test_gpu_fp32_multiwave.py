"""fp32 scalar type and multi-wave groups (problems whose columns do not fit one wave's registers):
BASELINE configs[4] (five exponentials + offset, fp32, m = 4096) and fp64 problems with m up to 4096."""
import numpy as np
import pytest

import varpro_amd as vp
from models import double_exp_builder_model
from oracle import oracle as O
from varpro_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m", [4096, 2500, 1025])
def test_fp64_four_waves_per_problem_matches_oracle(m):
    # m > 1024 in fp64: the rows of one problem are spread over a workgroup of 4 waves (LDS-exchanged reductions)
    d = synth.double_exp_batch(10, m=m, noise=1e-3)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    ev = bp.evaluate(d["tau_guess"])
    ref = O.evaluate_batch(mdl, d["x"], d["Y"], d["tau_guess"], n_threads=4)
    assert (ev["status"] == 0).all()
    for b in range(10):
        assert np.abs(ev["C"][b] - ref["C"][b]).max() <= 1e-10 * np.abs(ref["C"][b]).max()
        assert np.abs(ev["r"][b] - ref["r"][b]).max() <= 1e-10 * np.abs(d["Y"][b]).max()
        for k in range(2):
            assert np.abs(ev["J"][b, k] - ref["J"][b, k]).max() <= 1e-10 * np.abs(ref["J"][b, k]).max()
    alpha, C, rep = bp.fit(d["tau_guess"])
    a_ref, C_ref, rep_ref, _ = O.fit_batch(mdl, d["x"], d["Y"], d["tau_guess"], n_threads=4)
    ok = (rep_ref["termination"] > 0)
    assert ((rep["termination"] > 0) == ok).all()
    rel = np.abs(rep["objective"] - rep_ref["objective"])[ok] / rep_ref["objective"][ok]
    assert rel.max() <= 1e-6 and np.median(rel) <= 1e-11
    phi, dphi = bp.basis(d["tau_guess"])
    assert np.abs(phi[3] - O.eval_phi(mdl, d["x"], d["tau_guess"][3])).max() <= 1e-15
    assert np.abs(np.asarray(bp.best_fit())[ok] + np.asarray(bp.residuals())[ok] - d["Y"][ok]).max() <= 1e-9 * np.abs(d["Y"]).max()
    bp.close()


@pytest.mark.parametrize("m", [100, 400, 512, 1024, 1100, 2048])   # (sets of 2 / 8 / 16 / 32 rows per lane)
def test_fp32_double_exponential(m):
    # ScalarType = f32 (src/model/builder/mod.rs:66).  Tolerances: eps32 * cond(Phi) ~ 6e-8 * 1e2..1e3
    d = synth.double_exp_batch(16, m=m, noise=1e-3)
    mdl32 = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
    mdl64 = double_exp_builder_model(d["x"], d["tau_guess"][0])
    bp = vp.BatchProblem(mdl32, d["Y"].astype(np.float32), x=d["x"].astype(np.float32))
    g32 = d["tau_guess"].astype(np.float32)
    ev = bp.evaluate(g32)
    assert ev["r"].dtype == np.float32 and ev["C"].dtype == np.float32
    ref = O.evaluate_batch(mdl64, d["x"].astype(np.float32).astype(np.float64), d["Y"].astype(np.float32).astype(np.float64),
                           g32.astype(np.float64), n_threads=4)
    assert (ev["status"] == 0).all()
    for b in range(16):
        assert np.abs(ev["C"][b] - ref["C"][b]).max() <= 2e-3 * np.abs(ref["C"][b]).max()
        assert np.abs(ev["r"][b] - ref["r"][b]).max() <= 2e-5 * np.abs(d["Y"][b]).max()
        for k in range(2):
            assert np.abs(ev["J"][b, k] - ref["J"][b, k]).max() <= 2e-3 * np.abs(ref["J"][b, k]).max()
    alpha, C, rep = bp.fit(g32)
    a_ref, C_ref, rep_ref, _ = O.fit_batch(mdl64, d["x"], d["Y"], d["tau_guess"], n_threads=4)
    good = (rep["termination"] > 0) & (rep_ref["termination"] > 0)
    assert good.mean() >= 0.8
    # fp32 objective floor: the noise level 1e-3 is far above eps32, so the minima agree to ~1e-3 relative
    rel = np.abs(rep["objective"] - rep_ref["objective"])[good] / rep_ref["objective"][good]
    assert np.median(rel) <= 1e-2
    assert np.median(np.abs(alpha - a_ref)[good] / np.abs(a_ref)[good]) <= 1e-2
    bp.close()


def test_fp32_five_exponentials_config4_shape():
    # BASELINE configs[4]: m = 4096, n = 6 (5 exp + offset), q = 5, fp32, 4 waves per problem.
    # cond(Phi) >= 1e6 for five exponentials: the linear coefficients are NOT determined in fp32 (SURVEY.md 8(d):
    # "document, don't hide"); the projected residual and the cost are (they depend on range(Phi) only).
    taus = [0.5, 1.5, 3.0, 6.0, 12.0]
    d = synth.multi_exp_batch(6, 5, 4096, taus, noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
    mdl32 = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
    bp = vp.BatchProblem(mdl32, d["Y"], x=d["x"])
    ev = bp.evaluate(d["tau_guess"])
    mdl64 = vp.multi_exponential_model(d["x"].astype(np.float64), d["tau_guess"][0].astype(np.float64))
    ref = O.evaluate_batch(mdl64, d["x"].astype(np.float64), d["Y"].astype(np.float64),
                           d["tau_guess"].astype(np.float64), n_threads=4)
    assert (ev["status"] == 0).all()
    for b in range(6):
        ynorm = np.abs(d["Y"][b]).max()
        assert np.abs(ev["r"][b] - ref["r"][b]).max() <= 1e-4 * ynorm
        assert abs(ev["cost"][b] - ref["cost"][b]) <= 5e-2 * ref["cost"][b] + 1e-8 * ynorm ** 2 * 4096
        # J_k = -P_perp D_k c inherits the ill-conditioning of c: orthogonality to range(Phi) is what fp32 can promise
    phi, _ = bp.basis(d["tau_guess"])
    pn = np.linalg.norm(phi.astype(np.float64), axis=2)
    ortho = np.abs(np.einsum("bjm,bm->bj", phi.astype(np.float64), ev["r"].astype(np.float64))) / (
        pn * np.linalg.norm(d["Y"].astype(np.float64), axis=1)[:, None])
    assert ortho.max() <= 5e-5
    alpha, C, rep = bp.fit(d["tau_guess"])
    assert np.isfinite(alpha).all()
    # the fit must not increase the cost of any problem it reports as successful
    ok = rep["termination"] > 0
    assert (rep["objective"][ok] <= ev["cost"][ok] * (1 + 1e-3)).all()
    bp.close()


def test_fp32_config4_full_size_properties():
    # BASELINE configs[4] at FULL size: B = 8192 problems, m = 4096, five exponentials + offset, fp32, four waves
    # per problem.  What fp32 can promise (cond(Phi) >= 1e6): r orthogonal to range(Phi), y = Phi c + r, a fit that
    # never increases the cost of a problem it reports as successful, consistent summary.
    taus = [0.5, 1.5, 3.0, 6.0, 12.0]
    B, m = 8192, 4096
    d = synth.multi_exp_batch(B, 5, m, taus, noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
    mdl32 = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
    bp = vp.BatchProblem(mdl32, d["Y"], x=d["x"])
    ev = bp.evaluate(d["tau_guess"], want_jacobian=False)
    assert (ev["status"] == 0).mean() > 0.99
    sub = np.arange(0, B, 64)                      # host-side checks on every 64th problem (fp64 accumulation)
    phi, _ = bp.basis(d["tau_guess"])
    P = phi[sub].astype(np.float64)
    r = ev["r"][sub].astype(np.float64)
    Y = d["Y"][sub].astype(np.float64)
    ortho = np.abs(np.einsum("bjm,bm->bj", P, r)) / (np.linalg.norm(P, axis=2) * np.linalg.norm(Y, axis=1)[:, None])
    assert ortho.max() <= 1e-4
    recon = np.einsum("bjm,bj->bm", P, ev["C"][sub].astype(np.float64)) + r
    assert np.abs(recon - Y).max() <= 2e-3 * np.abs(Y).max()   # Phi c in fp32 with |c| up to cond(Phi) * |y|
    alpha, C, rep = bp.fit(d["tau_guess"])
    ok = rep["termination"] > 0
    # (the fp32 restatement of the reference algorithm fails on ~12-15 % of this problem set as well:
    # test_fp32_config4_fit_against_the_fp32_and_fp64_oracles)
    assert ok.mean() >= 0.8
    assert (rep["objective"][ok] <= ev["cost"][ok] * (1 + 1e-3) + 1e-6).all()
    s = bp.summary()
    assert s[1] == ok.sum() and s[1] + s[2] == B and s[3] == rep["n_evals"].sum()
    bp.close()


def test_fp32_config4_fit_against_the_fp32_and_fp64_oracles():
    # BASELINE configs[4] (five exponentials + offset, fp32, m = 4096) on a 384-problem sample, against
    #   (a) the fp32-storage build of the oracle (oracle/Makefile: the reference ALGORITHM in single precision) and
    #   (b) the fp64 oracle on the same (float -> double converted) inputs.
    # cond(Phi) of five exponentials is >= 1e6, beyond what fp32 resolves: a share of the fits ends with
    # TerminationReason::User (non-finite evaluation) -- in the fp32 restatement of the reference just as on the device.
    # Asserted: the device fails no more often than the fp32 oracle (+ 5 points), agrees with it on success / failure
    # for the bulk of the problems, and where both succeed reaches an objective that is as good (fp32 tolerance) -- and
    # its successful fits are as good as the fp64 oracle's to the accuracy fp32 data allow.
    taus = [0.5, 1.5, 3.0, 6.0, 12.0]
    B, m = 384, 4096
    d = synth.multi_exp_batch(B, 5, m, taus, noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
    mdl32 = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
    bp = vp.BatchProblem(mdl32, d["Y"], x=d["x"])
    alpha, C, rep = bp.fit(d["tau_guess"])
    bp.close()
    a32, c32, r32, _ = O.fit_batch_f32(mdl32, d["x"], d["Y"], d["tau_guess"], n_threads=8)
    x64, Y64, g64 = d["x"].astype(np.float64), d["Y"].astype(np.float64), d["tau_guess"].astype(np.float64)
    a64, c64, r64, _ = O.fit_batch(mdl32, x64, Y64, g64, n_threads=8)
    ok, ok32, ok64 = rep["termination"] > 0, r32["termination"] > 0, r64["termination"] > 0
    assert ok32.mean() < 0.97                                   # the fp32 reference algorithm does fail on this model
    assert (~ok).mean() <= (~ok32).mean() + 0.05                # the device is not worse than it
    assert (ok == ok32).mean() >= 0.8                           # and mostly fails / succeeds on the same problems
    assert ok.mean() >= 0.75
    scale = 0.5 * (Y64 ** 2).sum(1)
    both = ok & ok32
    # same minimum in fp32 terms: the objectives differ by no more than fp32 rounding of a residual of this size
    rel = np.abs(rep["objective"] - r32["objective"])[both] / np.maximum(r32["objective"][both], 1e-7 * scale[both])
    assert np.median(rel) <= 1e-2 and (rel <= 0.5).mean() >= 0.9
    # against fp64: a successful device fit is never (noticeably) worse than the fp64 oracle's minimum
    b64 = ok & ok64
    assert b64.mean() >= 0.5
    excess = (rep["objective"] - r64["objective"])[b64] / np.maximum(r64["objective"][b64], 1e-7 * scale[b64])
    assert np.median(excess) <= 1e-2 and (excess <= 0.5).mean() >= 0.9


def test_fp32_multiple_right_hand_sides():
    # global fit (one alpha, S right-hand sides) in fp32: evaluation vs the fp64 oracle to fp32 accuracy, and the
    # noise-free fit recovers the decay times to what fp32 resolves
    rng = np.random.default_rng(4)
    S, m = 64, 1000
    x = np.linspace(0.0, 12.5, m)
    Cm = rng.uniform(1, 100, (S, 3))
    Y = Cm[:, :1] * np.exp(-x / 1.0) + Cm[:, 1:2] * np.exp(-x / 3.0) + Cm[:, 2:]
    guess = np.array([1.3, 3.9])
    mdl32 = vp.multi_exponential_model(x.astype(np.float32), guess.astype(np.float32), dtype=np.float32)
    mdl64 = vp.multi_exponential_model(x, guess)
    bp = vp.BatchProblem(mdl32, Y[None].astype(np.float32), x=x.astype(np.float32))
    ev = bp.evaluate(guess[None].astype(np.float32))
    ref = O.Problem(mdl64, x, Y)
    ref.set_params(guess)
    assert ev["status"][0] == 0
    assert np.abs(ev["r"][0] - ref.residuals()).max() <= 2e-4 * np.abs(Y).max()
    assert np.abs(ev["C"][0] - ref.linear_coefficients()).max() <= 5e-3 * np.abs(ref.linear_coefficients()).max()
    alpha, C, rep = bp.fit(guess[None].astype(np.float32))
    assert rep["termination"][0] > 0 or rep["objective"][0] <= 1e-8 * 0.5 * (Y ** 2).sum()
    assert np.abs(np.sort(alpha[0]) - [1.0, 3.0]).max() <= 2e-2
    bp.close()
