"""The workgroup-cooperative MRHS kernels (fp64, 1024 < m <= 2048, m even: `mrhs_coop_out_kernel` for the trait-level
outputs r / J, `mrhs_coop_dma_kernel` for the fit pass) against the CPU oracle -- and the shapes just outside their
window (odd m, m <= 1024), which take the one-wave-per-column kernel: same numbers either way.
Reference: src/solvers/levmar/mod.rs:42-73 (set_params), :91-95 (residuals), :101-201 (Jacobian, MRHS branch :172-186)."""
import numpy as np
import pytest

import varpro_amd as vp
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _data(rng, S, m, tau, noise=1e-3):
    x = np.linspace(0.0, 12.5, m)
    Cm = rng.uniform(1, 50, (S, len(tau) + 1))
    Y = sum(Cm[:, j:j + 1] * np.exp(-x / tau[j]) for j in range(len(tau))) + Cm[:, -1:]
    return x, Y + noise * np.abs(Y).max() * rng.standard_normal(Y.shape)


@pytest.mark.parametrize("nexp", [2, 3])
# (the cooperative kernels serve the 32-rows-per-lane set: 1792 < m <= 2048 for two exponentials, 1536 < m <= 2048 for three,
# m even; the shorter and the odd lengths below run the one-wave-per-column kernel of the in-between sets)
@pytest.mark.parametrize("S,m,weighted", [(2, 2048, False), (3, 1026, True), (33, 1500, False), (64, 2048, True),
                                          (7, 1501, False), (129, 1030, False), (5, 2046, True), (3, 1794, True),
                                          (33, 1900, False), (7, 1801, False), (129, 1796, False)])
def test_trait_outputs_match_oracle(nexp, S, m, weighted):
    rng = np.random.default_rng(1000 * nexp + S + m)
    tau = [1.0, 3.0, 7.0][:nexp]
    guess = np.array([1.3, 3.6, 8.1][:nexp])
    x, Y = _data(rng, S, m, tau)
    w = rng.uniform(0.3, 2.0, m) if weighted else None
    mdl = vp.multi_exponential_model(x, guess, offset=True)
    bp = vp.BatchProblem(mdl, Y[None], x=x, weights=w)
    ev = bp.evaluate(guess[None])
    ref = O.Problem(mdl, x, Y, w=w)
    ref.set_params(guess)
    yw = Y if w is None else Y * w
    Cr = ref.linear_coefficients()
    assert ev["status"][0] == 0
    assert np.abs(ev["C"][0] - Cr).max() <= TOL * np.abs(Cr).max()
    assert np.abs(ev["r"][0] - ref.residuals()).max() <= TOL * np.abs(yw).max()
    Jr = ref.jacobian()
    for k in range(nexp):
        assert np.abs(ev["J"][0, k] - Jr[k]).max() <= 1e-10 * np.abs(Jr[k]).max()
    assert abs(ev["cost"][0] - 0.5 * (ref.residuals() ** 2).sum()) <= 1e-10 * ev["cost"][0]
    # residuals alone and the Jacobian alone (each output pointer may be absent)
    e2 = bp.evaluate(guess[None], want_jacobian=False)
    assert np.array_equal(e2["r"], ev["r"]) and np.array_equal(e2["C"], ev["C"])
    bp.close()


def test_trait_outputs_several_problems_each_with_its_own_parameters():
    rng = np.random.default_rng(5)
    B, S, m = 3, 19, 1900
    tau = [1.0, 3.0, 7.0]
    xs, Ys = zip(*[_data(rng, S, m, tau) for _ in range(B)])
    x = xs[0]
    Y = np.stack(Ys)
    guesses = np.array([[1.3, 3.6, 8.1], [0.9, 2.7, 6.0], [1.1, 3.3, 9.0]])
    mdl = vp.multi_exponential_model(x, guesses[0], offset=True)
    bp = vp.BatchProblem(mdl, Y, x=x)
    ev = bp.evaluate(guesses)
    for b in range(B):
        ref = O.Problem(mdl, x, Y[b])
        ref.set_params(guesses[b])
        assert np.abs(ev["C"][b] - ref.linear_coefficients()).max() <= TOL * np.abs(ref.linear_coefficients()).max()
        assert np.abs(ev["r"][b] - ref.residuals()).max() <= TOL * np.abs(Y[b]).max()
        Jr = ref.jacobian()
        for k in range(3):
            assert np.abs(ev["J"][b, k] - Jr[k]).max() <= 1e-10 * np.abs(Jr[k]).max()
    bp.close()


@pytest.mark.parametrize("S,m", [(9, 1500), (9, 1850), (40, 2048)])
def test_rank_deficient_basis_minimum_norm_solution_in_the_cooperative_kernel(S, m):
    # tau1 == tau2 (src/solvers/levmar/mod.rs:51-54: svd.solve(eps) -> minimum-norm coefficients for every column)
    rng = np.random.default_rng(S + m)
    x = np.linspace(0.0, 10.0, m)
    Y = rng.uniform(1, 5, (S, 1)) * np.exp(-x / 2.0) + rng.uniform(0, 1, (S, 1)) + 1e-3 * rng.standard_normal((S, m))
    mdl = vp.multi_exponential_model(x, [2.0, 2.0], offset=True)
    bp = vp.BatchProblem(mdl, Y[None], x=x, epsilon=1e-8)
    ev = bp.evaluate(np.array([[2.0, 2.0]]), want_jacobian=False)
    ref = O.Problem(mdl, x, Y, eps=1e-8)
    ref.set_params([2.0, 2.0])
    Cr = ref.linear_coefficients()
    assert ev["status"][0] == 0
    assert np.abs(ev["C"][0] - Cr).max() <= 1e-9 * np.abs(Cr).max()
    assert np.abs(ev["r"][0] - ref.residuals()).max() <= TOL * np.abs(Y).max()
    assert abs(ev["cost"][0] - 0.5 * (ref.residuals() ** 2).sum()) <= 1e-9 * ev["cost"][0]
    bp.close()


@pytest.mark.parametrize("S,m,weighted", [(24, 1500, False), (24, 1880, False), (70, 2048, True)])
def test_global_fit_trajectory_matches_oracle(S, m, weighted):
    rng = np.random.default_rng(S * m)
    tau = [1.0, 3.0, 7.0]
    x, Y = _data(rng, S, m, tau)
    w = rng.uniform(0.5, 1.5, m) if weighted else None
    guess = np.array([[1.2, 3.5, 8.0]])
    mdl = vp.multi_exponential_model(x, guess[0], offset=True)
    bp = vp.BatchProblem(mdl, Y[None], x=x, weights=w)
    alpha, C, rep, tr = bp.fit_trace(guess, max_rows=12)
    ref = O.Problem(mdl, x, Y, w=w)
    ref.set_params(guess[0])
    rr, tr_ref = ref.fit_trace(max_rows=12)
    assert rep["termination"][0] > 0 and rr.termination > 0
    for i in range(min(5, len(tr_ref), int(rep["n_evals"][0]))):
        assert np.abs(tr[0, i, :3] - tr_ref[i, :3]).max() <= 1e-6 * np.abs(tr_ref[i, :3]).max(), i
    assert abs(rep["objective"][0] - rr.objective) <= 1e-8 * rr.objective
    assert np.abs(alpha[0] - ref.params()).max() <= 1e-6 * np.abs(ref.params()).max()
    r = bp.residuals()
    assert abs(0.5 * (r ** 2).sum() - rep["objective"][0]) <= 1e-9 * rep["objective"][0]
    bp.close()


def test_repeated_fits_on_one_handle_follow_the_graph_length_adaptation():
    # the whole-fit graph is sized from the handle's PREVIOUS fit (evaluations + 1) and continued by a 12-iteration tail
    # graph when a fit outlasts it: a sequence of fits of different lengths on ONE handle must give, fit by fit, exactly
    # what a fresh handle gives (same kernels, same order of operations: bit-identical reports and parameters)
    rng = np.random.default_rng(11)
    m, S = 1900, 12
    tau = [1.0, 3.0, 7.0]
    x, Y_easy = _data(rng, S, m, tau, noise=1e-4)
    _, Y_hard = _data(rng, S, m, [0.8, 1.1, 9.0], noise=5e-2)
    guess_easy, guess_hard = np.array([[1.1, 3.2, 7.5]]), np.array([[0.3, 4.0, 20.0]])
    mdl = vp.multi_exponential_model(x, guess_easy[0], offset=True)

    def fresh(Y, g):
        bp = vp.BatchProblem(mdl, Y[None], x=x)
        out = bp.fit(g)
        bp.close()
        return out

    ref_easy, ref_hard = fresh(Y_easy, guess_easy), fresh(Y_hard, guess_hard)
    assert ref_hard[2]["n_evals"][0] > ref_easy[2]["n_evals"][0] + 3   # the sequence really changes length
    bp = vp.BatchProblem(mdl, Y_easy[None], x=x)
    for Y, g, ref in ((Y_easy, guess_easy, ref_easy), (Y_hard, guess_hard, ref_hard), (Y_easy, guess_easy, ref_easy),
                      (Y_easy, guess_easy, ref_easy), (Y_hard, guess_hard, ref_hard)):
        bp.set_observations(Y[None])
        a, C, rep = bp.fit(g)
        assert rep["n_evals"][0] == ref[2]["n_evals"][0] and rep["termination"][0] == ref[2]["termination"][0]
        assert np.array_equal(a, ref[0]) and np.array_equal(C, ref[1])
        assert rep["objective"][0] == ref[2]["objective"][0]
    # changed LM options re-capture the graph: tighter patience ends the hard fit early with LostPatience
    slv = vp.LevenbergMarquardt().with_patience(2)
    bp.set_observations(Y_hard[None])
    a, C, rep = bp.fit(guess_hard, solver=slv)
    assert rep["n_evals"][0] <= 2 * 4 + 1
    bp.close()


@pytest.mark.parametrize("m,S,B", [(1500, 40, 1), (600, 9, 3), (1900, 40, 1), (2048, 64, 2)])
def test_device_pointer_global_fit_equals_the_host_pointer_fit(m, S, B):
    # device-pointer handles: the whole-fit graph reads alpha_0 from, and writes alpha / C / reports into, the caller's device
    # arrays through the pinned MrhsIo record; host-pointer handles stage through the library's buffers.  Same kernels, same
    # numbers -- bit for bit -- and the handle's own state (params, coefficients) agrees with what was returned.
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(m + S)
    tau = [1.0, 3.0, 7.0]
    xs, Ys = zip(*[_data(rng, S, m, tau) for _ in range(B)])
    x, Y = xs[0], np.stack(Ys)
    guess = np.tile(np.array([[1.2, 3.4, 7.9]]), (B, 1)) * rng.uniform(0.95, 1.05, (B, 3))
    mdl = vp.multi_exponential_model(x, guess[0], offset=True)
    bh = vp.BatchProblem(mdl, Y, x=x)
    ah, Ch, reph = bh.fit(guess)
    bh.close()
    bd = vp.BatchProblem(mdl, torch.from_numpy(Y).to(dev), x=torch.from_numpy(x).to(dev))
    g = torch.from_numpy(guess).to(dev)
    for _ in range(2):   # (the second fit replays the graph sized from the first)
        ad, Cd, repd = bd.fit(g)
        repd = bd.report_to_numpy(repd)
        assert np.array_equal(ad.cpu().numpy(), ah) and np.array_equal(Cd.cpu().numpy(), Ch)
        assert np.array_equal(repd["n_evals"], reph["n_evals"]) and np.array_equal(repd["objective"], reph["objective"])
        assert np.array_equal(g.cpu().numpy(), guess)                 # the caller's initial parameters are left alone
    assert np.array_equal(bd.params().cpu().numpy(), ah)
    assert np.array_equal(bd.linear_coefficients().cpu().numpy(), Ch)
    a2, C2, rep2 = bd.fit(g, want_coefficients=False)                 # no coefficient array from the caller
    assert C2 is None and np.array_equal(a2.cpu().numpy(), ah)
    bd.close()


def test_device_pointer_fits_that_outlast_the_graph():
    # a fit longer than the captured graph continues with the tail graph; on a device-pointer handle every replay writes the
    # caller's arrays through the MrhsIo record -- easy / hard / easy data on one handle against host-pointer handles
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(21)
    m, S = 1840, 10
    x, Y_easy = _data(rng, S, m, [1.0, 3.0, 7.0], noise=1e-4)
    _, Y_hard = _data(rng, S, m, [0.8, 1.1, 9.0], noise=5e-2)
    g_easy, g_hard = np.array([[1.1, 3.2, 7.5]]), np.array([[0.3, 4.0, 20.0]])
    mdl = vp.multi_exponential_model(x, g_easy[0], offset=True)

    def host(Y, g):
        bp = vp.BatchProblem(mdl, Y[None], x=x)
        out = bp.fit(g)
        bp.close()
        return out

    refs = {"easy": host(Y_easy, g_easy), "hard": host(Y_hard, g_hard)}
    assert refs["hard"][2]["n_evals"][0] > 13          # longer than the first graph (12 iterations + the initial evaluation)
    bd = vp.BatchProblem(mdl, torch.from_numpy(Y_easy[None]).to(dev), x=torch.from_numpy(x).to(dev))
    for kind in ("hard", "easy", "hard", "easy"):
        Y, g = (Y_hard, g_hard) if kind == "hard" else (Y_easy, g_easy)
        bd.set_observations(torch.from_numpy(Y[None]).to(dev))
        a, C, rep = bd.fit(torch.from_numpy(g).to(dev))
        rep = bd.report_to_numpy(rep)
        ra, rC, rrep = refs[kind]
        assert np.array_equal(a.cpu().numpy(), ra) and np.array_equal(C.cpu().numpy(), rC)
        assert rep["n_evals"][0] == rrep["n_evals"][0] and rep["termination"][0] == rrep["termination"][0]
    bd.close()
