"""Parity census at BATCH scale ("next" item 2): the oracle fits the WHOLE batch the device fits -- every
one of BASELINE configs[1]'s 4 096 problems, every one of the 65 536 problems of a configs[3] per-GPU shard (the headline
workload of bench.py), and a 2 048-problem sample of configs[4] -- and the two are compared problem by problem: what
`LevMarSolver::fit` returns for each (/root/reference/src/solvers/levmar/mod.rs:238-254: Ok / Err by
`termination.was_successful()`; /root/reference/src/fit.rs:113-122).  The slot kernel's refill / queue path, the lone tail
and the long fits (50-114 evaluations) are all inside these batches.

Contract asserted for the fp64 double exponential: the SAME success class for every problem (a disagreement would be
listed with both termination codes, evaluation counts and objectives), the same failures by termination code, objective of
the common successes to 1e-12 (median) / 1e-6 (max), |delta n_evals| <= 3 on >= 95 %, the largest evaluation count within 5 %.
For fp32 configs[4] the oracle runs in fp64 on the converted inputs with the handle's fp32 tolerances (30 eps_32): five
exponentials are conditioned 1e6+, a trial point that steps a decay time through zero ends the fit as `User` (non-finite
evaluation) on whichever side takes that step -- the census states the rates, and asserts that EVERY disagreement is of that
kind."""
import json

import numpy as np
import pytest

import contracts as K
import varpro_amd as vp
from oracle import census as CS
from oracle import oracle as O
from varpro_amd import synth

pytestmark = pytest.mark.gpu


def _double_exp_census(B, first=0, m=1024):
    d = synth.double_exp_batch(B, m=m, first_problem=first, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    a, _c, rep = bp.fit(d["tau_guess"])
    bp.close()
    ao, _co, ro, _s = O.fit_batch(mdl, d["x"], d["Y"], d["tau_guess"], n_threads=min(16, O.max_threads()))
    res = CS.census(rep, a, ro, ao, max_listed=200)
    print(json.dumps({k: v for k, v in res.items() if k != "disagreements"}))
    for dis in res["disagreements"]:
        print("DISAGREEMENT", dis)
    return res


# the largest evaluation count of a batch belongs to ONE fit that creeps along a flat valley for > 100 evaluations; its count
# is 118 on the device against 114 in the oracle (round 5: the two-parameter lmpar, vp_fit.hpp lmpar_q2, solves the same
# trust-region problem with a different rounding pattern than MINPACK's Givens sweep; rounds 2-4 happened to land on 114 =
# 114) -- 5 % on that one number, every other clause unchanged
def _assert_fp64_contract(res, max_evals_slack=K.FP64["max_evals_slack"], sum_evals_slack=K.FP64["sum_evals_slack"]):
    # (the numbers live in tests/contracts.py; tests/test_contracts_frozen.py fails when one is loosened)
    assert res["success_class_disagreements"] == K.FP64["success_class_disagreements"], res["disagreements"]
    assert res["failures_by_code_device"] == res["failures_by_code_oracle"]
    assert res["failed_on_both"] == res["failed_device"] == res["failed_oracle"]
    assert res["objective_rel_diff_median_common_successes"] <= K.FP64["objective_rel_median_max"]
    assert res["objective_rel_diff_max_common_successes"] <= K.FP64["objective_rel_max_max"]
    assert res["share_evals_within_3"] >= K.FP64["share_evals_within_3_min"]
    if max_evals_slack is not None:
        assert abs(res["max_evals_device"] - res["max_evals_oracle"]) <= max_evals_slack * res["max_evals_oracle"]
    assert abs(res["sum_evals_device"] - res["sum_evals_oracle"]) <= sum_evals_slack * res["sum_evals_oracle"]


def test_census_configs1_all_4096_problems():
    _assert_fp64_contract(_double_exp_census(4096))


def test_census_streamed_bench_leg_m10000_all_16384_problems():
    # bench.py's `streamed` leg at full size (m = 10 000: beyond every register-resident set, the length-agnostic
    # blk_fit_kernel of vp_block.hpp): the oracle fits all 16 384 problems (~10 s on 16 threads)
    # (the longest fit creeps along a flat valley for > 100 evaluations; its count is 120 here and 126 in the oracle -- the
    # TSQR carry and the oracle's Householder sweep round differently -- hence the 10 % on the largest count; every other
    # clause is the contract of the resident kernels; with blocks of 1 024 rows -- block_rows_long -- the device stops a
    # little EARLIER than the oracle on the whole: 143 164 evaluations against 146 161, 2.05 % fewer (1.8 % with 512-row
    # blocks), within 3 of the oracle's count on 98.2 % of the fits: the sum is held to 3 % on this leg)
    _assert_fp64_contract(_double_exp_census(16384, m=10000), max_evals_slack=K.STREAMED_M10000["max_evals_slack"],
                          sum_evals_slack=K.STREAMED_M10000["sum_evals_slack"])


def test_census_generic_fallback_bench_leg_oleary_m5000_all_4096_problems():
    # bench.py's `generic_fallback` leg at full size: the O'Leary exp*cos model (shared_test_code/src/models.rs:397-425;
    # n = 2, q = 3, four dependency pairs, a shared parameter) at m = 5000 -- the run-time-descriptor instance of the
    # length-agnostic kernel (blk_fit_kernel<RtModel<2,3,4>>); same workload generator as bench.py
    Bg, mg = 4096, 5000
    tg = np.linspace(0.0, 1.5, mg)
    rg = synth.SplitMix64(np.uint64(0x5EED3000) + np.arange(Bg, dtype=np.uint64))
    at = np.stack([1.0 * (1 + 0.1 * rg.uniform(-1, 1)), 2.5 * (1 + 0.1 * rg.uniform(-1, 1)), 4.0 * (1 + 0.1 * rg.uniform(-1, 1))], 1)
    cg = np.stack([rg.uniform(4.0, 8.0), rg.uniform(0.5, 2.0)], 1)
    Yg = (cg[:, :1] * np.exp(-at[:, 1:2] * tg[None]) * np.cos(at[:, 2:3] * tg[None])
          + cg[:, 1:2] * np.exp(-at[:, 0:1] * tg[None]) * np.cos(at[:, 1:2] * tg[None]))
    Yg = Yg + 1e-3 * np.abs(Yg).max(1, keepdims=True) * rg.normal(mg)
    gg0 = at * np.stack([1 + 0.1 * rg.uniform(-1, 1) for _ in range(3)], 1)
    mdl = (vp.SeparableModelBuilder(["alpha1", "alpha2", "alpha3"]).initial_parameters(gg0[0]).independent_variable(tg)
           .function(["alpha2", "alpha3"], vp.basis.EXP_COS).partial_deriv("alpha2").partial_deriv("alpha3")
           .function(["alpha1", "alpha2"], vp.basis.EXP_COS).partial_deriv("alpha1").partial_deriv("alpha2").build())
    bp = vp.BatchProblem(mdl, Yg, x=tg)
    a, _c, rep = bp.fit(gg0)
    ao, _co, ro, _s = O.fit_batch(mdl, tg, Yg, gg0, n_threads=min(16, O.max_threads()))
    res = CS.census(rep, a, ro, ao, max_listed=50)
    print(json.dumps({k: v for k, v in res.items() if k != "disagreements"}))
    for dis in res["disagreements"]:
        print("DISAGREEMENT", dis)
    # the largest evaluation count is not compared on this leg: with 4 096 problems it is set by one or two fits that creep
    # along the flat cos-frequency valley (the three fitted parameters within 0.5 % of each other: the two basis functions
    # all but coincide) until a tolerance fires; the counts agree within 3 on 98.5 % and in their sum to 1 %.
    # WHERE such a fit stops is not a property of the problem: the ORACLE's own result moves by 1e-4 ... 0.3 in the objective
    # when every datum is moved by at most one ulp (problems 818, 1104, 898 of this batch: 10 %, 64 %, 48 % of 256 such
    # perturbations end more than 1e-6 away from the unperturbed run -- tools/valley_probe.py, profiles/r05_valley_probe.json).
    # Contract: 1e-6 on every problem the oracle itself reproduces to 1e-6; a problem beyond it (at most 0.1 % of the batch)
    # must (i) be one the oracle does NOT reproduce under one-ulp perturbations of its data and (ii) sit within 1e-6 of one
    # of the oracle's own outcomes.
    rd = bp.report_to_numpy(rep)
    bp.close()
    od, oo = rd["objective"].astype(np.float64), ro["objective"].astype(np.float64)
    rel = np.abs(od - oo) / oo
    beyond = [int(i) for i in np.nonzero(rel > 1e-6)[0]]
    print("beyond 1e-6:", [(i, float(rel[i])) for i in beyond])
    assert len(beyond) <= K.OLEARY_M5000["beyond_1e-6_max_problems"]
    for i in beyond:
        n = 256
        Yp = np.repeat(Yg[i:i + 1], n, 0) * (1 + 2.0 ** -52 * np.random.default_rng(i).choice([-1.0, 0.0, 1.0], (n, mg)))
        _a, _c2, rp, _s2 = O.fit_batch(mdl, tg, Yp, np.repeat(gg0[i:i + 1], n, 0), n_threads=min(16, O.max_threads()))
        outcomes = rp["objective"].astype(np.float64)
        print(i, "oracle outcomes beyond 1e-6 of its unperturbed run:", float((np.abs(outcomes - oo[i]) / oo[i] > 1e-6).mean()),
              "device to the nearest outcome:", float(np.min(np.abs(outcomes - od[i])) / oo[i]))
        assert (np.abs(outcomes - oo[i]) / oo[i] > 1e-6).any(), "a well-posed problem beyond 1e-6"
        assert np.min(np.abs(outcomes - od[i])) / oo[i] <= 1e-6
    res_wp = dict(res)
    if beyond:
        keep = np.ones(Bg, bool)
        keep[beyond] = False
        res_wp["objective_rel_diff_max_common_successes"] = float(rel[keep & (rd["termination"] > 0) & (ro["termination"] > 0)].max())
    _assert_fp64_contract(res_wp, max_evals_slack=None)


def test_census_configs3_shard_all_65536_problems():
    res = _double_exp_census(65536)
    _assert_fp64_contract(res)
    # the failures of the headline set (bench.py reports them as fits_failed): non-finite evaluations on BOTH sides
    assert set(res["failures_by_code_device"]) <= {"User", "LostPatience", "NoImprovementPossible", "Numerical"}


def test_census_second_shard_of_configs3():
    # shard 7 of the 8-GPU problem set (problems 458752 ..): the one that holds a 300-evaluation LostPatience fit
    res = _double_exp_census(65536, first=7 * 65536)
    _assert_fp64_contract(res)


def test_census_configs4_sample_of_2048():
    B = 2048
    d = synth.multi_exp_batch(B, 5, 4096, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    a, _c, rep = bp.fit(d["tau_guess"])
    bp.close()
    x64, Y64, g64 = d["x"].astype(np.float64), d["Y"].astype(np.float64), d["tau_guess"].astype(np.float64)
    mdl64 = vp.multi_exponential_model(x64, g64[0])
    e32 = float(np.finfo(np.float32).eps)
    ao, _co, ro, _s = O.fit_batch(mdl64, x64, Y64, g64, n_threads=min(16, O.max_threads()),
                                  opts=O.default_opts(ftol=30 * e32, xtol=30 * e32, gtol=30 * e32))
    res = CS.census(rep, a, ro, ao, max_listed=B)
    print(json.dumps({k: v for k, v in res.items() if k != "disagreements"}))
    # every disagreement: one side ended `User` (a trial point with a non-positive / overflowing decay time), the other converged
    for dis in res["disagreements"]:
        assert "User" in (dis["device"], dis["oracle"]), dis
    assert res["same_success_class"] >= K.CFG4_SAMPLE["same_success_class_min"]
    assert set(res["failures_by_code_device"]) <= {"User"} and set(res["failures_by_code_oracle"]) <= {"User", "LostPatience"}
    assert res["failed_device"] <= K.CFG4_SAMPLE["failed_device_max_share"] * B and res["failed_oracle"] <= K.CFG4_SAMPLE["failed_oracle_max_share"] * B
    # the common successes sit in the same valley of a very flat objective (cond(Phi) >= 1e6: parameters are NOT comparable)
    assert res["objective_rel_diff_median_common_successes"] <= 1e-4
    assert res["share_objective_within_1e-3"] >= 0.9
    assert abs(res["sum_evals_device"] - res["sum_evals_oracle"]) <= 0.1 * res["sum_evals_oracle"]


def test_census_configs4_all_8192_problems():
    """BASELINE configs[4] at its FULL batch (round 4 censused a sample of 2 048): the fp64 oracle at the handle's fp32 tolerances
    fits every problem; plus the contract on what a successful fit REPORTS -- its objective against the fp64 oracle's cost at
    the parameters it RETURNS (the Gram formulation squares cond(Phi) >= 1e6: stated bound kappa^2 x 1e-13).  Thresholds =
    measured (round 5: same class 96.1 %, 2.5 % failed, 0.2 % of the reported objectives off by more than 1e-3) minus a margin."""
    B = 8192
    d = synth.multi_exp_batch(B, 5, 4096, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    a, _c, rep = bp.fit(d["tau_guess"])
    bp.close()
    x64, Y64, g64 = d["x"].astype(np.float64), d["Y"].astype(np.float64), d["tau_guess"].astype(np.float64)
    mdl64 = vp.multi_exponential_model(x64, g64[0])
    e32 = float(np.finfo(np.float32).eps)
    nt = min(16, O.max_threads())
    ao, _co, ro, _s = O.fit_batch(mdl64, x64, Y64, g64, n_threads=nt, opts=O.default_opts(ftol=30 * e32, xtol=30 * e32, gtol=30 * e32))
    res = CS.census(rep, a, ro, ao, max_listed=B)
    print(json.dumps({k: v for k, v in res.items() if k != "disagreements"}))
    # (one fit of the 8 192 ends `Numerical` on the device: a non-finite trust-region step out of a Gram factor at the edge of
    # its resolution -- the same class of trial point as `User`, caught one statement later)
    for dis in res["disagreements"]:
        assert "User" in (dis["device"], dis["oracle"]) or dis["device"] == "Numerical", dis
    C4 = K.CFG4_ALL
    assert res["same_success_class"] >= C4["same_success_class_min"]
    assert set(res["failures_by_code_device"]) <= {"User", "Numerical"} and res["failures_by_code_device"].get("Numerical", 0) <= C4["numerical_failures_max"]
    assert set(res["failures_by_code_oracle"]) <= {"User", "LostPatience"}
    assert res["failed_device"] <= C4["failed_device_max_share"] * B and res["failed_oracle"] <= C4["failed_oracle_max_share"] * B
    assert res["objective_rel_diff_median_common_successes"] <= C4["objective_rel_median_max"]
    assert res["share_objective_within_1e-3"] >= C4["share_objective_within_1e-3_min"]
    assert abs(res["sum_evals_device"] - res["sum_evals_oracle"]) <= C4["sum_evals_slack"] * res["sum_evals_oracle"]
    # reported objective vs the true cost at the returned point (fp64 thin-SVD solve on the lattice the kernel takes the uniform grid as)
    ok = rep["termination"] > 0
    t0 = float(d["x"][0])
    grid = t0 + np.arange(d["x"].shape[-1]) * ((float(d["x"][-1]) - t0) / (d["x"].shape[-1] - 1))
    ref = O.evaluate_batch(vp.multi_exponential_model(grid, a[0].astype(np.float64)), grid, Y64[ok], a[ok].astype(np.float64),
                           n_threads=nt, want_jac=False)
    rel = np.abs(rep["objective"][ok] - ref["cost"]) / ref["cost"]
    print(json.dumps({"successes": int(ok.sum()), "median": float(np.median(rel)), "p99": float(np.quantile(rel, 0.99)),
                      "max": float(rel.max()), "share_above_1e-3": float((rel > 1e-3).mean()), "share_above_1e-2": float((rel > 1e-2).mean())}))
    assert (rel > 1e-3).mean() <= C4["reported_objective_share_above_1e-3_max"]
    assert (rel > 1e-2).mean() <= C4["reported_objective_share_above_1e-2_max"] and np.median(rel) <= C4["reported_objective_median_max"]


# ---- the one disagreement of the round-4 census of configs[3] (shard 3, problem 29 433) ------------------------------
@pytest.mark.parametrize("kernel", ["wave", "slots"])
def test_column_near_overflow_is_an_ordinary_trial_point(kernel):
    """The third evaluation of this fit is a trial decay time tau_2 = -0.0353: exp(+t/0.0353) reaches 1e153, its coefficient is
    1e-152 and its derivative column 1e157.  The reference forms D_k c (1e5) before it projects (src/solvers/levmar/mod.rs:
    156-171) and goes on to a minimum; the device kernels carry the unscaled derivative column through the sweep, where the
    dot product with the 1e153 column is not representable -- rescue_jacobian (vp_fit.hpp) repeats such an evaluation with
    the columns scaled by powers of two.  Same success class and minimum as the oracle (round 4: `Numerical`)."""
    first = 3 * 65536 + 29433
    d = synth.double_exp_batch(4, m=1024, first_problem=first, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    bp.set_fit_kernel(kernel)
    a, c, rep, tr = bp.fit_trace(d["tau_guess"], max_rows=8)
    bp.close()
    p = O.Problem(mdl, d["x"], d["Y"][0])
    p.set_params(d["tau_guess"][0])
    r, tro = p.fit_trace(max_rows=8)
    # the trial point that used to end the fit is the third evaluation of both drivers
    assert tr[0][2][1] < 0 and abs(tr[0][2][1] - tro[2][1]) <= 1e-9 * abs(tro[2][1])
    assert (int(rep["termination"][0]) > 0) == (int(r.termination) > 0), (rep[0], r.termination)
    assert int(rep["termination"][0]) != -2  # VP_TERM_NUMERICAL: how round 4 ended this fit
    assert abs(rep["objective"][0] - r.objective) <= 1e-6 * r.objective, (rep["objective"][0], r.objective)
    assert abs(int(rep["n_evals"][0]) - int(r.n_evals)) <= 8, (rep["n_evals"][0], r.n_evals)


# ---- flag-and-refit (round 6): the same event in EVERY kernel family ----------------------------------------------------
# A start inside the window the register kernels cannot represent column by column: tau_2 = -0.0356 on t in [0, 12.5] makes
# exp(+t/0.0356) reach 3e152 with a coefficient of ~1e-151 and a derivative column of ~1e157.  The evaluation is fine, the
# Jacobian of the unscaled derivative columns is not; the reference forms D_k c first (src/solvers/levmar/mod.rs:156-171).
# Round 5 repaired this inside the full-length unweighted static kernels only; now every family flags the problem and
# vp_fit's second launch re-fits it with power-of-two column scaling (vp_fit.hpp jac_not_finite, gen::evaluate).
def _window_batch(m, nprob=6):
    d = synth.double_exp_batch(nprob, m=m, first_problem=4242, noise=1e-3)
    g = d["tau_guess"].copy()
    g[1, 1] = -0.0356   # problem 1 STARTS in the window: flagged at its first Jacobian
    g[4, 0] = -0.03555  # and problem 4 with the other parameter
    return d, g


def _oracle_fit(mdl, x, y, guess, w=None):
    p = O.Problem(mdl, x, y, w=w) if w is not None else O.Problem(mdl, x, y)
    p.set_params(guess)
    r = p.fit()
    return r, p.params()


@pytest.mark.parametrize("m,kernel,weighted", [
    (1024, "wave", False), (1024, "slots", False),   # full length: fit_kernel / fit2_kernel<PADM 1> (round 5's only coverage)
    (1024, "wave", True),                            # weighted kernels
    (1000, "wave", False), (1000, "slots", False),   # nearly full (PADM 2)
    (700, "wave", False), (700, "slots", False),     # general length (PADM 0)
    (3000, "auto", False),                           # four waves per problem
    (10000, "auto", False), (10000, "auto", True),   # rows streamed in blocks (blk_fit_kernel)
])
def test_refit_of_unrepresentable_jacobians_in_every_kernel_family(m, kernel, weighted):
    d, g = _window_batch(m)
    mdl = vp.multi_exponential_model(d["x"], g[0])
    w = np.ones(m) if weighted else None
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"], weights=w)
    if kernel != "auto":
        bp.set_fit_kernel(kernel)
    # without the second launch: what the kernels report themselves
    bp.set_refit(False)
    a0, _c0, rep0 = bp.fit(g)
    rep0 = bp.report_to_numpy(rep0)
    for b in (1, 4):
        assert int(rep0["termination"][b]) == -2 and int(rep0["n_evals"][b]) == 1, (b, rep0[b])
    bp.set_refit(True)
    a1, c1, rep1 = bp.fit(g)
    rep1 = bp.report_to_numpy(rep1)
    # a second fit on the same handle (the list's ping-pong counters): identical
    a2, c2, rep2 = bp.fit(g)
    rep2 = bp.report_to_numpy(rep2)
    assert np.array_equal(np.asarray(a1), np.asarray(a2)) and np.array_equal(rep1["n_evals"], rep2["n_evals"])
    bp.close()
    for b in range(g.shape[0]):
        r, a_ref = _oracle_fit(mdl, d["x"], d["Y"][b], g[b], w)
        if b not in (1, 4):
            # every problem that is not flagged is bit for bit what the fit kernels returned
            assert np.array_equal(np.asarray(a1)[b], np.asarray(a0)[b]) and rep1[b] == rep0[b]
        assert (int(rep1["termination"][b]) > 0) == (int(r.termination) > 0), (b, rep1[b], r.termination)
        if int(r.termination) > 0:
            assert abs(rep1["objective"][b] - r.objective) <= K.REFIT["objective_rel_max"] * r.objective, (b, rep1["objective"][b], r.objective)
            # (a start this wild takes 30-80 evaluations through a region where the basis is conditioned 1e150: the count is
            # held loosely, the minimum tightly)
            assert abs(int(rep1["n_evals"][b]) - int(r.n_evals)) <= max(K.REFIT["evals_abs_slack"], K.REFIT["evals_rel_slack"] * int(r.n_evals)), (b, rep1["n_evals"][b], r.n_evals)
        else:
            assert int(rep1["termination"][b]) == int(r.termination), (b, rep1[b], r.termination)


def test_refit_with_run_time_descriptor_model():
    """the same start for a model of the descriptor language's run-time family (exp + exp, no offset)"""
    m = 512
    d, g = _window_batch(m)
    mdl = (vp.SeparableModelBuilder(["t1", "t2"]).initial_parameters(g[0]).independent_variable(d["x"])
           .function(["t1"], vp.basis.EXP_DECAY).partial_deriv("t1")
           .function(["t2"], vp.basis.EXP_DECAY).partial_deriv("t2").build())
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    bp.set_refit(False)
    _a0, _c0, rep0 = bp.fit(g)
    rep0 = bp.report_to_numpy(rep0)
    bp.set_refit(True)
    a1, _c1, rep1 = bp.fit(g)
    rep1 = bp.report_to_numpy(rep1)
    bp.close()
    assert int(rep0["termination"][1]) == -2
    for b in (1, 4):
        r, _a = _oracle_fit(mdl, d["x"], d["Y"][b], g[b])
        assert (int(rep1["termination"][b]) > 0) == (int(r.termination) > 0), (b, rep1[b], r.termination)
        if int(r.termination) > 0:
            assert abs(rep1["objective"][b] - r.objective) <= 1e-6 * r.objective
