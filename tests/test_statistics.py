"""Fit statistics: covariance, reduced chi^2, confidence band.
CPU tier pins the oracle on the reference's fixtures; GPU tier checks vp_statistics against both."""
import numpy as np
import pytest
from scipy import stats as sst

import refdata as rd
import varpro_amd as vp
from models import double_exp_builder_model, oleary_model
from oracle import oracle as O

# tests/integration_tests/main.rs:780-823 (O'Leary / MATLAB varpro output, epsilon 1e-5)
OLEARY_COV = np.array([
    [4.4887e-03, -4.4309e-03, -2.1613e-04, -4.6980e-04, -1.9052e-03],
    [-4.4309e-03, 4.3803e-03, 2.1087e-04, 4.7170e-04, 1.8828e-03],
    [-2.1613e-04, 2.1087e-04, 2.6925e-04, -3.6450e-05, 5.1919e-05],
    [-4.6980e-04, 4.7170e-04, -3.6450e-05, 8.5784e-05, 2.0534e-04],
    [-1.9052e-03, 1.8828e-03, 5.1919e-05, 2.0534e-04, 8.2272e-04]])
OLEARY_CORR = np.array([
    [1.0000, -0.9993, -0.1966, -0.7571, -0.9914],
    [-0.9993, 1.0000, 0.1942, 0.7695, 0.9918],
    [-0.1966, 0.1942, 1.0000, -0.2398, 0.1103],
    [-0.7571, 0.7695, -0.2398, 1.0000, 0.7729],
    [-0.9914, 0.9918, 0.1103, 0.7729, 1.0000]])
LMFIT_CHI2 = {False: 1.0109e-4, True: 3.2117e-5}


def _lmfit_case(weighted):
    pre = "weighted_multiexp" if weighted else "multiexp"
    x = rd.read_raw_f64(pre + "_xdata_1000_64bit.raw")
    y = rd.read_raw_f64(pre + "_ydata_1000_64bit.raw")
    cov = rd.read_raw_f64(pre + "_covmat_5x5_64bit.raw").reshape(5, 5)
    conf = rd.read_raw_f64(pre + "_conf_1000_64bit.raw")
    w = 1.0 / np.sqrt(y) if weighted else None
    return x, y, w, cov, conf


@pytest.mark.parametrize("weighted", [False, True])
def test_oracle_statistics_match_lmfit_fixtures(weighted):
    # tests/integration_tests/main.rs:600-612, 670-687: covariance 1e-6, confidence band (p = 0.88) 1e-6, chi2 1e-8
    x, y, w, cov, conf = _lmfit_case(weighted)
    p = O.Problem(double_exp_builder_model(x, [1., 7.]), x, y, w=w)
    p.set_params([1., 7.])
    assert p.fit().termination > 0
    st = p.statistics()
    assert abs(st["reduced_chi2"] - LMFIT_CHI2[weighted]) < 1e-8
    assert np.abs(st["cov"] - cov).max() < 1e-6
    assert np.abs(sst.t.ppf((0.88 + 1) / 2, st["dof"]) * st["conf_sigma"] - conf).max() < 1e-6


def test_oracle_statistics_match_oleary_matlab_output():
    p = O.Problem(oleary_model(rd.OLEARY_T, rd.OLEARY_GUESS), rd.OLEARY_T, rd.OLEARY_Y, w=rd.OLEARY_W)
    p.set_params(rd.OLEARY_GUESS)
    assert p.fit().termination > 0
    st = p.statistics()
    assert abs(np.sqrt(st["reduced_chi2"]) - 2.7539e-03) < 1e-5
    assert np.abs(st["cov"] - OLEARY_COV).max() < 1e-5
    d = np.sqrt(np.diag(st["cov"]))
    assert np.abs(st["cov"] / np.outer(d, d) - OLEARY_CORR).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("weighted", [False, True])
def test_gpu_fit_with_statistics_matches_lmfit_fixtures(weighted):
    x, y, w, cov, conf = _lmfit_case(weighted)
    mdl = double_exp_builder_model(x, [1., 7.])
    b = vp.SeparableProblemBuilder(mdl).observations(y)
    prob = (b.weights(w) if weighted else b).build()
    res, st = vp.LevMarSolver.default().fit_with_statistics(prob)
    assert res.was_successful()
    assert abs(st.reduced_chi2() - LMFIT_CHI2[weighted]) < 1e-8
    assert abs(st.regression_standard_error() - np.sqrt(st.reduced_chi2())) < 1e-15
    assert np.abs(st.covariance_matrix() - cov).max() < 1e-6
    assert np.abs(st.confidence_band_radius(0.88) - conf).max() < 1e-6
    assert np.abs(st.weighted_residuals() - res.problem.residuals()).max() == 0
    assert st.linear_coefficients_variance().shape == (3,) and st.nonlinear_parameters_variance().shape == (2,)


@pytest.mark.gpu
def test_gpu_statistics_oleary_and_batch_vs_oracle():
    prob = (vp.SeparableProblemBuilder(oleary_model(rd.OLEARY_T, rd.OLEARY_GUESS)).observations(rd.OLEARY_Y)
            .weights(rd.OLEARY_W).build())
    res, st = vp.LevMarSolver.default().fit_with_statistics(prob)
    assert abs(st.regression_standard_error() - 2.7539e-03) < 1e-5
    assert np.abs(st.covariance_matrix() - OLEARY_COV).max() < 1e-5
    assert np.abs(st.calculate_correlation_matrix() - OLEARY_CORR).max() < 1e-4
    assert np.abs(st.nonlinear_parameters_variance() - [2.6925e-04, 8.5784e-05, 8.2272e-04]).max() < 1e-5
    assert np.abs(st.linear_coefficients_variance() - [4.4887e-03, 4.3803e-03]).max() < 1e-5
    # a batch against the oracle at 1e-9
    from varpro_amd import synth
    d = synth.double_exp_batch(12, m=1024, noise=1e-3)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    alpha, C, rep = bp.fit(d["tau_guess"])
    s = bp.statistics()
    for b in range(12):
        p = O.Problem(mdl, d["x"], d["Y"][b])
        p.set_params(alpha[b])
        ref = p.statistics()
        assert abs(s["reduced_chi2"][b] - ref["reduced_chi2"]) <= 1e-10 * ref["reduced_chi2"]
        assert np.abs(s["cov"][b] - ref["cov"]).max() <= 1e-8 * np.abs(ref["cov"]).max()
        assert np.abs(s["conf_sigma"][b] - ref["conf_sigma"]).max() <= 1e-8 * ref["conf_sigma"].max()
    # underdetermined (m <= n + q) is reported, not computed
    xs = np.linspace(0, 1, 5)
    mdl5 = double_exp_builder_model(xs, [0.3, 2.0])
    bp5 = vp.BatchProblem(mdl5, (np.exp(-xs / 0.3) + 1)[None, :], x=xs)
    bp5.set_params(np.array([[0.3, 2.0]]))
    assert bp5.statistics()["status"][0] == 4
    bp.close()
    bp5.close()
