"""Host-side mirror of ``LevMarSolver`` / ``FitResult`` (src/solvers/levmar/mod.rs:204-315, src/fit.rs).

``fit`` runs the device-resident Levenberg-Marquardt kernel (``vp_fit``); success is decided solely by
the termination reason, as in the reference (src/solvers/levmar/mod.rs:249-253).
"""
import numpy as np

from .batch import LevenbergMarquardt

TERMINATION_NAMES = {
    1: "ResidualsZero", 2: "Orthogonal", 3: "Converged{ftol}", 4: "Converged{xtol}", 5: "Converged{ftol,xtol}",
    0: "NotRun", -1: "User", -2: "Numerical", -3: "NoImprovementPossible", -4: "LostPatience", -5: "NoParameters",
    -6: "NoResiduals", -7: "WrongDimensions",
}


class TerminationReason:
    def __init__(self, code):
        self.code = int(code)

    def was_successful(self):
        return self.code > 0

    def __repr__(self):
        return "TerminationReason(%s)" % TERMINATION_NAMES.get(self.code, self.code)


class MinimizationReport:
    """== levenberg_marquardt::MinimizationReport (src/fit.rs:24-29)"""

    def __init__(self, termination, number_of_evaluations, objective_function):
        self.termination = TerminationReason(termination)
        self.number_of_evaluations = int(number_of_evaluations)
        self.objective_function = float(objective_function)

    def __repr__(self):
        return "MinimizationReport(%r, evals=%d, objective=%.6g)" % (self.termination, self.number_of_evaluations,
                                                                      self.objective_function)


class FitResult:
    """== ``FitResult`` (src/fit.rs:15-29): the final problem + the minimization report"""

    def __init__(self, problem, minimization_report):
        self.problem = problem
        self.minimization_report = minimization_report

    def nonlinear_parameters(self):
        return self.problem.model().params()  # src/fit.rs:113-115

    def linear_coefficients(self):
        return self.problem.linear_coefficients()  # src/fit.rs:45, 73

    def best_fit(self):
        """UNWEIGHTED Phi(alpha) C (src/fit.rs:55-59, 87-91); None if no coefficients"""
        if self.problem.linear_coefficients() is None:
            return None
        f = self.problem._batch.best_fit()
        return np.ascontiguousarray(f[0].T) if self.problem._mrhs else f[0].copy()

    def was_successful(self):
        return self.minimization_report.termination.was_successful()  # src/fit.rs:120-122


class FitStatistics:
    """== ``FitStatistics`` (src/statistics/mod.rs:27-304) of one single-RHS fit"""

    def __init__(self, covariance_matrix, reduced_chi2, weighted_residuals, conf_sigma, n_linear, n_nonlinear, dof):
        self._cov = np.asarray(covariance_matrix)
        self._chi2 = float(reduced_chi2)
        self._wres = np.asarray(weighted_residuals)
        self._sigma = np.asarray(conf_sigma)
        self._n, self._q, self._dof = int(n_linear), int(n_nonlinear), int(dof)

    def covariance_matrix(self):
        """ordering [linear coefficients, nonlinear parameters] (src/statistics/mod.rs:129)"""
        return self._cov

    def calculate_correlation_matrix(self):
        d = np.sqrt(np.diag(self._cov))
        return self._cov / np.outer(d, d)

    correlation_matrix = calculate_correlation_matrix

    def weighted_residuals(self):
        return self._wres

    def reduced_chi2(self):
        return self._chi2

    def regression_standard_error(self):
        return float(np.sqrt(self._chi2))

    def linear_coefficients_variance(self):
        return np.diag(self._cov)[:self._n].copy()

    def nonlinear_parameters_variance(self):
        return np.diag(self._cov)[self._n:].copy()

    def confidence_band_radius(self, probability):
        """Student-t scaled confidence band (src/statistics/mod.rs:271-304)"""
        if not (np.isfinite(probability) and 0.0 < probability < 1.0):
            raise ValueError("probability must be in open interval (0.,1.)")
        from scipy import stats as _st
        return _st.t.ppf((probability + 1.0) / 2.0, self._dof) * self._sigma


class FitError(RuntimeError):
    """the ``Err(FitResult)`` arm of ``LevMarSolver::fit`` (src/solvers/levmar/mod.rs:248-253)"""

    def __init__(self, result):
        super().__init__("fit did not terminate successfully: %r" % (result.minimization_report,))
        self.result = result


class LevMarSolver:
    def __init__(self, solver=None):
        self._solver = solver

    @classmethod
    def default(cls):
        return cls()

    @classmethod
    def with_solver(cls, solver):
        return cls(solver)

    def fit(self, problem):
        """Consumes the problem (uses its current parameters as the initial guess).  Returns a FitResult
        on success, raises FitError(result) otherwise -- the Ok/Err split of the reference."""
        solver = self._solver or LevenbergMarquardt(problem.model().dtype)
        b = problem._batch
        alpha, _c, rep = b.fit(problem.params().reshape(1, -1), solver=solver)
        problem._model.set_params(np.asarray(alpha)[0])
        r = rep[0]
        result = FitResult(problem, MinimizationReport(r["termination"], r["n_evals"], r["objective"]))
        if not result.was_successful():
            raise FitError(result)
        return result

    def fit_with_statistics(self, problem):
        """== ``fit_with_statistics`` (src/solvers/levmar/mod.rs:275-304): single RHS only; raises FitError if
        the fit or the statistics (underdetermined / singular) fail"""
        if problem._mrhs:
            raise ValueError("fit_with_statistics is only supported for problems with a single right hand side")
        result = self.fit(problem)
        st = problem._batch.statistics()
        if int(st["status"][0]) != 0 or problem.linear_coefficients() is None:
            raise FitError(result)
        stats = FitStatistics(st["cov"][0], st["reduced_chi2"][0], problem.residuals(), st["conf_sigma"][0],
                              problem.n, problem.q, st["dof"])
        return result, stats
