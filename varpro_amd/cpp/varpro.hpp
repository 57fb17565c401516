// varpro.hpp -- C++17 host-side mirror of the reference's public surface over the C ABI (include/varpro_hip.h).
//
// The reference (geo-ant/varpro) is compiled code (Rust); its toolchain is absent from the build image, so the
// host side above the C ABI is written in C++ (header-only) with the reference's names, argument meaning and
// error behaviour:
//
//   varpro::SeparableModelBuilder / SeparableModel      src/model/builder/mod.rs:252-571, src/model/mod.rs:239-517
//   varpro::SeparableProblemBuilder / SeparableProblem  src/problem/builder.rs:116-324, src/problem.rs:57-212,
//                                                        impl LeastSquaresProblem: src/solvers/levmar/mod.rs:22-202
//   varpro::LevMarSolver / FitResult / MinimizationReport / LevenbergMarquardt
//                                                        src/solvers/levmar/mod.rs:204-315, src/fit.rs
//   varpro::BatchProblem                                 new: B independent problems per launch
//
// Rust's `Result<T, E>` / `Option<T>` map to exceptions / std::optional; matrices are column-major
// std::vector<double> exactly as nalgebra stores them.  Closures are replaced by basis kinds.
// Link with -lvarpro_hip (varpro_amd/lib).  No CPU fallback: without a gfx950 device every compute call throws.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/varpro_hip.h"

namespace varpro {

enum class Basis : int32_t {
    Const = VP_BASIS_CONST,
    ExpDecay = VP_BASIS_EXP_DECAY,
    ExpRate = VP_BASIS_EXP_RATE,
    ExpCos = VP_BASIS_EXP_COS,
    SinPhase = VP_BASIS_SIN_PHASE
};
inline int arity(Basis k) {
    switch (k) {
    case Basis::Const: return 0;
    case Basis::ExpDecay:
    case Basis::ExpRate: return 1;
    default: return 2;
    }
}

// == ModelBuildError (src/model/builder/error.rs): `variant` is the Rust enum variant name
struct ModelBuildError : std::runtime_error {
    std::string variant;
    ModelBuildError(std::string v, const std::string &msg) : std::runtime_error(v + ": " + msg), variant(std::move(v)) {}
};
// == ModelError (src/model/errors.rs): an error of the model AT EVALUATION time; UnexpectedFunctionOutput carries the
// expected and the actual column length (src/model/model_basis_function.rs:70)
struct ModelError : std::runtime_error {
    std::string variant;
    size_t function_index, expected_length, actual_length;
    ModelError(std::string v, size_t j, size_t expected, size_t actual)
        : std::runtime_error(v + ": basis function " + std::to_string(j) + " returned " + std::to_string(actual) + " elements, expected " +
                             std::to_string(expected)),
          variant(std::move(v)), function_index(j), expected_length(expected), actual_length(actual) {}
};
// == SeparableProblemBuilderError (src/problem/builder.rs:15-46)
struct SeparableProblemBuilderError : std::runtime_error {
    std::string variant;
    SeparableProblemBuilderError(std::string v, const std::string &msg)
        : std::runtime_error(v + ": " + msg), variant(std::move(v)) {}
};
// call-level failure of the C ABI (VP_ERR_*)
struct HipError : std::runtime_error {
    int code;
    HipError(int c, const std::string &msg) : std::runtime_error("varpro_hip error " + std::to_string(c) + ": " + msg), code(c) {}
};
inline void check(int rc) {
    if (rc != 0) throw HipError(rc, vp_last_error());
}

// == SeparableModel: the descriptor the builder produces + the SeparableNonlinearModel trait surface
class SeparableModel {
  public:
    std::vector<std::string> parameter_names;
    vp_model_desc desc{};
    std::vector<double> x;
    std::vector<double> alpha;

    size_t parameter_count() const { return (size_t)desc.n_params; }      // src/model/mod.rs:256
    size_t base_function_count() const { return (size_t)desc.n_basis; }   // :259
    size_t output_len() const { return x.size(); }                        // :263
    void set_params(const std::vector<double> &p) {                       // :266-267
        if (p.size() != parameter_count()) throw std::invalid_argument("IncorrectParameterCount");
        alpha = p;
    }
    const std::vector<double> &params() const { return alpha; }           // :271
};

// == SeparableModelBuilder (src/model/builder/mod.rs:338-525); errors are latched and reported by build()
class SeparableModelBuilder {
    struct Fn {
        std::vector<std::string> params;
        Basis kind;
        std::vector<std::string> derivs;
    };
    std::vector<std::string> names_;
    std::vector<Fn> fns_;
    std::vector<double> x_, init_;
    bool have_x_ = false, have_init_ = false, building_ = false;
    std::optional<ModelBuildError> err_;
    void fail(const char *v, const std::string &m) {
        if (!err_) err_.emplace(v, m);
    }
    static bool contains(const std::vector<std::string> &v, const std::string &s) {
        for (auto &e : v)
            if (e == s) return true;
        return false;
    }

  public:
    explicit SeparableModelBuilder(std::vector<std::string> parameter_names) : names_(std::move(parameter_names)) {
        if (names_.empty()) fail("EmptyParameters", "A function or model parameter list is empty!");
        for (size_t i = 0; i < names_.size(); ++i) {
            if (names_[i].find(',') != std::string::npos) fail("CommaInParameterNameNotAllowed", names_[i]);
            for (size_t j = i + 1; j < names_.size(); ++j)
                if (names_[i] == names_[j]) fail("DuplicateParameterNames", "Parameter list contains duplicates!");
        }
    }
    SeparableModelBuilder &invariant_function(Basis kind = Basis::Const) {
        if (err_) return *this;
        if (arity(kind) != 0) fail("IncorrectParameterCount", "invariant function takes no parameters");
        else fns_.push_back({{}, kind, {}});
        building_ = false;
        return *this;
    }
    SeparableModelBuilder &function(std::vector<std::string> params, Basis kind) {
        if (err_) return *this;
        if (params.empty()) return fail("EmptyParameters", "A function or model parameter list is empty!"), *this;
        for (size_t i = 0; i < params.size(); ++i) {
            for (size_t j = i + 1; j < params.size(); ++j)
                if (params[i] == params[j]) return fail("DuplicateParameterNames", "duplicates in function parameters"), *this;
            if (!contains(names_, params[i]))
                return fail("FunctionParameterNotInModel", "Function parameter '" + params[i] + "' is not part of the model parameters."), *this;
        }
        if ((size_t)arity(kind) != params.size()) return fail("IncorrectParameterCount", "Incorrect number of parameters for function"), *this;
        fns_.push_back({std::move(params), kind, {}});
        building_ = true;
        return *this;
    }
    SeparableModelBuilder &partial_deriv(const std::string &parameter) {
        if (err_) return *this;
        if (!building_) return fail("IllegalCallToPartialDeriv", "can only follow 'function' or 'partial_deriv'"), *this;
        Fn &f = fns_.back();
        if (!contains(f.params, parameter)) return fail("InvalidDerivative", "Parameter '" + parameter + "' does not exist in parameter list"), *this;
        if (contains(f.derivs, parameter)) return fail("DuplicateDerivative", "Derivative for parameter '" + parameter + "' was already provided!"), *this;
        f.derivs.push_back(parameter);
        return *this;
    }
    SeparableModelBuilder &independent_variable(std::vector<double> x) {
        x_ = std::move(x);
        have_x_ = true;
        building_ = false;
        return *this;
    }
    SeparableModelBuilder &initial_parameters(std::vector<double> p) {
        if (err_) return *this;
        if (p.size() != names_.size()) return fail("IncorrectParameterCount", "wrong number of initial parameters"), *this;
        init_ = std::move(p);
        have_init_ = true;
        building_ = false;
        return *this;
    }
    SeparableModel build() const {
        if (err_) throw *err_;
        if (fns_.empty()) throw ModelBuildError("EmptyModel", "Tried to construct model with no functions.");
        for (auto &f : fns_)
            for (auto &p : f.params)
                if (!contains(f.derivs, p)) throw ModelBuildError("MissingDerivative", "missing derivative for parameter '" + p + "'");
        for (auto &n : names_) {
            bool used = false;
            for (auto &f : fns_) used = used || contains(f.params, n);
            if (!used) throw ModelBuildError("UnusedParameter", "Model depends on parameter '" + n + "', but none of its functions use it.");
        }
        if (!have_x_) throw ModelBuildError("MissingX", "Missing vector for independent variable x");
        if (!have_init_) throw ModelBuildError("MissingInitialParameters", "Missing initial guesses for model parameters");
        if (fns_.size() > VP_MAX_BASIS || names_.size() > VP_MAX_PARAMS) throw ModelBuildError("IncorrectParameterCount", "model exceeds VP_MAX_BASIS/VP_MAX_PARAMS");
        SeparableModel m;
        m.parameter_names = names_;
        m.x = x_;
        m.alpha = init_;
        m.desc.n_basis = (int32_t)fns_.size();
        m.desc.n_params = (int32_t)names_.size();
        for (int j = 0; j < VP_MAX_BASIS; ++j) {
            m.desc.kind[j] = 0;
            for (int a = 0; a < VP_MAX_BASIS_PARAMS; ++a) m.desc.param[j][a] = -1;
        }
        for (size_t j = 0; j < fns_.size(); ++j) {
            m.desc.kind[j] = (int32_t)fns_[j].kind;
            for (size_t a = 0; a < fns_[j].params.size(); ++a)
                for (size_t k = 0; k < names_.size(); ++k)
                    if (names_[k] == fns_[j].params[a]) m.desc.param[j][a] = (int32_t)k;
        }
        return m;
    }
};

// == levenberg_marquardt::LevenbergMarquardt builder knobs (src/solvers/levmar/mod.rs:221-223)
struct LevenbergMarquardt {
    vp_lm_opts o;
    LevenbergMarquardt() { vp_lm_opts_default(&o, VP_F64); }
    LevenbergMarquardt &with_ftol(double v) { return o.ftol = v, *this; }
    LevenbergMarquardt &with_xtol(double v) { return o.xtol = v, *this; }
    LevenbergMarquardt &with_gtol(double v) { return o.gtol = v, *this; }
    LevenbergMarquardt &with_stepbound(double v) { return o.stepbound = v, *this; }
    LevenbergMarquardt &with_patience(int v) { return o.patience = v, *this; }
    LevenbergMarquardt &with_scale_diag(bool v) { return o.scale_diag = v ? 1 : 0, *this; }
};

// ---- Student-t quantile: the scalar the reference takes from the `distrs` crate (src/statistics/mod.rs:285-288) to
// turn the unscaled confidence sigma into confidence_band_radius(p) = sigma * t_ppf((1+p)/2, dof) ------------------
namespace detail {
// regularised incomplete beta function I_x(a, b) (Lentz continued fraction)
inline double betainc(double a, double b, double x) {
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double lbeta = std::lgamma(a) + std::lgamma(b) - std::lgamma(a + b);
    const double front = std::exp(a * std::log(x) + b * std::log1p(-x) - lbeta);
    auto cf = [](double a_, double b_, double x_) {
        const double tiny = 1e-300;
        double c = 1.0, d = 1.0 - (a_ + b_) * x_ / (a_ + 1.0);
        if (std::fabs(d) < tiny) d = tiny;
        d = 1.0 / d;
        double h = d;
        for (int m = 1; m <= 500; ++m) {
            const double m2 = 2.0 * m;
            double aa = m * (b_ - m) * x_ / ((a_ + m2 - 1.0) * (a_ + m2));
            d = 1.0 + aa * d;
            if (std::fabs(d) < tiny) d = tiny;
            c = 1.0 + aa / c;
            if (std::fabs(c) < tiny) c = tiny;
            d = 1.0 / d;
            h *= d * c;
            aa = -(a_ + m) * (a_ + b_ + m) * x_ / ((a_ + m2) * (a_ + m2 + 1.0));
            d = 1.0 + aa * d;
            if (std::fabs(d) < tiny) d = tiny;
            c = 1.0 + aa / c;
            if (std::fabs(c) < tiny) c = tiny;
            d = 1.0 / d;
            const double del = d * c;
            h *= del;
            if (std::fabs(del - 1.0) < 1e-16) break;
        }
        return h;
    };
    if (x < (a + 1.0) / (a + b + 2.0)) return front * cf(a, b, x) / a;
    return 1.0 - front * cf(b, a, 1.0 - x) / b;
}
inline double student_t_cdf(double t, double dof) {
    const double x = dof / (dof + t * t);
    const double tail = 0.5 * betainc(0.5 * dof, 0.5, x);
    return t >= 0.0 ? 1.0 - tail : tail;
}
} // namespace detail

// t such that P(T <= t) = p for Student's t with `dof` degrees of freedom (bisection on the CDF; |error| < 1e-12 t)
inline double student_t_quantile(double p, double dof) {
    if (!(p > 0.0 && p < 1.0) || !(dof > 0.0)) throw std::invalid_argument("student_t_quantile: p in (0,1), dof > 0");
    if (p == 0.5) return 0.0;
    const bool upper = p > 0.5;
    const double pp = upper ? p : 1.0 - p;
    double lo = 0.0, hi = 1.0;
    while (detail::student_t_cdf(hi, dof) < pp && hi < 1e300) hi *= 2.0;
    for (int it = 0; it < 200 && hi - lo > 1e-14 * hi; ++it) {
        const double mid = 0.5 * (lo + hi);
        (detail::student_t_cdf(mid, dof) < pp ? lo : hi) = mid;
    }
    const double t = 0.5 * (lo + hi);
    return upper ? t : -t;
}

// B independent problems x S right-hand sides on one GPU (host-pointer mode; fp64)
class BatchProblem {
    vp_batch *h_ = nullptr;

  public:
    int64_t m = 0, S = 1, B = 0;
    int n = 0, q = 0;
    BatchProblem() = default;
    BatchProblem(const SeparableModel &model, const std::vector<double> &Y, int64_t B_, int64_t S_,
                 const std::vector<double> *weights = nullptr, double epsilon = -1.0, int device = 0) {
        m = (int64_t)model.output_len();
        B = B_;
        S = S_;
        n = model.desc.n_basis;
        q = model.desc.n_params;
        if ((int64_t)Y.size() != B * S * m) throw std::invalid_argument("Y must hold B*S*m values");
        check(vp_batch_create(&h_, &model.desc, VP_F64, m, S, B, model.x.data(), Y.data(), weights ? weights->data() : nullptr,
                              epsilon, VP_FLAG_OWN_STREAM, device, nullptr));
    }
    BatchProblem(const BatchProblem &) = delete;
    BatchProblem &operator=(const BatchProblem &) = delete;
    BatchProblem(BatchProblem &&o) noexcept { *this = std::move(o); }
    BatchProblem &operator=(BatchProblem &&o) noexcept {
        std::swap(h_, o.h_);
        m = o.m, S = o.S, B = o.B, n = o.n, q = o.q;
        return *this;
    }
    ~BatchProblem() {
        if (h_) vp_batch_destroy(h_);
    }
    vp_batch *handle() const { return h_; }
    void set_params(const std::vector<double> &alpha) { check(vp_set_params(h_, alpha.data())); }
    // next frame of a stream of same-shaped data: new observations, same model / grid / weights, no re-allocation
    void set_observations(const std::vector<double> &Y) {
        if ((int64_t)Y.size() != B * S * m) throw std::invalid_argument("Y must hold B*S*m values");
        check(vp_set_observations(h_, Y.data()));
    }
    std::vector<int32_t> status() const {
        std::vector<int32_t> st((size_t)B);
        check(vp_linear_coeffs(h_, nullptr, st.data()));
        return st;
    }
    std::vector<double> residuals() const {
        std::vector<double> r((size_t)(B * S * m));
        check(vp_residuals(h_, r.data(), nullptr));
        return r;
    }
    std::vector<double> jacobian() const {
        std::vector<double> J((size_t)(B * q * S * m));
        check(vp_jacobian(h_, J.data(), nullptr));
        return J;
    }
    std::vector<double> linear_coefficients() const {
        std::vector<double> C((size_t)(B * S * n));
        check(vp_linear_coeffs(h_, C.data(), nullptr));
        return C;
    }
    std::vector<double> best_fit() const {
        std::vector<double> f((size_t)(B * S * m));
        check(vp_best_fit(h_, f.data()));
        return f;
    }
    std::vector<vp_report> fit(std::vector<double> &alpha_inout, const LevenbergMarquardt &solver = LevenbergMarquardt()) {
        std::vector<vp_report> rep((size_t)B);
        check(vp_fit(h_, &solver.o, alpha_inout.data(), nullptr, rep.data()));
        return rep;
    }
    // == FitStatistics::try_calculate for every problem (src/statistics/mod.rs:352-441); single RHS only
    struct Statistics {
        std::vector<double> covariance;   // [B][(n+q)^2] column-major, ordering [linear coefficients, parameters]
        std::vector<double> reduced_chi2; // [B]
        std::vector<double> conf_sigma;   // [B][m]  multiply by the Student-t quantile for a confidence band
        std::vector<int32_t> status;      // [B]  0 ok, 4 = Underdetermined / MatrixInversion
        int64_t m = 0, dof = 0;
        // == FitStatistics::confidence_band_radius (src/statistics/mod.rs:271-304) of problem b
        std::vector<double> confidence_band_radius(int64_t b, double probability) const {
            if (!(probability > 0.0 && probability < 1.0)) throw std::invalid_argument("probability must be in (0,1)");
            const double tq = student_t_quantile(0.5 * (1.0 + probability), (double)dof);
            std::vector<double> r((size_t)m);
            for (int64_t i = 0; i < m; ++i) r[(size_t)i] = tq * conf_sigma[(size_t)(b * m + i)];
            return r;
        }
    };
    Statistics statistics() const {
        Statistics s;
        const size_t k = (size_t)(n + q);
        s.covariance.resize((size_t)B * k * k);
        s.reduced_chi2.resize((size_t)B);
        s.conf_sigma.resize((size_t)(B * m));
        s.status.resize((size_t)B);
        s.m = m;
        s.dof = m - (int64_t)k;
        check(vp_statistics(h_, s.covariance.data(), s.reduced_chi2.data(), s.conf_sigma.data(), s.status.data()));
        return s;
    }
    // one global fit sharded by right-hand sides over ranks: `fn` sums `count` device doubles over all ranks on
    // the given HIP stream (ncclAllReduce on RCCL); see include/varpro_hip.h
    void set_rhs_allreduce(vp_allreduce_fn fn, void *user, int64_t global_rhs_count) {
        check(vp_set_rhs_allreduce(h_, fn, user, global_rhs_count));
    }
};

// ---- models OUTSIDE the descriptor language: the reference's closure-based `SeparableModel` (src/model/mod.rs:367-517),
// built as `SeparableModelBuilder::function(params, f).partial_deriv(param, df)` (src/model/builder/mod.rs:338-525) with REAL
// C++ callables.  f(x, params) -> m values.  The model is evaluated where it lives (the host); everything downstream of
// eval() / eval_partial_deriv(k) runs on the device (vp_batch_create_external, include/varpro_hip.h).
class ClosureModel {
  public:
    using Fn = std::function<std::vector<double>(const std::vector<double> &x, const std::vector<double> &params)>;

  private:
    struct Basis_ {
        std::vector<int> idx;        // model parameter indices of the function's parameter list
        Fn f;
        std::map<int, Fn> derivs;    // model parameter index -> d f / d parameter
    };
    std::vector<std::string> names_;
    std::vector<double> x_;
    std::vector<Basis_> fns_;
    int find(const std::string &nm) const {
        for (size_t i = 0; i < names_.size(); ++i)
            if (names_[i] == nm) return (int)i;
        return -1;
    }

  public:
    ClosureModel(std::vector<std::string> parameter_names, std::vector<double> x) : names_(std::move(parameter_names)), x_(std::move(x)) {
        if (names_.empty()) throw ModelBuildError("EmptyParameters", "A function or model parameter list is empty!");
        for (size_t i = 0; i < names_.size(); ++i) {
            if (names_[i].find(',') != std::string::npos)
                throw ModelBuildError("CommaInParameterNameNotAllowed", "Parameter names may not contain comma separator");
            for (size_t j = 0; j < i; ++j)
                if (names_[i] == names_[j]) throw ModelBuildError("DuplicateParameterNames", "Parameter list contains duplicates!");
        }
    }
    ClosureModel &invariant_function(Fn f) {
        fns_.push_back(Basis_{{}, std::move(f), {}});
        return *this;
    }
    ClosureModel &function(const std::vector<std::string> &params, Fn f) {
        if (params.empty()) throw ModelBuildError("EmptyParameters", "A function or model parameter list is empty!");
        Basis_ b;
        for (const auto &nm : params) {
            const int k = find(nm);
            if (k < 0) throw ModelBuildError("FunctionParameterNotInModel", "Function parameter '" + nm + "' is not part of the model parameters.");
            for (int kk : b.idx)
                if (kk == k) throw ModelBuildError("DuplicateParameterNames", "Parameter list contains duplicates!");
            b.idx.push_back(k);
        }
        b.f = std::move(f);
        fns_.push_back(std::move(b));
        return *this;
    }
    ClosureModel &partial_deriv(const std::string &param, Fn df) {
        if (fns_.empty() || fns_.back().idx.empty()) throw ModelBuildError("IllegalCallToPartialDeriv", "partial_deriv needs a preceding function");
        const int k = find(param);
        Basis_ &b = fns_.back();
        bool in = false;
        for (int kk : b.idx) in = in || kk == k;
        if (k < 0 || !in) throw ModelBuildError("InvalidDerivative", "Parameter '" + param + "' is not in the function's parameter list");
        if (b.derivs.count(k)) throw ModelBuildError("DuplicateDerivative", "Derivative for parameter '" + param + "' was already provided!");
        b.derivs[k] = std::move(df);
        return *this;
    }
    // == SeparableModelBuilder::build (src/model/builder/mod.rs:527-553)
    void validate() const {
        if (fns_.empty()) throw ModelBuildError("EmptyModel", "Tried to construct model with no functions.");
        std::vector<bool> used(names_.size(), false);
        for (const auto &b : fns_)
            for (int k : b.idx) {
                used[(size_t)k] = true;
                if (!b.derivs.count(k)) throw ModelBuildError("MissingDerivative", "missing derivative for parameter '" + names_[(size_t)k] + "'");
            }
        for (size_t k = 0; k < used.size(); ++k)
            if (!used[k]) throw ModelBuildError("UnusedParameter", "Parameter '" + names_[k] + "' is not used by any function of the model");
    }
    size_t parameter_count() const { return names_.size(); }
    size_t base_function_count() const { return fns_.size(); }
    size_t output_len() const { return x_.size(); }
    // the (basis j, parameter k) pairs with a non-zero derivative, in the column order of derivs()
    std::vector<std::pair<int, int>> pairs() const {
        std::vector<std::pair<int, int>> p;
        for (size_t j = 0; j < fns_.size(); ++j)
            for (int k : fns_[j].idx) p.emplace_back((int)j, k);
        return p;
    }
    // == eval() (src/model/mod.rs:441-471) for a batch of parameter sets alpha [B][q]: Phi [B][n][m], UNWEIGHTED
    std::vector<double> eval_batch(const std::vector<double> &alpha, int64_t B) const {
        const size_t q = names_.size(), n = fns_.size(), m = x_.size();
        std::vector<double> out((size_t)B * n * m);
        for (int64_t b = 0; b < B; ++b)
            for (size_t j = 0; j < n; ++j) {
                std::vector<double> pr;
                for (int k : fns_[j].idx) pr.push_back(alpha[(size_t)b * q + (size_t)k]);
                const std::vector<double> v = fns_[j].f(x_, pr);
                // == ModelError::UnexpectedFunctionOutput (src/model/model_basis_function.rs:70): a column of the wrong length is
                // an error of the model, never a write past (or short of) its slot
                if (v.size() != m) throw ModelError("UnexpectedFunctionOutput", j, m, v.size());
                std::copy(v.begin(), v.end(), out.begin() + ((size_t)b * n + j) * m);
            }
        return out;
    }
    // == the non-zero columns of eval_partial_deriv(k) (src/model/mod.rs:473-512) in pair order: dPhi [B][p][m]
    std::vector<double> derivs_batch(const std::vector<double> &alpha, int64_t B) const {
        const size_t q = names_.size(), m = x_.size();
        const auto prs = pairs();
        std::vector<double> out((size_t)B * prs.size() * m);
        for (int64_t b = 0; b < B; ++b)
            for (size_t p = 0; p < prs.size(); ++p) {
                const Basis_ &bs = fns_[(size_t)prs[p].first];
                std::vector<double> pr;
                for (int k : bs.idx) pr.push_back(alpha[(size_t)b * q + (size_t)k]);
                const std::vector<double> v = bs.derivs.at(prs[p].second)(x_, pr);
                if (v.size() != m) throw ModelError("UnexpectedDerivativeOutput", (size_t)prs[p].first, m, v.size());
                std::copy(v.begin(), v.end(), out.begin() + ((size_t)b * prs.size() + p) * m);
            }
        return out;
    }
};

// B problems of one closure model (single right-hand side, fp64, host-pointer mode): trait-level evaluation and the
// batched LM fit by reverse communication -- LevMarSolver::fit over any SeparableNonlinearModel
// (src/solvers/levmar/mod.rs:238-254): the LM drivers on the device, the model with the caller
class ExternalBatchProblem {
    vp_batch *h_ = nullptr;
    // the model is COPIED (closures are shared_ptr-backed std::function objects: cheap): a problem built from a temporary
    // model must not dangle, and the reference's problem owns its model too (src/problem.rs:55-70)
    std::shared_ptr<const ClosureModel> model_;

  public:
    int64_t m = 0, B = 0, S = 1;
    int n = 0, q = 0, np = 0;
    // Y: [B][S][m] -- S right-hand sides per problem share the nonlinear parameters (SeparableProblemBuilder::mrhs,
    // src/problem/builder.rs:194-225); S = 1: the single-right-hand-side problem
    ExternalBatchProblem(const ClosureModel &model, const std::vector<double> &Y, int64_t B_, const std::vector<double> *weights = nullptr,
                         double epsilon = -1.0, int device = 0, int64_t S_ = 1)
        : model_(std::make_shared<const ClosureModel>(model)) {
        model.validate();
        m = (int64_t)model.output_len();
        B = B_;
        S = S_;
        n = (int)model.base_function_count();
        q = (int)model.parameter_count();
        const auto prs = model.pairs();
        np = (int)prs.size();
        std::vector<int32_t> pb, pp;
        for (const auto &pr : prs) {
            pb.push_back(pr.first);
            pp.push_back(pr.second);
        }
        if ((int64_t)Y.size() != B * S * m) throw std::invalid_argument("Y must hold B*S*m values");
        check(vp_batch_create_external(&h_, n, q, np, pb.data(), pp.data(), VP_F64, m, S, B, Y.data(), weights ? weights->data() : nullptr,
                                       epsilon, VP_FLAG_OWN_STREAM, device, nullptr));
    }
    ExternalBatchProblem(const ExternalBatchProblem &) = delete;
    ExternalBatchProblem &operator=(const ExternalBatchProblem &) = delete;
    ~ExternalBatchProblem() {
        if (h_) vp_batch_destroy(h_);
    }
    struct Evaluation {
        std::vector<double> residuals, jacobian, coefficients, cost; // [B][m], [B][q][m], [B][n], [B]
        std::vector<int32_t> status;
    };
    // set_params + residuals + jacobian of every problem at alpha [B][q] (src/solvers/levmar/mod.rs:42-73, 91-95, 101-201)
    Evaluation evaluate(const std::vector<double> &alpha) const {
        const std::vector<double> Phi = model_->eval_batch(alpha, B), dPhi = model_->derivs_batch(alpha, B);
        Evaluation e;
        e.residuals.resize((size_t)(B * S * m));      // [B][S][m]
        e.jacobian.resize((size_t)(B * q * S * m));   // [B][q][S][m]
        e.coefficients.resize((size_t)(B * S * n));   // [B][S][n]
        e.cost.resize((size_t)(B * S));
        e.status.resize((size_t)(B * S));
        check(vp_evaluate_with_basis(h_, alpha.data(), Phi.data(), dPhi.data(), e.residuals.data(), e.jacobian.data(),
                                     e.coefficients.data(), e.cost.data(), e.status.data()));
        return e;
    }
    struct BatchFit {
        std::vector<double> nonlinear_parameters, linear_coefficients; // [B][q], [B][n]
        std::vector<vp_report> reports;                                // termination > 0 <=> FitResult Ok (:249-253)
        int steps = 0;
    };
    BatchFit fit(const std::vector<double> &alpha0, const LevenbergMarquardt &solver = LevenbergMarquardt(), bool derivatives_on_accept = false) {
        check(vp_fit_begin(h_, &solver.o, alpha0.data(), derivatives_on_accept ? VP_FIT_DERIVATIVES_ON_ACCEPT : 0));
        std::vector<double> trial = alpha0;
        std::vector<int32_t> want((size_t)B, VP_WANT_BASIS | VP_WANT_DERIVATIVES);
        int64_t active = B;
        BatchFit out;
        const int limit = 2 * (solver.o.patience * (q + 1) + 2);
        while (active > 0 && out.steps < limit) {
            // (the columns of every problem are evaluated: a model may skip the problems whose want word is 0)
            const std::vector<double> Phi = model_->eval_batch(trial, B), dPhi = model_->derivs_batch(trial, B);
            check(vp_fit_step_with_basis(h_, Phi.data(), dPhi.data(), trial.data(), want.data(), &active));
            ++out.steps;
        }
        out.nonlinear_parameters.resize((size_t)(B * q));
        out.linear_coefficients.resize((size_t)(B * S * n)); // [B][S][n]
        out.reports.resize((size_t)B);
        check(vp_fit_end(h_, out.nonlinear_parameters.data(), out.linear_coefficients.data(), out.reports.data()));
        return out;
    }
};

// == SeparableProblem (single problem, S right-hand sides); `None` of the reference == std::nullopt
class SeparableProblem {
    SeparableModel model_;
    BatchProblem batch_;
    std::optional<std::vector<double>> weights_;

  public:
    SeparableProblem(SeparableModel model, const std::vector<double> &Y, int64_t S, std::optional<std::vector<double>> weights,
                     double epsilon)
        : model_(std::move(model)), batch_(model_, Y, 1, S, weights ? &*weights : nullptr, epsilon), weights_(std::move(weights)) {
        set_params(model_.params()); // build(): initial set_params (src/problem/builder.rs:321)
    }
    void set_params(const std::vector<double> &p) { // src/solvers/levmar/mod.rs:42-73
        model_.set_params(p);
        batch_.set_params(p);
    }
    const std::vector<double> &params() const { return model_.params(); } // :80-82
    std::optional<std::vector<double>> residuals() const {               // :91-95
        if (batch_.status()[0] != VP_ST_OK) return std::nullopt;
        return batch_.residuals();
    }
    std::optional<std::vector<double>> jacobian() const { // :101-201  (m*S) x q column-major
        if (batch_.status()[0] != VP_ST_OK) return std::nullopt;
        return batch_.jacobian();
    }
    std::optional<std::vector<double>> linear_coefficients() const { // src/problem.rs:142-183  n x S column-major
        if (batch_.status()[0] != VP_ST_OK) return std::nullopt;
        return batch_.linear_coefficients();
    }
    const SeparableModel &model() const { return model_; }
    SeparableModel &model_mut() { return model_; }
    BatchProblem &batch() { return batch_; }
    const BatchProblem &batch() const { return batch_; }
};

// == SeparableProblemBuilder::{new, mrhs, observations, weights, epsilon, build}
class SeparableProblemBuilder {
    SeparableModel model_;
    bool mrhs_;
    std::optional<std::vector<double>> Y_;
    int64_t S_ = 1;
    std::optional<std::vector<double>> weights_;
    double eps_ = -1.0;
    SeparableProblemBuilder(SeparableModel m, bool mrhs) : model_(std::move(m)), mrhs_(mrhs) {}

  public:
    static SeparableProblemBuilder new_(SeparableModel m) { return SeparableProblemBuilder(std::move(m), false); }
    static SeparableProblemBuilder mrhs(SeparableModel m) { return SeparableProblemBuilder(std::move(m), true); }
    // single RHS: y (m values);  MRHS: Y column-major m x S
    SeparableProblemBuilder &observations(std::vector<double> Y, int64_t S = 1) {
        Y_ = std::move(Y);
        S_ = mrhs_ ? S : 1;
        return *this;
    }
    SeparableProblemBuilder &weights(std::vector<double> w) { return weights_ = std::move(w), *this; }
    SeparableProblemBuilder &epsilon(double e) { return eps_ = std::fabs(e), *this; }
    SeparableProblem build() {
        if (!Y_) throw SeparableProblemBuilderError("YDataMissing", "Right hand side(s) not provided");
        const size_t xlen = model_.output_len();
        if (xlen == 0 || Y_->empty()) throw SeparableProblemBuilderError("ZeroLengthVector", "x or y must have nonzero number of elements.");
        if (Y_->size() != xlen * (size_t)S_)
            throw SeparableProblemBuilderError("InvalidLengthOfData", "Vectors x and y must have same lengths.");
        if (weights_ && weights_->size() != xlen)
            throw SeparableProblemBuilderError("InvalidLengthOfWeights", "The weights must have the same length as the data y.");
        return SeparableProblem(model_, *Y_, S_, weights_, eps_);
    }
};

// == MinimizationReport / FitResult (src/fit.rs)
struct MinimizationReport {
    int termination;
    int number_of_evaluations;
    double objective_function;
    bool was_successful() const { return termination > 0; }
};
struct FitResult {
    SeparableProblem problem;
    MinimizationReport minimization_report;
    const std::vector<double> &nonlinear_parameters() const { return problem.model().params(); } // src/fit.rs:113-115
    std::optional<std::vector<double>> linear_coefficients() const { return problem.linear_coefficients(); }
    std::optional<std::vector<double>> best_fit() const { // unweighted Phi C (src/fit.rs:55-59, 87-91)
        if (!problem.linear_coefficients()) return std::nullopt;
        return problem.batch().best_fit();
    }
    bool was_successful() const { return minimization_report.was_successful(); } // :120-122
};
// the Err(FitResult) arm of LevMarSolver::fit (src/solvers/levmar/mod.rs:248-253)
struct FitError : std::runtime_error {
    FitResult result;
    explicit FitError(FitResult r) : std::runtime_error("fit did not terminate successfully"), result(std::move(r)) {}
};

class LevMarSolver {
    LevenbergMarquardt solver_;

  public:
    LevMarSolver() = default;
    static LevMarSolver with_solver(LevenbergMarquardt s) {
        LevMarSolver l;
        l.solver_ = s;
        return l;
    }
    // consumes the problem; returns the FitResult on success, throws FitError(result) otherwise
    FitResult fit(SeparableProblem problem) const {
        std::vector<double> alpha = problem.params();
        std::vector<vp_report> rep = problem.batch().fit(alpha, solver_);
        problem.model_mut().set_params(alpha);
        FitResult res{std::move(problem), MinimizationReport{rep[0].termination, rep[0].n_evals, rep[0].objective}};
        if (!res.was_successful()) throw FitError(std::move(res));
        return res;
    }
};

} // namespace varpro
