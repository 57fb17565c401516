// C++ host-side mirror test: the reference's integration tests written against varpro.hpp.
//   mode "errors": builder error variants (no GPU needed)
//   mode "gpu"   : tests/integration_tests/main.rs:160-227 (double exponential, noise-free, 1e-8),
//                  :399-463 (MRHS S = 2), Jacobian == finite differences at the truth (src/solvers/levmar/test.rs:21-40)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../varpro_amd/cpp/varpro.hpp"

using namespace varpro;

static int failures = 0;
#define EXPECT(cond)                                                                     \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);               \
            ++failures;                                                                  \
        }                                                                                \
    } while (0)

// shared_test_code::linspace incl. its sign quirk (shared_test_code/src/lib.rs:20-34)
static std::vector<double> linspace_reference(double first, double last, size_t count) {
    std::vector<double> v(count);
    for (size_t n = 0; n < count; ++n) v[n] = first + (first - last) / double(count - 1) * double(n);
    return v;
}

static SeparableModel double_exp_model(const std::vector<double> &x, std::vector<double> guess) {
    return SeparableModelBuilder({"tau1", "tau2"})
        .initial_parameters(std::move(guess))
        .function({"tau1"}, Basis::ExpDecay).partial_deriv("tau1")
        .function({"tau2"}, Basis::ExpDecay).partial_deriv("tau2")
        .invariant_function(Basis::Const)
        .independent_variable(x)
        .build();
}

template <class F> static std::string variant_of(F &&f) {
    try {
        f();
    } catch (const ModelBuildError &e) {
        return e.variant;
    } catch (const SeparableProblemBuilderError &e) {
        return e.variant;
    }
    return "";
}

static void test_errors() {
    std::vector<double> x = {0, 1, 2, 3};
    EXPECT(variant_of([&] { SeparableModelBuilder({"a", "a"}).build(); }) == "DuplicateParameterNames");
    EXPECT(variant_of([&] { SeparableModelBuilder({}).build(); }) == "EmptyParameters");
    EXPECT(variant_of([&] { SeparableModelBuilder({"a"}).function({"b"}, Basis::ExpDecay).build(); }) == "FunctionParameterNotInModel");
    EXPECT(variant_of([&] {
               SeparableModelBuilder({"a"}).function({"a"}, Basis::ExpDecay).independent_variable(x).initial_parameters({1.0}).build();
           }) == "MissingDerivative");
    EXPECT(variant_of([&] { SeparableModelBuilder({"a"}).partial_deriv("a").build(); }) == "IllegalCallToPartialDeriv");
    EXPECT(variant_of([&] {
               SeparableModelBuilder({"a"}).function({"a"}, Basis::ExpDecay).partial_deriv("a").initial_parameters({1.0}).build();
           }) == "MissingX");
    SeparableModel m = double_exp_model(x, {1.0, 2.0});
    EXPECT(m.parameter_count() == 2 && m.base_function_count() == 3 && m.output_len() == 4);
    EXPECT(variant_of([&] { SeparableProblemBuilder::new_(m).build(); }) == "YDataMissing");
    EXPECT(variant_of([&] { SeparableProblemBuilder::new_(m).observations({1, 2, 3}).build(); }) == "InvalidLengthOfData");
    EXPECT(variant_of([&] { SeparableProblemBuilder::new_(m).observations({1, 2, 3, 4}).weights({1, 2}).build(); }) ==
           "InvalidLengthOfWeights");
    // Student-t quantiles (the `distrs` scalar of src/statistics/mod.rs:285-288) against table values
    EXPECT(std::fabs(student_t_quantile(0.975, 10.0) - 2.2281388519649385) < 1e-10);
    EXPECT(std::fabs(student_t_quantile(0.95, 3.0) - 2.3533634348018264) < 1e-10);
    EXPECT(std::fabs(student_t_quantile(0.995, 995.0) - 2.5807794935342003) < 1e-9);
    EXPECT(std::fabs(student_t_quantile(0.025, 10.0) + 2.2281388519649385) < 1e-10);
    EXPECT(student_t_quantile(0.5, 7.0) == 0.0);
}

static void test_gpu() {
    // double exponential without noise (handrolled-model test of the reference)
    const double tau1 = 1., tau2 = 3., c1 = 4., c2 = 2.5, c3 = 1.;
    std::vector<double> x = linspace_reference(0., 12.5, 1024), y(x.size());
    for (size_t i = 0; i < x.size(); ++i) y[i] = c1 * std::exp(-x[i] / tau1) + c2 * std::exp(-x[i] / tau2) + c3;
    SeparableProblem problem = SeparableProblemBuilder::new_(double_exp_model(x, {2., 6.5})).observations(y).build();
    FitResult fit = LevMarSolver().fit(std::move(problem));
    EXPECT(fit.was_successful());
    const auto &tau = fit.nonlinear_parameters();
    auto c = fit.linear_coefficients();
    EXPECT(c.has_value());
    EXPECT(std::fabs(tau[0] - tau1) < 1e-8 && std::fabs(tau[1] - tau2) < 1e-8);
    EXPECT(std::fabs((*c)[0] - c1) < 1e-8 && std::fabs((*c)[1] - c2) < 1e-8 && std::fabs((*c)[2] - c3) < 1e-8);
    auto bf = fit.best_fit();
    double worst = 0;
    for (size_t i = 0; i < y.size(); ++i) worst = std::fmax(worst, std::fabs((*bf)[i] - y[i]));
    EXPECT(worst < 1e-5);
    std::printf("double-exp fit: tau = (%.12f, %.12f), %d evaluations, termination %d\n", tau[0], tau[1],
                fit.minimization_report.number_of_evaluations, fit.minimization_report.termination);

    // Jacobian vs central differences at the truth (where Kaufman's approximation is exact)
    {
        std::vector<double> t(11), yy(11);
        for (int i = 0; i < 11; ++i) {
            t[i] = i;
            yy[i] = 2 * std::exp(-t[i] / 2) + std::exp(-t[i] / 4) + 1;
        }
        SeparableProblem p = SeparableProblemBuilder::new_(double_exp_model(t, {2., 4.})).observations(yy).build();
        auto J = p.jacobian();
        EXPECT(J.has_value());
        const double h = 1e-6;
        for (int k = 0; k < 2; ++k) {
            std::vector<double> ap = {2., 4.}, am = {2., 4.};
            ap[k] += h;
            am[k] -= h;
            p.set_params(ap);
            auto rp = *p.residuals();
            p.set_params(am);
            auto rm = *p.residuals();
            for (int i = 0; i < 11; ++i) EXPECT(std::fabs((rp[i] - rm[i]) / (2 * h) - (*J)[k * 11 + i]) < 1e-4);
        }
    }
    // MRHS, S = 2
    {
        std::vector<double> xs = linspace_reference(0., 12.5, 20), Y(40);
        const double a[3] = {2., 4., 0.2}, b[3] = {5., 1., 9.};
        for (int i = 0; i < 20; ++i) {
            Y[i] = a[0] * std::exp(-xs[i] / 1.) + a[1] * std::exp(-xs[i] / 3.) + a[2];
            Y[20 + i] = b[0] * std::exp(-xs[i] / 1.) + b[1] * std::exp(-xs[i] / 3.) + b[2];
        }
        SeparableProblem p = SeparableProblemBuilder::mrhs(double_exp_model(xs, {2.5, 6.5})).observations(Y, 2).build();
        FitResult f2 = LevMarSolver().fit(std::move(p));
        const auto &tt = f2.nonlinear_parameters();
        const int i1 = tt[0] < tt[1] ? 0 : 1, i2 = 1 - i1;
        auto C = *f2.linear_coefficients();
        EXPECT(std::fabs(tt[i1] - 1.) < 1e-8 && std::fabs(tt[i2] - 3.) < 1e-8);
        EXPECT(std::fabs(C[i1] - a[0]) < 1e-8 && std::fabs(C[i2] - a[1]) < 1e-8 && std::fabs(C[2] - a[2]) < 1e-8);
        EXPECT(std::fabs(C[3 + i1] - b[0]) < 1e-8 && std::fabs(C[3 + i2] - b[1]) < 1e-8 && std::fabs(C[5] - b[2]) < 1e-8);
    }
}

// a model the descriptor language cannot express (Gaussian + Lorentzian + offset; n = 3, q = 4, four derivative columns)
static ClosureModel peaks_model(const std::vector<double> &x) {
    auto gauss = [](const std::vector<double> &x, const std::vector<double> &p) {
        std::vector<double> v(x.size());
        for (size_t i = 0; i < x.size(); ++i) v[i] = std::exp(-0.5 * (x[i] - p[0]) * (x[i] - p[0]) / (p[1] * p[1]));
        return v;
    };
    auto lorentz = [](const std::vector<double> &x, const std::vector<double> &p) {
        std::vector<double> v(x.size());
        for (size_t i = 0; i < x.size(); ++i) v[i] = p[1] * p[1] / ((x[i] - p[0]) * (x[i] - p[0]) + p[1] * p[1]);
        return v;
    };
    ClosureModel cm({"mu1", "s1", "mu2", "g2"}, x);
    cm.function({"mu1", "s1"}, gauss)
        .partial_deriv("mu1", [gauss](const std::vector<double> &x, const std::vector<double> &p) {
            auto v = gauss(x, p);
            for (size_t i = 0; i < x.size(); ++i) v[i] *= (x[i] - p[0]) / (p[1] * p[1]);
            return v;
        })
        .partial_deriv("s1", [gauss](const std::vector<double> &x, const std::vector<double> &p) {
            auto v = gauss(x, p);
            for (size_t i = 0; i < x.size(); ++i) v[i] *= (x[i] - p[0]) * (x[i] - p[0]) / (p[1] * p[1] * p[1]);
            return v;
        });
    cm.function({"mu2", "g2"}, lorentz)
        .partial_deriv("mu2", [](const std::vector<double> &x, const std::vector<double> &p) {
            std::vector<double> v(x.size());
            for (size_t i = 0; i < x.size(); ++i) {
                const double d = x[i] - p[0], den = d * d + p[1] * p[1];
                v[i] = 2 * p[1] * p[1] * d / (den * den);
            }
            return v;
        })
        .partial_deriv("g2", [](const std::vector<double> &x, const std::vector<double> &p) {
            std::vector<double> v(x.size());
            for (size_t i = 0; i < x.size(); ++i) {
                const double d = x[i] - p[0], den = d * d + p[1] * p[1];
                v[i] = 2 * p[1] * d * d / (den * den);
            }
            return v;
        });
    cm.invariant_function([](const std::vector<double> &x, const std::vector<double> &) { return std::vector<double>(x.size(), 1.0); });
    return cm;
}

static void test_closure_errors() {
    std::vector<double> x = {0, 1, 2, 3};
    auto id = [](const std::vector<double> &x, const std::vector<double> &) { return x; };
    EXPECT(variant_of([&] { ClosureModel({}, x); }) == "EmptyParameters");
    EXPECT(variant_of([&] { ClosureModel({"a", "a"}, x); }) == "DuplicateParameterNames");
    EXPECT(variant_of([&] { ClosureModel({"a"}, x).validate(); }) == "EmptyModel");
    EXPECT(variant_of([&] { ClosureModel({"a"}, x).function({"b"}, id); }) == "FunctionParameterNotInModel");
    EXPECT(variant_of([&] { ClosureModel({"a"}, x).function({"a"}, id).validate(); }) == "MissingDerivative");
    EXPECT(variant_of([&] { ClosureModel({"a", "b"}, x).function({"a"}, id).partial_deriv("a", id).validate(); }) == "UnusedParameter");
    EXPECT(variant_of([&] { ClosureModel({"a", "b"}, x).function({"a"}, id).partial_deriv("b", id); }) == "InvalidDerivative");
    EXPECT(variant_of([&] { ClosureModel({"a"}, x).function({"a"}, id).partial_deriv("a", id).partial_deriv("a", id); }) == "DuplicateDerivative");
    ClosureModel cm = peaks_model(x);
    EXPECT(cm.parameter_count() == 4 && cm.base_function_count() == 3 && cm.pairs().size() == 4);
    // == ModelError::UnexpectedFunctionOutput (src/model/model_basis_function.rs:70): a closure that returns a column of
    // the wrong length is an error, never a write outside its column (too long) or a silently zero tail (too short)
    auto too_long = [](const std::vector<double> &x, const std::vector<double> &) { return std::vector<double>(x.size() + 3, 1.0); };
    auto too_short = [](const std::vector<double> &x, const std::vector<double> &) { return std::vector<double>(x.size() - 1, 1.0); };
    auto model_error_of = [](const std::function<void()> &f) -> std::string {
        try {
            f();
        } catch (const ModelError &e) {
            return e.variant + ":" + std::to_string(e.expected_length) + ":" + std::to_string(e.actual_length);
        }
        return "";
    };
    {
        ClosureModel bad({"a"}, x);
        bad.function({"a"}, too_long).partial_deriv("a", id);
        EXPECT(model_error_of([&] { bad.eval_batch({1.0, 2.0}, 2); }) == "UnexpectedFunctionOutput:4:7");
        ClosureModel bad2({"a"}, x);
        bad2.function({"a"}, id).partial_deriv("a", too_short);
        EXPECT(model_error_of([&] { bad2.eval_batch({1.0}, 1); }) == "");
        EXPECT(model_error_of([&] { bad2.derivs_batch({1.0}, 1); }) == "UnexpectedDerivativeOutput:4:3");
    }
}

// LevMarSolver::fit for a BATCH of closure models: LM drivers on the device, the model on the host
static void test_closure_gpu() {
    const int64_t B = 6, m = 400;
    std::vector<double> x((size_t)m);
    for (int64_t i = 0; i < m; ++i) x[(size_t)i] = 10.0 * (double)i / (double)(m - 1);
    ClosureModel cm = peaks_model(x);
    std::vector<double> truth((size_t)(B * 4)), guess((size_t)(B * 4)), Y((size_t)(B * m));
    for (int64_t b = 0; b < B; ++b) {
        const double t[4] = {2.6 + 0.15 * b, 0.5 + 0.05 * b, 6.1 + 0.1 * b, 0.6 + 0.08 * b};
        const double f[4] = {1.04, 0.93, 0.97, 1.08};
        for (int k = 0; k < 4; ++k) truth[(size_t)(b * 4 + k)] = t[k], guess[(size_t)(b * 4 + k)] = t[k] * f[k];
    }
    const std::vector<double> Phi = cm.eval_batch(truth, B);
    for (int64_t b = 0; b < B; ++b)
        for (int64_t i = 0; i < m; ++i)
            Y[(size_t)(b * m + i)] = (10.0 + b) * Phi[(size_t)((b * 3 + 0) * m + i)] + (20.0 - b) * Phi[(size_t)((b * 3 + 1) * m + i)] + 1.5;
    ExternalBatchProblem prob(cm, Y, B);
    // the trait-level evaluation at the truth: zero residual, coefficients recovered
    auto ev = prob.evaluate(truth);
    for (int64_t b = 0; b < B; ++b) {
        EXPECT(ev.status[(size_t)b] == 0 && ev.cost[(size_t)b] < 1e-20);
        EXPECT(std::fabs(ev.coefficients[(size_t)(b * 3)] - (10.0 + b)) < 1e-9 && std::fabs(ev.coefficients[(size_t)(b * 3 + 2)] - 1.5) < 1e-9);
    }
    // Jacobian against central differences of the residual at the truth, where Kaufman's approximation is exact (the
    // reference's own check, src/solvers/levmar/test.rs:21-40)
    {
        auto e0 = prob.evaluate(truth);
        const double h = 1e-6;
        for (int k = 0; k < 4; ++k) {
            std::vector<double> ap = truth, am = truth;
            for (int64_t b = 0; b < B; ++b) ap[(size_t)(b * 4 + k)] += h, am[(size_t)(b * 4 + k)] -= h;
            auto rp = prob.evaluate(ap).residuals, rm = prob.evaluate(am).residuals;
            double worst = 0, scale = 0;
            for (int64_t b = 0; b < B; ++b)
                for (int64_t i = 0; i < m; ++i) {
                    const double fd = (rp[(size_t)(b * m + i)] - rm[(size_t)(b * m + i)]) / (2 * h);
                    worst = std::fmax(worst, std::fabs(fd - e0.jacobian[(size_t)((b * 4 + k) * m + i)]));
                    scale = std::fmax(scale, std::fabs(fd));
                }
            EXPECT(worst <= 1e-5 * scale);
        }
    }
    for (int lazy = 0; lazy < 2; ++lazy) {
        auto fit = prob.fit(guess, LevenbergMarquardt(), lazy != 0);
        for (int64_t b = 0; b < B; ++b) {
            EXPECT(fit.reports[(size_t)b].termination > 0);
            for (int k = 0; k < 4; ++k) EXPECT(std::fabs(fit.nonlinear_parameters[(size_t)(b * 4 + k)] - truth[(size_t)(b * 4 + k)]) < 1e-6);
            EXPECT(std::fabs(fit.linear_coefficients[(size_t)(b * 3 + 1)] - (20.0 - b)) < 1e-5);
        }
        std::printf("closure-model batch fit (%s): %d steps, problem 0: %d evaluations, objective %.3e\n",
                    lazy ? "derivatives on accept" : "eager", fit.steps, fit.reports[0].n_evals, fit.reports[0].objective);
    }
    // S = 2 right-hand sides per problem sharing the nonlinear parameters (SeparableProblemBuilder::mrhs,
    // src/problem/builder.rs:194-225; round 6: the batched fit of caller-evaluated models takes them): exact data -> the
    // global fit recovers the parameters, and each column its own coefficients
    {
        const int64_t S = 2;
        std::vector<double> Y2((size_t)(B * S * m));
        for (int64_t b = 0; b < B; ++b)
            for (int64_t sidx = 0; sidx < S; ++sidx)
                for (int64_t i = 0; i < m; ++i)
                    Y2[(size_t)((b * S + sidx) * m + i)] = (10.0 + b + 3.0 * sidx) * Phi[(size_t)((b * 3 + 0) * m + i)] +
                                                          (20.0 - b - 2.0 * sidx) * Phi[(size_t)((b * 3 + 1) * m + i)] + 1.5 + sidx;
        ExternalBatchProblem prob2(cm, Y2, B, nullptr, -1.0, 0, S);
        auto fit = prob2.fit(guess);
        for (int64_t b = 0; b < B; ++b) {
            EXPECT(fit.reports[(size_t)b].termination > 0);
            for (int k = 0; k < 4; ++k) EXPECT(std::fabs(fit.nonlinear_parameters[(size_t)(b * 4 + k)] - truth[(size_t)(b * 4 + k)]) < 1e-6);
            for (int64_t sidx = 0; sidx < S; ++sidx) {
                EXPECT(std::fabs(fit.linear_coefficients[(size_t)((b * S + sidx) * 3 + 0)] - (10.0 + b + 3.0 * sidx)) < 1e-5);
                EXPECT(std::fabs(fit.linear_coefficients[(size_t)((b * S + sidx) * 3 + 2)] - (1.5 + sidx)) < 1e-5);
            }
        }
        std::printf("closure-model GLOBAL fit (S = 2): %d steps, problem 0: %d evaluations, objective %.3e\n", fit.steps,
                    fit.reports[0].n_evals, fit.reports[0].objective);
    }
}

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "errors";
    test_errors();
    test_closure_errors();
    if (mode == "gpu") {
        if (vp_device_count() <= 0) {
            std::printf("no GPU visible\n");
            return 2;
        }
        test_gpu();
        test_closure_gpu();
    } else {
        // without a device the product must refuse to compute (no CPU fallback)
        if (vp_device_count() <= 0) {
            bool threw = false;
            try {
                std::vector<double> x = {0, 1, 2, 3};
                SeparableProblemBuilder::new_(double_exp_model(x, {1.0, 2.0})).observations({1, 2, 3, 4}).build();
            } catch (const HipError &e) {
                threw = e.code == VP_ERR_NO_DEVICE;
            }
            EXPECT(threw);
        }
    }
    std::printf("%s: %d failure(s)\n", mode.c_str(), failures);
    return failures ? 1 : 0;
}
