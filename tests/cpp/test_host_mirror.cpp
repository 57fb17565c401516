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

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "errors";
    test_errors();
    if (mode == "gpu") {
        if (vp_device_count() <= 0) {
            std::printf("no GPU visible\n");
            return 2;
        }
        test_gpu();
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
