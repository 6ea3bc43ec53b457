// test_mixed_tree.cpp — the include-path switch of INTEGRATION.md exercised against limbo's OWN tree:
// include/limbo_amd comes first (model, kernels, means, acquisition functions, Rprop ... are the drop-in),
// /root/reference/src second — every header the drop-in does not shadow (opt::GridSearch, opt::RandomPoint,
// tools::random_generator / math / sys, external/rand_utils) is limbo's unmodified file.  A hand-written BO loop
// in the shape of bayes_opt/boptimizer.hpp:128-166 (model.compute -> acquisition optimiser over
// acqui(x, afun, g) -> eval -> model.add_sample) then runs limbo's inner optimiser over the device model.
// (bo_base.hpp itself needs boost::parameter / fusion / filesystem, which are not in this image.)
// Built only where /root/reference exists; exit code = failed checks.
#include <cstdio>

#include <limbo/acqui/ei.hpp>
#include <limbo/acqui/ucb.hpp>
#include <limbo/model/gp.hpp>
#include <limbo/model/gp/kernel_lf_opt.hpp>
#include <limbo/opt/batch_search.hpp>
#include <limbo/opt/parallel_repeater.hpp>
// --- from here on: files that only exist in the reference tree ---
#include <limbo/opt/grid_search.hpp>
#include <limbo/opt/random_point.hpp>
#include <limbo/tools/math.hpp>
#include <limbo/tools/random_generator.hpp>
#include <limbo/tools/sys.hpp>

using namespace limbo;
using Eigen::VectorXd;

struct Params {
    struct kernel : public defaults::kernel {
        BO_PARAM(double, noise, 1e-6);
    };
    struct kernel_squared_exp_ard : public defaults::kernel_squared_exp_ard {};
    struct kernel_maternfivehalves : public defaults::kernel_maternfivehalves {};
    struct opt_rprop : public defaults::opt_rprop {
        BO_PARAM(int, iterations, 20);
    };
    struct opt_parallelrepeater : public defaults::opt_parallelrepeater {
        BO_PARAM(int, repeats, 2);
    };
    struct opt_gridsearch : public defaults::opt_gridsearch {
        BO_PARAM(int, bins, 6);
    };
    struct opt_batchrandomsearch : public defaults::opt_batchrandomsearch {};
    struct acqui_ucb : public defaults::acqui_ucb {};
    struct acqui_ei : public defaults::acqui_ei {};
};

static int g_failed = 0;
#define CHECK(c)                                                      \
    do {                                                              \
        if (!(c)) {                                                   \
            ++g_failed;                                               \
            std::printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
        }                                                             \
    } while (0)

// maximum 0 at (0.3, 0.6)
static double objective(const VectorXd& x) { return -((x(0) - 0.3) * (x(0) - 0.3) + (x(1) - 0.6) * (x(1) - 0.6)); }

template <typename Acqui, typename InnerOpt>
static double bo_loop(int iterations, const InnerOpt& inner_opt, bool hp_opt)
{
    using GP_t = model::GP<Params, kernel::SquaredExpARD<Params>, mean::Data<Params>, model::gp::KernelLFOpt<Params>>;
    auto first = [](const VectorXd& v) { return v(0); };
    GP_t gp(2, 1);
    std::vector<VectorXd> X, Y;
    for (int i = 0; i < 6; ++i) { // init::RandomSampling's body (init/random_sampling.hpp): random points in [0,1]^d
        X.push_back(tools::random_vector(2, true)); // limbo's tools/random_generator.hpp
        Y.push_back(tools::make_vector(objective(X.back()))); // limbo's tools/math.hpp
    }
    gp.compute(X, Y); // boptimizer.hpp:137
    double best = -1e300;
    for (const auto& y : Y)
        best = std::max(best, y(0));
    for (int it = 0; it < iterations; ++it) {
        Acqui acqui(gp, it); // :149
        auto acqui_optimization = [&](const VectorXd& x, bool g) { return acqui(x, first, g); }; // :151-152
        VectorXd starting_point = tools::random_vector(2, true); // :153
        VectorXd new_sample = inner_opt(acqui_optimization, starting_point, true); // :154
        CHECK(!tools::is_nan_or_inf(new_sample));
        VectorXd y = tools::make_vector(objective(new_sample));
        gp.add_sample(new_sample, y); // bo_base.hpp:240 via eval_and_add
        if (hp_opt && (it + 1) % 5 == 0)
            gp.optimize_hyperparams(); // boptimizer.hpp:162-163
        best = std::max(best, y(0));
    }
    return best;
}

int main()
{
    std::printf("host %s pid %s (limbo's tools/sys.hpp)\n", tools::hostname().c_str(), tools::getpid().c_str());
    // limbo's own GridSearch over the device model, UCB and EI
    const double b1 = bo_loop<acqui::UCB<Params, model::GP<Params, kernel::SquaredExpARD<Params>, mean::Data<Params>, model::gp::KernelLFOpt<Params>>>>(
        12, opt::GridSearch<Params>(), true);
    std::printf("GridSearch + UCB : best %.5f\n", b1);
    CHECK(b1 > -0.02);
    const double b2 = bo_loop<acqui::EI<Params, model::GP<Params, kernel::SquaredExpARD<Params>, mean::Data<Params>, model::gp::KernelLFOpt<Params>>>>(
        12, opt::GridSearch<Params>(), false);
    std::printf("GridSearch + EI  : best %.5f\n", b2);
    CHECK(b2 > -0.02);
    const double b3 = bo_loop<acqui::UCB<Params, model::GP<Params, kernel::SquaredExpARD<Params>, mean::Data<Params>, model::gp::KernelLFOpt<Params>>>>(
        12, opt::RandomPoint<Params>(), false);
    std::printf("RandomPoint      : best %.5f (control)\n", b3);
    // limbo's GridSearch and the batched one agree on the device model (same grid, same tie-breaking)
    {
        using GP_t = model::GP<Params, kernel::SquaredExpARD<Params>, mean::Data<Params>>;
        std::vector<VectorXd> X, Y;
        for (int i = 0; i < 40; ++i) {
            X.push_back(tools::random_vector(2, true));
            Y.push_back(tools::make_vector(objective(X.back())));
        }
        GP_t gp;
        gp.compute(X, Y);
        auto first = [](const VectorXd& v) { return v(0); };
        acqui::EI<Params, GP_t> ei(gp);
        auto plain = [&](const VectorXd& x, bool g) { return ei(x, first, g); };
        auto obj = opt::make_batch_objective(ei, first); // EI caches f_max: needs the non-const objective
        const VectorXd init = VectorXd::Constant(2, 0.5);
        const VectorXd a = opt::GridSearch<Params>()(plain, init, true);
        const VectorXd b = opt::BatchGridSearch<Params>()(obj, init, true);
        CHECK((a - b).norm() == 0.0);
    }
    std::printf("%d failed checks\n", g_failed);
    return g_failed;
}
