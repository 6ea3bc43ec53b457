// limbo/opt/rprop.hpp — resilient back-propagation, the default hyper-parameter optimiser
// (contract and constants: src/limbo/opt/rprop.hpp:82-145: delta0 0.1, delta in [1e-6, 50],
// eta- 0.5, eta+ 1.2; maximises f; returns the best point SEEN, not the last).
// Host code by design: it drives <= dim(theta) scalars; every f(theta) is one device evaluation.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_AMD_OPT_RPROP_HPP
#define LIMBO_AMD_OPT_RPROP_HPP
// With limbo's own tree on the include path BEHIND this directory, this file steps aside: limbo's
// <limbo/opt/rprop.hpp> is the one that gets compiled, so every other limbo header keeps seeing exactly what it was
// written against.  Stand-alone (no limbo tree), the definitions below provide the same names.
#if defined(__has_include_next)
#if __has_include_next(<limbo/opt/rprop.hpp>)
#define LIMBO_AMD_OPT_RPROP_HPP_FORWARDED 1
#include_next <limbo/opt/rprop.hpp>
#endif
#endif
#ifndef LIMBO_AMD_OPT_RPROP_HPP_FORWARDED
#ifndef LIMBO_OPT_RPROP_HPP
#define LIMBO_OPT_RPROP_HPP
#include <algorithm>
#include <cmath>
#include <limits>
#include <limbo/opt/optimizer.hpp>
#include <limbo/tools/macros.hpp>
#include <limbo/tools/math.hpp>
namespace limbo {
    namespace defaults {
        struct opt_rprop {
            BO_PARAM(int, iterations, 300);
            BO_PARAM(double, eps_stop, 0.0);
        };
    } // namespace defaults
    namespace opt {
        template <typename Params>
        struct Rprop {
            template <typename F>
            Eigen::VectorXd operator()(const F& f, const Eigen::VectorXd& init, bool bounded) const
            {
                assert(Params::opt_rprop::eps_stop() >= 0.);
                const int dim = (int)init.size();
                const double delta0 = 0.1, delta_min = 1e-6, delta_max = 50, eta_minus = 0.5, eta_plus = 1.2;
                const double eps_stop = Params::opt_rprop::eps_stop();
                auto clamp01 = [&](double v) { return bounded ? std::min(1.0, std::max(0.0, v)) : v; };

                Eigen::VectorXd params = init;
                for (int j = 0; j < dim; ++j)
                    params(j) = clamp01(params(j));
                Eigen::VectorXd step = Eigen::VectorXd::Constant(dim, delta0);
                Eigen::VectorXd prev = Eigen::VectorXd::Zero(dim); // previous descent direction (zeroed where the sign flipped)
                Eigen::VectorXd best_params = params;
                double best = -std::numeric_limits<double>::infinity();

                for (int it = 0; it < Params::opt_rprop::iterations(); ++it) {
                    const eval_t perf = eval_grad(f, params);
                    if (fun(perf) > best) {
                        best = fun(perf);
                        best_params = params;
                    }
                    Eigen::VectorXd g = -grad(perf); // descent direction of -f
                    double gnorm2 = 0.0;
                    for (int j = 0; j < dim; ++j) {
                        const double s = prev(j) * g(j);
                        if (s > 0)
                            step(j) = std::min(step(j) * eta_plus, delta_max);
                        else if (s < 0) {
                            step(j) = std::max(step(j) * eta_minus, delta_min);
                            g(j) = 0;
                        }
                        params(j) = clamp01(params(j) - tools::signum(g(j)) * step(j));
                        gnorm2 += g(j) * g(j);
                    }
                    prev = g;
                    if (std::sqrt(gnorm2) < eps_stop)
                        break;
                }
                return best_params;
            }
        };
    } // namespace opt
} // namespace limbo
#endif // LIMBO_OPT_RPROP_HPP
#endif // stand-alone
#endif
