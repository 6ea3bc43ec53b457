// limbo/opt/batched_rprop.hpp — Rprop for G independent problems advanced in LOCK-STEP (an addition; not in limbo).
//
// limbo runs hyper-parameter restarts (opt/parallel_repeater.hpp:84-105) and per-output fits
// (model/multi_gp/parallel_lf_opt.hpp:64-67) as independent TBB tasks, each a sequential Rprop whose every objective
// evaluation factors one kernel matrix.  On the device G such evaluations are one launch sequence
// (gpe_batch_hp_objective, gridDim.z = member) — if the G optimisers ask for them at the same time.  This is Rprop with
// limbo's constants and update rule (src/limbo/opt/rprop.hpp:82-145: delta0 0.1, delta in [1e-6, 50], eta- 0.5,
// eta+ 1.2, maximises, returns the best point SEEN) for G members at once: iteration i of every member is evaluated by ONE
// call fb(xs, true).  Member g's iterates are exactly those of opt::Rprop started from inits[g].
#ifndef LIMBO_AMD_OPT_BATCHED_RPROP_HPP
#define LIMBO_AMD_OPT_BATCHED_RPROP_HPP
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <type_traits>
#include <utility>
#include <vector>
#include <limbo/opt/rprop.hpp>
#include <limbo/tools/math.hpp>
namespace limbo {
    namespace opt {
        /// is Optimizer limbo's Rprop (this tree's or limbo's own)?
        template <typename T>
        struct is_rprop : std::false_type {};
        template <typename P>
        struct is_rprop<Rprop<P>> : std::true_type {
            using params = P; // the Params Rprop<P> itself reads its iterations / eps_stop from (rprop.hpp:84-88)
        };

        /// the Params an optimiser reads its Rprop settings from: P for Rprop<P>, `Fallback` for anything else (never used then)
        template <typename T, typename Fallback>
        struct rprop_params_of {
            using type = Fallback;
        };
        template <typename P, typename Fallback>
        struct rprop_params_of<Rprop<P>, Fallback> {
            using type = P;
        };

        /// does an objective functor evaluate many points at once?  eval_batch(xs, grad) -> std::vector<eval_t>
        template <typename F, typename = void>
        struct has_eval_batch : std::false_type {};
        template <typename F>
        struct has_eval_batch<F, decltype((void)std::declval<const F&>().eval_batch(std::declval<const std::vector<Eigen::VectorXd>&>(), true))> : std::true_type {};

        /// fb(xs, true) -> std::vector<eval_t>, xs.size() == inits.size().  Returns (best params, best value) per member.
        template <typename Params, typename BatchF>
        std::vector<std::pair<Eigen::VectorXd, double>> rprop_lockstep(const BatchF& fb, const std::vector<Eigen::VectorXd>& inits, bool bounded)
        {
            const size_t G = inits.size();
            const double delta0 = 0.1, delta_min = 1e-6, delta_max = 50, eta_minus = 0.5, eta_plus = 1.2;
            const double eps_stop = Params::opt_rprop::eps_stop();
            assert(eps_stop >= 0.);
            auto clamp01 = [&](double v) { return bounded ? std::min(1.0, std::max(0.0, v)) : v; };
            std::vector<Eigen::VectorXd> params(inits), step(G), prev(G), best_params(G);
            std::vector<double> best(G, -std::numeric_limits<double>::infinity());
            std::vector<char> live(G, 1);
            for (size_t g = 0; g < G; ++g) {
                const int dim = (int)inits[g].size();
                for (int j = 0; j < dim; ++j)
                    params[g](j) = clamp01(params[g](j));
                step[g] = Eigen::VectorXd::Constant(dim, delta0);
                prev[g] = Eigen::VectorXd::Zero(dim);
                best_params[g] = params[g];
            }
            for (int it = 0; it < Params::opt_rprop::iterations(); ++it) {
                if (std::find(live.begin(), live.end(), 1) == live.end())
                    break;
                const std::vector<eval_t> perf = fb(params, true); // a stopped member is evaluated where it stands (ignored)
                for (size_t g = 0; g < G; ++g) {
                    if (!live[g])
                        continue;
                    const int dim = (int)params[g].size();
                    if (fun(perf[g]) > best[g]) {
                        best[g] = fun(perf[g]);
                        best_params[g] = params[g];
                    }
                    Eigen::VectorXd gr = -grad(perf[g]);
                    double gnorm2 = 0.0;
                    for (int j = 0; j < dim; ++j) {
                        const double s = prev[g](j) * gr(j);
                        if (s > 0)
                            step[g](j) = std::min(step[g](j) * eta_plus, delta_max);
                        else if (s < 0) {
                            step[g](j) = std::max(step[g](j) * eta_minus, delta_min);
                            gr(j) = 0;
                        }
                        params[g](j) = clamp01(params[g](j) - tools::signum(gr(j)) * step[g](j));
                        gnorm2 += gr(j) * gr(j);
                    }
                    prev[g] = gr;
                    if (std::sqrt(gnorm2) < eps_stop)
                        live[g] = 0;
                }
            }
            std::vector<std::pair<Eigen::VectorXd, double>> out;
            for (size_t g = 0; g < G; ++g)
                out.emplace_back(best_params[g], best[g]);
            return out;
        }
        /// LIMBO_AMD_BATCH_RESTARTS=0 switches the lock-step paths off (restarts / outputs as host threads, one launch
        /// chain each: rounds 1-2, and what tests compare against)
        inline bool batch_restarts_enabled()
        {
            const char* e = std::getenv("LIMBO_AMD_BATCH_RESTARTS");
            return !(e && std::atoi(e) == 0);
        }
    } // namespace opt
} // namespace limbo
#endif
