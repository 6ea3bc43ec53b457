// limbo/opt/parallel_repeater.hpp — run the same local optimiser from `repeats` perturbed starts
// and keep the best (contract: src/limbo/opt/parallel_repeater.hpp:77-107, which uses TBB
// tools::par::max).  Here every restart is a host thread; objective functors that own a device
// GP (model/gp/kernel_lf_opt.hpp) clone it once per thread, so the restarts are independent GPs
// on independent HIP streams and their kernels interleave on the MI355X — the latency-bound
// factorisation of one restart leaves most CUs idle for the others.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_OPT_PARALLEL_REPEATER_HPP
#define LIMBO_OPT_PARALLEL_REPEATER_HPP
#include <future>
#include <limits>
#include <random>
#include <vector>
#include <limbo/opt/batched_rprop.hpp>
#include <limbo/opt/optimizer.hpp>
#include <limbo/tools/macros.hpp>
namespace limbo {
    namespace defaults {
        struct opt_parallelrepeater {
            BO_PARAM(int, repeats, 10);
            BO_PARAM(double, epsilon, 1e-2);
        };
    } // namespace defaults
    namespace opt {
        template <typename Params, typename Optimizer>
        struct ParallelRepeater {
            template <typename F>
            Eigen::VectorXd operator()(const F& f, const Eigen::VectorXd& init, bool bounded) const
            {
                const int repeats = Params::opt_parallelrepeater::repeats();
                const double eps = Params::opt_parallelrepeater::epsilon();
                assert(repeats > 0 && eps > 0.);
                std::random_device rd;
                std::vector<unsigned> seeds(repeats);
                for (auto& s : seeds)
                    s = rd();
                if constexpr (has_eval_batch<F>::value && is_rprop<Optimizer>::value) {
                    // The objective can evaluate many points at once (a device GP per restart: kernel_lf_opt.hpp) and the
                    // local optimiser is Rprop: the restarts advance in lock-step, iteration i of all of them being ONE
                    // batched device evaluation (gpe_batch_hp_objective) instead of `repeats` launch chains racing each
                    // other.  Same perturbed starts, same Rprop iterates per restart, same arg-max.
                    if (batch_restarts_enabled()) {
                        std::vector<Eigen::VectorXd> starts;
                        for (int i = 0; i < repeats; ++i) {
                            std::mt19937_64 g(seeds[i]);
                            std::uniform_real_distribution<double> u(-eps, eps);
                            Eigen::VectorXd start = init;
                            for (int j = 0; j < (int)start.size(); ++j)
                                start(j) += u(g);
                            starts.push_back(start);
                        }
                        // (Optimizer = Rprop<P>: P's opt_rprop settings, which need not be this repeater's Params — ADVICE r3)
                        auto res = rprop_lockstep<typename is_rprop<Optimizer>::params>([&](const std::vector<Eigen::VectorXd>& xs, bool gr) { return f.eval_batch(xs, gr); }, starts, bounded);
                        Eigen::VectorXd best = init;
                        double best_val = -std::numeric_limits<float>::max();
                        for (auto& r : res)
                            if (r.second > best_val) { // (the value at the returned point: what opt::eval(f, v) would recompute)
                                best_val = r.second;
                                best = r.first;
                            }
                        return best;
                    }
                }
                auto body = [&](int i) {
                    std::mt19937_64 g(seeds[i]);
                    std::uniform_real_distribution<double> u(-eps, eps);
                    Eigen::VectorXd start = init;
                    for (int j = 0; j < (int)start.size(); ++j)
                        start(j) += u(g);
                    Eigen::VectorXd v = Optimizer()(f, start, bounded);
                    return std::make_pair(v, opt::eval(f, v));
                };
                std::vector<std::future<std::pair<Eigen::VectorXd, double>>> jobs;
                for (int i = 0; i < repeats; ++i)
                    jobs.push_back(std::async(std::launch::async, body, i));
                Eigen::VectorXd best = init;
                double best_val = -std::numeric_limits<float>::max();
                for (auto& j : jobs) {
                    auto r = j.get();
                    if (r.second > best_val) {
                        best_val = r.second;
                        best = r.first;
                    }
                }
                return best;
            }
        };
    } // namespace opt
} // namespace limbo
#endif
