// limbo/model/multi_gp/parallel_lf_opt.hpp — fit the hyper-parameters of every output GP of a
// MultiGP, all at once (contract: src/limbo/model/multi_gp/parallel_lf_opt.hpp:56-70).  Each output
// GP is an independent device GP on its own stream: the fits overlap on the MI355X.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_MODEL_MULTI_GP_PARALLEL_LF_OPT_HPP
#define LIMBO_MODEL_MULTI_GP_PARALLEL_LF_OPT_HPP
#include <type_traits>
#include <utility>
#include <limbo/model/gp/hp_opt.hpp>
#include <limbo/tools/parallel.hpp>
namespace limbo {
    namespace model {
        namespace multi_gp {
            /// does the per-output policy know how to fit several GPs in lock-step (gp::KernelLFOpt::fit_many)?
            template <typename H, typename GPs, typename = void>
            struct has_fit_many : std::false_type {};
            template <typename H, typename GPs>
            struct has_fit_many<H, GPs, decltype((void)H::fit_many(std::declval<GPs&>()))> : std::true_type {};

            template <typename Params, typename HyperParamsOptimizer = limbo::model::gp::NoLFOpt<Params>>
            struct ParallelLFOpt : public limbo::model::gp::HPOpt<Params> {
                template <typename GP>
                void operator()(GP& gp)
                {
                    this->_called = true;
                    auto& gps = gp.gp_models();
                    if constexpr (has_fit_many<HyperParamsOptimizer, decltype(gps)>::value) {
                        // every output's Rprop iteration i as ONE batched device evaluation (kernel_lf_opt.hpp: fit_many)
                        HyperParamsOptimizer hp_optimize; // (constructed so that a policy that was meant to run does not warn)
                        if (HyperParamsOptimizer::fit_many(gps)) {
                            hp_optimize.mark_called();
                            return;
                        }
                        hp_optimize.mark_called();
                    }
                    limbo::tools::par::loop(0, gps.size(), [&](size_t i) {
                        HyperParamsOptimizer hp_optimize;
                        hp_optimize(gps[i]);
                    });
                }
            };
        } // namespace multi_gp
    } // namespace model
} // namespace limbo
#endif
