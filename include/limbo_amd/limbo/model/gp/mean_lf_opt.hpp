// limbo/model/gp/mean_lf_opt.hpp — maximise the log marginal likelihood over the mean function's
// hyper-parameters (contract: src/limbo/model/gp/mean_lf_opt.hpp:55-98): the factor stays, obs_mean and alpha move.
// Interface attribution: the names of this header (the policy and its nested objective type) are those of resibots/limbo
// (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info), file named above — a drop-in must keep them.  What they
// do is written once, in this project's own terms: limbo_amd::fit (hp_opt.hpp).
#ifndef LIMBO_MODEL_GP_MEAN_LF_OPT_HPP
#define LIMBO_MODEL_GP_MEAN_LF_OPT_HPP
#include <limbo/model/gp/hp_opt.hpp>
namespace limbo {
    namespace model {
        namespace gp {
            template <typename Params, typename Optimizer = opt::Rprop<Params>>
            struct MeanLFOpt : public HPOpt<Params, Optimizer> {
                template <typename GP>
                void operator()(GP& gp)
                {
                    this->_called = true;
                    limbo_amd::fit::run<Optimizer, limbo_amd::fit::MeanParams, limbo_amd::fit::LogLik, MeanLFOptimization<GP>>(gp);
                }

            protected:
                template <typename GP>
                using MeanLFOptimization = limbo_amd::fit::Objective<Params, GP, limbo_amd::fit::MeanParams, limbo_amd::fit::LogLik>;
            };
        } // namespace gp
    } // namespace model
} // namespace limbo
#endif
