// limbo/model/gp/kernel_mean_lf_opt.hpp — maximise the log marginal likelihood over the kernel's AND the mean
// function's hyper-parameters, the kernel's first in the parameter vector (contract:
// src/limbo/model/gp/kernel_mean_lf_opt.hpp:55-118).
// Interface attribution: the names of this header (the policy and its nested objective type) are those of resibots/limbo
// (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info), file named above — a drop-in must keep them.  What they
// do is written once, in this project's own terms: limbo_amd::fit (hp_opt.hpp).
#ifndef LIMBO_MODEL_GP_KERNEL_MEAN_LF_OPT_HPP
#define LIMBO_MODEL_GP_KERNEL_MEAN_LF_OPT_HPP
#include <limbo/model/gp/hp_opt.hpp>
namespace limbo {
    namespace model {
        namespace gp {
            template <typename Params, typename Optimizer = opt::Rprop<Params>>
            struct KernelMeanLFOpt : public HPOpt<Params, Optimizer> {
                template <typename GP>
                void operator()(GP& gp)
                {
                    this->_called = true;
                    limbo_amd::fit::run<Optimizer, limbo_amd::fit::KernelAndMeanParams, limbo_amd::fit::LogLik, KernelMeanLFOptimization<GP>>(gp);
                }

            protected:
                template <typename GP>
                using KernelMeanLFOptimization = limbo_amd::fit::Objective<Params, GP, limbo_amd::fit::KernelAndMeanParams, limbo_amd::fit::LogLik>;
            };
        } // namespace gp
    } // namespace model
} // namespace limbo
#endif
