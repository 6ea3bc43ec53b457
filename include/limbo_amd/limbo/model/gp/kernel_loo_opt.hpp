// limbo/model/gp/kernel_loo_opt.hpp — maximise the leave-one-out CV log probability over the kernel
// hyper-parameters (contract: src/limbo/model/gp/kernel_loo_opt.hpp:55-97).
// One persistent device clone per calling host thread instead of the reference's deep copy per evaluation
// (kernel_loo_opt.hpp:79).
// Interface attribution: the names of this header (the policy and its nested objective type) are those of resibots/limbo
// (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info), file named above — a drop-in must keep them.  What they
// do is written once, in this project's own terms: limbo_amd::fit (hp_opt.hpp).
#ifndef LIMBO_MODEL_GP_KERNEL_LOO_OPT_HPP
#define LIMBO_MODEL_GP_KERNEL_LOO_OPT_HPP
#include <limbo/model/gp/hp_opt.hpp>
namespace limbo {
    namespace model {
        namespace gp {
            template <typename Params, typename Optimizer = opt::Rprop<Params>>
            struct KernelLooOpt : public HPOpt<Params, Optimizer> {
                template <typename GP>
                void operator()(GP& gp)
                {
                    this->_called = true;
                    limbo_amd::fit::run<Optimizer, limbo_amd::fit::KernelParams, limbo_amd::fit::LogLooCv, KernelLooOptimization<GP>>(gp);
                }

            protected:
                template <typename GP>
                using KernelLooOptimization = limbo_amd::fit::Objective<Params, GP, limbo_amd::fit::KernelParams, limbo_amd::fit::LogLooCv>;
            };
        } // namespace gp
    } // namespace model
} // namespace limbo
#endif
