// limbo/model/gp/kernel_loo_opt.hpp — maximise the leave-one-out CV log probability over the kernel
// hyper-parameters (contract: src/limbo/model/gp/kernel_loo_opt.hpp:55-97).
// As in kernel_lf_opt.hpp here: one persistent device clone per calling host thread instead of the
// reference's deep copy per evaluation (kernel_loo_opt.hpp:79).
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_MODEL_GP_KERNEL_LOO_OPT_HPP
#define LIMBO_MODEL_GP_KERNEL_LOO_OPT_HPP
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <limbo/model/gp/hp_opt.hpp>
namespace limbo {
    namespace model {
        namespace gp {
            template <typename Params, typename Optimizer = opt::Rprop<Params>>
            struct KernelLooOpt : public HPOpt<Params, Optimizer> {
            public:
                template <typename GP>
                void operator()(GP& gp)
                {
                    this->_called = true;
                    KernelLooOptimization<GP> optimization(gp);
                    Optimizer optimizer;
                    Eigen::VectorXd params = optimizer(optimization, gp.kernel_function().h_params(), false);
                    gp.kernel_function().set_h_params(params);
                    gp.recompute(false);
                    gp.compute_log_loo_cv();
                }

            protected:
                template <typename GP>
                struct KernelLooOptimization {
                public:
                    KernelLooOptimization(const GP& gp) : _original_gp(gp) {}

                    opt::eval_t operator()(const Eigen::VectorXd& params, bool compute_grad) const
                    {
                        GP& gp = _workers.get(_original_gp);
                        gp.kernel_function().set_h_params(params);
                        gp.recompute(false);
                        const double loo = gp.compute_log_loo_cv();
                        if (!compute_grad)
                            return opt::no_grad(loo);
                        return {loo, opt::eval_t::second_type(gp.compute_kernel_grad_log_loo_cv())};
                    }

                protected:
                    const GP& _original_gp;
                    limbo_amd::WorkerClones<Params, GP> _workers;
                };
            };
        } // namespace gp
    } // namespace model
} // namespace limbo
#endif
