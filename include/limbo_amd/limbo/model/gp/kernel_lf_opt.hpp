// limbo/model/gp/kernel_lf_opt.hpp — maximise the log marginal likelihood over the kernel
// hyper-parameters (contract: src/limbo/model/gp/kernel_lf_opt.hpp:56-97).
//
// The reference's objective deep-copies the whole GP (K, L, K^-1: 3 N^2 doubles, 403 MB at
// N = 4096) for EVERY evaluation (kernel_lf_opt.hpp:79).  Here an evaluation is "same X in HBM,
// new theta": each host thread that calls the objective keeps ONE private device clone of the
// original GP for the lifetime of the optimisation and re-runs build -> factor -> solve on it.
// The original GP is untouched until the optimiser returns, exactly as in the reference.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_MODEL_GP_KERNEL_LF_OPT_HPP
#define LIMBO_MODEL_GP_KERNEL_LF_OPT_HPP
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <limbo/model/gp/hp_opt.hpp>
namespace limbo {
    namespace model {
        namespace gp {
            template <typename Params, typename Optimizer = opt::Rprop<Params>>
            struct KernelLFOpt : public HPOpt<Params, Optimizer> {
            public:
                template <typename GP>
                void operator()(GP& gp)
                {
                    this->_called = true;
                    KernelLFOptimization<GP> optimization(gp);
                    Optimizer optimizer;
                    Eigen::VectorXd params = optimizer(optimization, gp.kernel_function().h_params(), false);
                    gp.kernel_function().set_h_params(params);
                    gp.recompute(false);
                    gp.compute_log_lik();
                }

            protected:
                template <typename GP>
                struct KernelLFOptimization {
                public:
                    KernelLFOptimization(const GP& gp) : _original_gp(gp) {}

                    opt::eval_t operator()(const Eigen::VectorXd& params, bool compute_grad) const
                    {
                        GP& gp = _workers.get(_original_gp);
                        gp.kernel_function().set_h_params(params);
                        gp.recompute(false);
                        const double lik = gp.compute_log_lik();
                        if (!compute_grad)
                            return opt::no_grad(lik);
                        return {lik, opt::eval_t::second_type(gp.compute_kernel_grad_log_lik())};
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
