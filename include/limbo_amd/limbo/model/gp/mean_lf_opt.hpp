// limbo/model/gp/mean_lf_opt.hpp — maximise the log marginal likelihood over the MEAN function's
// hyper-parameters only (contract: src/limbo/model/gp/mean_lf_opt.hpp:55-100).
// K and L do not change during this optimisation: an evaluation is recompute(true, false), i.e. new
// obs_mean -> two triangular sweeps on the device (gpe_update_alpha); the factor stays in HBM.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_MODEL_GP_MEAN_LF_OPT_HPP
#define LIMBO_MODEL_GP_MEAN_LF_OPT_HPP
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <limbo/model/gp/hp_opt.hpp>
namespace limbo {
    namespace model {
        namespace gp {
            template <typename Params, typename Optimizer = opt::Rprop<Params>>
            struct MeanLFOpt : public HPOpt<Params, Optimizer> {
            public:
                template <typename GP>
                void operator()(GP& gp)
                {
                    this->_called = true;
                    MeanLFOptimization<GP> optimization(gp);
                    Optimizer optimizer;
                    Eigen::VectorXd params = optimizer(optimization, gp.mean_function().h_params(), false);
                    gp.mean_function().set_h_params(params);
                    gp.recompute(true, false);
                    gp.compute_log_lik();
                }

            protected:
                template <typename GP>
                struct MeanLFOptimization {
                public:
                    MeanLFOptimization(const GP& gp) : _original_gp(gp)
                    {
                        _original_gp.compute_inv_kernel(); // mean_lf_opt.hpp:78: every worker copy inherits K^-1
                    }

                    opt::eval_t operator()(const Eigen::VectorXd& params, bool compute_grad) const
                    {
                        GP& gp = _workers.get(_original_gp);
                        gp.mean_function().set_h_params(params);
                        gp.recompute(true, false);
                        const double lik = gp.compute_log_lik();
                        if (!compute_grad)
                            return opt::no_grad(lik);
                        return {lik, opt::eval_t::second_type(gp.compute_mean_grad_log_lik())};
                    }

                protected:
                    GP _original_gp;
                    limbo_amd::WorkerClones<Params, GP> _workers;
                };
            };
        } // namespace gp
    } // namespace model
} // namespace limbo
#endif
