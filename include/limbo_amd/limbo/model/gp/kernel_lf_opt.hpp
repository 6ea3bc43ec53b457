// limbo/model/gp/kernel_lf_opt.hpp — maximise the log marginal likelihood over the kernel
// hyper-parameters (contract: src/limbo/model/gp/kernel_lf_opt.hpp:56-97).
//
// The reference's objective deep-copies the whole GP (K, L, K^-1: 3 N^2 doubles, 403 MB at
// N = 4096) for EVERY evaluation (kernel_lf_opt.hpp:79).  Here an evaluation is "same X in HBM,
// new theta": each host thread that calls the objective keeps ONE private device clone of the
// original GP for the lifetime of the optimisation and re-runs build -> factor -> solve on it.
// The original GP is untouched until the optimiser returns, exactly as in the reference.
// Interface attribution: the names of this header (the policy and its nested objective type) are those of resibots/limbo
// (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info), file named above — a drop-in must keep them.  What they
// do is written once, in this project's own terms: limbo_amd::fit (hp_opt.hpp).
#ifndef LIMBO_MODEL_GP_KERNEL_LF_OPT_HPP
#define LIMBO_MODEL_GP_KERNEL_LF_OPT_HPP
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <limbo/model/gp/hp_opt.hpp>
#include <limbo/opt/batched_rprop.hpp>
#include <limbo/tools/parallel.hpp>
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
                    limbo_amd::fit::run<Optimizer, limbo_amd::fit::KernelParams, limbo_amd::fit::LogLik, KernelLFOptimization<GP>>(gp);
                }

                /// Addition: the fits of several GPs (the outputs of a MultiGP: multi_gp/parallel_lf_opt.hpp:64-67) in
                /// lock-step — with Rprop as the optimiser, iteration i of every fit is one batched device evaluation.
                /// Each fit works on a private clone and touches its GP only at the end, as operator() does.
                template <typename GP>
                static bool fit_many(std::vector<GP>& gps)
                {
                    if (!opt::is_rprop<Optimizer>::value || !opt::batch_restarts_enabled() || gps.size() < 2)
                        return false;
                    std::vector<std::unique_ptr<GP>> clones;
                    std::vector<GP*> ptrs;
                    std::vector<Eigen::VectorXd> inits;
                    for (auto& g : gps) {
                        clones.emplace_back(new GP(g));
                        ptrs.push_back(clones.back().get());
                        inits.push_back(g.kernel_function().h_params());
                    }
                    auto fb = [&](const std::vector<Eigen::VectorXd>& xs, bool gr) {
                        for (size_t i = 0; i < xs.size(); ++i)
                            ptrs[i]->kernel_function().set_h_params(xs[i]);
                        std::vector<double> liks;
                        std::vector<Eigen::VectorXd> grads;
                        GP::hp_objectives_batched(ptrs, gr, liks, grads);
                        std::vector<opt::eval_t> out;
                        for (size_t i = 0; i < xs.size(); ++i)
                            out.push_back(gr ? opt::eval_t{liks[i], opt::eval_t::second_type(grads[i])} : opt::no_grad(liks[i]));
                        return out;
                    };
                    auto res = opt::rprop_lockstep<typename opt::rprop_params_of<Optimizer, Params>::type>(fb, inits, false); // (Rprop<P>: P's settings)
                    // recompute(false) + compute_log_lik() of operator(): batched for the models that live on the device; a model
                    // that lives on the host stays there (the batched call would move it to the device for good, and
                    // every later add_sample() / query() of a small output model with it)
                    std::vector<GP*> orig;
                    for (size_t i = 0; i < gps.size(); ++i) {
                        gps[i].kernel_function().set_h_params(res[i].first);
                        if (gps[i].host_resident()) {
                            gps[i].recompute(false);
                            gps[i].compute_log_lik();
                        }
                        else
                            orig.push_back(&gps[i]);
                    }
                    std::vector<double> liks;
                    std::vector<Eigen::VectorXd> none;
                    if (!orig.empty())
                        GP::hp_objectives_batched(orig, false, liks, none);
                    return true;
                }

            protected:
                /// the objective (kernel_lf_opt.hpp:72-96) = limbo_amd::fit::Objective over (kernel parameters, log-lik), plus the
                /// batched evaluation below
                template <typename GP>
                struct KernelLFOptimization : public limbo_amd::fit::Objective<Params, GP, limbo_amd::fit::KernelParams, limbo_amd::fit::LogLik> {
                    using Base = limbo_amd::fit::Objective<Params, GP, limbo_amd::fit::KernelParams, limbo_amd::fit::LogLik>;
                    explicit KernelLFOptimization(const GP& gp) : Base(gp), _original_gp(gp) {}

                    /// Addition: the objective at params.size() points at once — one private device clone of the original
                    /// GP per point (kept for the lifetime of the optimisation, dealt over the visible devices), all
                    /// evaluated by one batched launch sequence per device (GP::hp_objectives_batched).  What lets
                    /// opt::ParallelRepeater step its restarts in lock-step.
                    std::vector<opt::eval_t> eval_batch(const std::vector<Eigen::VectorXd>& params, bool compute_grad) const
                    {
                        std::lock_guard<std::mutex> lk(_batch_mu);
                        while (_batch.size() < params.size())
                            _batch.emplace_back(new GP(_original_gp, limbo_amd::deal_device<Params>(_batch.size(), _original_gp.device())));
                        std::vector<GP*> gps;
                        for (size_t i = 0; i < params.size(); ++i) {
                            _batch[i]->kernel_function().set_h_params(params[i]);
                            gps.push_back(_batch[i].get());
                        }
                        std::vector<double> liks;
                        std::vector<Eigen::VectorXd> grads;
                        GP::hp_objectives_batched(gps, compute_grad, liks, grads);
                        std::vector<opt::eval_t> out;
                        for (size_t i = 0; i < params.size(); ++i)
                            out.push_back(compute_grad ? opt::eval_t{liks[i], opt::eval_t::second_type(grads[i])} : opt::no_grad(liks[i]));
                        return out;
                    }

                    /// devices the lock-step clones were put on (instrumentation / tests)
                    std::vector<int> batch_devices() const
                    {
                        std::lock_guard<std::mutex> lk(_batch_mu);
                        std::vector<int> d;
                        for (auto& g : _batch)
                            d.push_back(g->device());
                        return d;
                    }

                protected:
                    const GP& _original_gp;
                    mutable std::mutex _batch_mu;
                    mutable std::vector<std::unique_ptr<GP>> _batch;
                };
            };
        } // namespace gp
    } // namespace model
} // namespace limbo
#endif
