// limbo/model/gp/hp_opt.hpp — base of the hyper-parameter optimisation policies
// (contract: src/limbo/model/gp/hp_opt.hpp:58-73: default/copy constructible, warns if never used)
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_MODEL_GP_HP_OPT_HPP
#define LIMBO_MODEL_GP_HP_OPT_HPP
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <limbo/opt/rprop.hpp>
#include <limbo/tools/macros.hpp>

#include "../../../../gpe.h"

namespace limbo_amd {
    /// Params::gpu::device() when the user's Params has it, defaults::gpu::device() (-1: automatic) otherwise
    template <typename Params, typename = void>
    struct param_device {
        static int get() { return limbo::defaults::gpu::device(); }
    };
    template <typename Params>
    struct param_device<Params, decltype((void)Params::gpu::device())> {
        static int get() { return Params::gpu::device(); }
    };
    inline int visible_devices()
    {
        int n = 1;
        if (gpe_device_count(&n) != GPE_OK || n < 1)
            n = 1;
        return n;
    }
    /// where a GP the user creates lives
    template <typename Params>
    inline int home_device()
    {
        const int d = param_device<Params>::get();
        return d >= 0 ? d : 0;
    }
    /// where the k-th of a family of independent GPs goes when the family starts on device `first`: pinned by
    /// Params::gpu::device() >= 0, otherwise round-robin over the visible devices — what tools::par::loop / par::max
    /// (tools/parallel.hpp:138-191) are to host cores in the reference, with GPUs in place of cores
    template <typename Params>
    inline int deal_device(size_t k, int first = 0)
    {
        const int d = param_device<Params>::get();
        return d >= 0 ? d : (int)((first + k) % (size_t)visible_devices());
    }

    /// One private device clone of the original GP per host thread that evaluates a hyper-parameter objective, kept
    /// for the lifetime of the optimisation (the reference deep-copies the whole GP per EVALUATION,
    /// kernel_lf_opt.hpp:79).  The clones of concurrent restarts (opt::ParallelRepeater) are dealt over the devices,
    /// the first one staying where the original is; the arg-max over restarts happens on the host.
    template <typename Params, typename GP>
    class WorkerClones {
    public:
        GP& get(const GP& original) const
        {
            std::lock_guard<std::mutex> lk(_mu);
            auto& w = _workers[std::this_thread::get_id()];
            if (!w)
                w.reset(new GP(original, deal_device<Params>(_workers.size() - 1, original.device())));
            return *w;
        }
        size_t size() const
        {
            std::lock_guard<std::mutex> lk(_mu);
            return _workers.size();
        }
        /// devices the clones were put on (instrumentation / tests)
        std::vector<int> devices() const
        {
            std::lock_guard<std::mutex> lk(_mu);
            std::vector<int> d;
            for (auto& kv : _workers)
                d.push_back(kv.second->device());
            return d;
        }

    private:
        mutable std::mutex _mu;
        mutable std::map<std::thread::id, std::unique_ptr<GP>> _workers;
    };

    // ---- the hyper-parameter fits, written once ----------------------------------------------------------------------
    // limbo's four fitting policies (model/gp/kernel_lf_opt.hpp, kernel_loo_opt.hpp, mean_lf_opt.hpp, kernel_mean_lf_opt.hpp)
    // differ in two things only: WHICH hyper-parameters move — a Subject: read them from a GP, put them into a GP and
    // refresh what depends on them — and WHAT is maximised — a Score: its value and its gradient pieces.  fit::Objective
    // is the functor the optimiser calls (one persistent device clone per calling host thread: WorkerClones), fit::run the
    // policy's operator().  The policy headers instantiate them; the names limbo's users and tests see stay theirs.
    namespace fit {
        inline Eigen::VectorXd join(const Eigen::VectorXd& a, const Eigen::VectorXd& b)
        {
            Eigen::VectorXd v(a.size() + b.size());
            for (int i = 0; i < (int)a.size(); ++i)
                v(i) = a(i);
            for (int i = 0; i < (int)b.size(); ++i)
                v(a.size() + i) = b(i);
            return v;
        }
        /// log marginal likelihood (gp.hpp:267-330)
        struct LogLik {
            template <typename GP> static double value(GP& gp) { return gp.compute_log_lik(); }
            template <typename GP> static Eigen::VectorXd wrt_kernel(GP& gp) { return gp.compute_kernel_grad_log_lik(); }
            template <typename GP> static Eigen::VectorXd wrt_mean(GP& gp) { return gp.compute_mean_grad_log_lik(); }
        };
        /// leave-one-out cross-validation log probability (gp.hpp:339-402)
        struct LogLooCv {
            template <typename GP> static double value(GP& gp) { return gp.compute_log_loo_cv(); }
            template <typename GP> static Eigen::VectorXd wrt_kernel(GP& gp) { return gp.compute_kernel_grad_log_loo_cv(); }
        };
        /// the kernel's hyper-parameters: a new kernel matrix, same obs_mean (recompute(false))
        struct KernelParams {
            static constexpr bool copies_original = false;
            template <typename GP> static void prepare(GP&) {}
            template <typename GP> static Eigen::VectorXd read(const GP& gp) { return gp.kernel_function().h_params(); }
            template <typename GP> static void put(GP& gp, const Eigen::VectorXd& x)
            {
                gp.kernel_function().set_h_params(x);
                gp.recompute(false);
            }
            template <typename Score, typename GP> static Eigen::VectorXd gradient(GP& gp) { return Score::wrt_kernel(gp); }
        };
        /// the mean function's: same factor, new obs_mean and alpha (recompute(true, false)); K^-1 is formed once on a copy of
        /// the original that every worker clone then inherits (mean_lf_opt.hpp:78)
        struct MeanParams {
            static constexpr bool copies_original = true;
            template <typename GP> static void prepare(GP& original) { original.compute_inv_kernel(); }
            template <typename GP> static Eigen::VectorXd read(const GP& gp) { return gp.mean_function().h_params(); }
            template <typename GP> static void put(GP& gp, const Eigen::VectorXd& x)
            {
                gp.mean_function().set_h_params(x);
                gp.recompute(true, false);
            }
            template <typename Score, typename GP> static Eigen::VectorXd gradient(GP& gp) { return Score::wrt_mean(gp); }
        };
        /// both, the kernel's first (kernel_mean_lf_opt.hpp:63-66): everything is recomputed
        struct KernelAndMeanParams {
            static constexpr bool copies_original = false;
            template <typename GP> static void prepare(GP&) {}
            template <typename GP> static Eigen::VectorXd read(const GP& gp) { return join(gp.kernel_function().h_params(), gp.mean_function().h_params()); }
            template <typename GP> static void put(GP& gp, const Eigen::VectorXd& x)
            {
                const int nk = gp.kernel_function().h_params_size(), nm = gp.mean_function().h_params_size();
                gp.kernel_function().set_h_params(Eigen::VectorXd(x.head(nk)));
                gp.mean_function().set_h_params(Eigen::VectorXd(x.tail(nm)));
                gp.recompute(true);
            }
            template <typename Score, typename GP> static Eigen::VectorXd gradient(GP& gp) { return join(Score::wrt_kernel(gp), Score::wrt_mean(gp)); }
        };

        // the original GP by reference, or (Subject::copies_original) by value after Subject::prepare
        template <typename GP, bool Copy> struct Original {
            const GP& gp;
            explicit Original(const GP& g) : gp(g) {}
        };
        template <typename GP> struct Original<GP, true> {
            GP gp;
            explicit Original(const GP& g) : gp(g) {}
        };

        template <typename Params, typename GP, typename Subject, typename Score>
        class Objective {
        public:
            explicit Objective(const GP& gp) : _original(gp) { Subject::prepare(const_cast<GP&>(_original.gp)); }
            limbo::opt::eval_t operator()(const Eigen::VectorXd& x, bool want_gradient) const
            {
                GP& worker = _workers.get(_original.gp);
                Subject::put(worker, x);
                const double v = Score::value(worker);
                if (!want_gradient)
                    return limbo::opt::no_grad(v);
                return {v, limbo::opt::eval_t::second_type(Subject::template gradient<Score>(worker))};
            }

        protected:
            Original<GP, Subject::copies_original> _original;
            WorkerClones<Params, GP> _workers;
        };

        /// a policy's operator(): optimise from where the GP stands, leave the GP at the optimum with its score computed
        template <typename Optimizer, typename Subject, typename Score, typename Obj, typename GP>
        inline void run(GP& gp)
        {
            Obj objective(gp);
            Optimizer optimizer;
            const Eigen::VectorXd best = optimizer(objective, Subject::read(gp), false);
            Subject::put(gp, best);
            Score::value(gp);
        }
    } // namespace fit
} // namespace limbo_amd

namespace limbo {
    namespace model {
        namespace gp {
            template <typename Params, typename Optimizer = opt::Rprop<Params>>
            struct HPOpt {
            public:
                HPOpt() : _called(false) {}
                HPOpt(const HPOpt&) : _called(true) {} // copies never warn
                HPOpt& operator=(const HPOpt&) { return *this; }
                void mark_called() { _called = true; } // (addition: a policy object used only through a static batched path)
                ~HPOpt()
                {
                    if (!_called)
                        std::cerr << "'HPOpt' was never called!" << std::endl;
                }

            protected:
                bool _called;
            };
        } // namespace gp
    } // namespace model
} // namespace limbo
#endif
