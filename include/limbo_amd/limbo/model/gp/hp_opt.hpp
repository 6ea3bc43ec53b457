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
