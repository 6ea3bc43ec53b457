// limbo/kernel/kernel.hpp — CRTP base of the kernel functors (policy contract of
// src/limbo/kernel/kernel.hpp:73-146): noise handling, log-space hyper-parameters, grad().
// The functors stay ordinary host code: they define the semantics, they are what user code and
// acquisition functions may call directly, and they are the fallback for kernels the engine has
// no device code for.  `limbo_amd::device_kernel<K>` (bottom of each kernel header) tells the GP
// which engine kind evaluates the same function on the MI355X.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_KERNEL_KERNEL_HPP
#define LIMBO_KERNEL_KERNEL_HPP

#include <Eigen/Core>
#include <cassert>
#include <cmath>

#include <limbo/tools/macros.hpp>

namespace limbo {
    namespace defaults {
        struct kernel {
            /// signal noise sigma_n^2
            BO_PARAM(double, noise, 0.01);
            BO_PARAM(bool, optimize_noise, false);
        };
    } // namespace defaults

    namespace kernel {
        template <typename Params, typename Kernel>
        struct BaseKernel {
        public:
            BaseKernel(size_t /*dim*/ = 1) : _noise(Params::kernel::noise()), _noise_p(std::log(std::sqrt(Params::kernel::noise()))) {}

            /// k(v1, v2); the jitter noise + 1e-8 is added only when the caller says the two
            /// arguments are the same training sample (i == j) — query-time calls use the defaults
            double operator()(const Eigen::VectorXd& v1, const Eigen::VectorXd& v2, int i = -1, int j = -2) const
            {
                const double k = derived().kernel(v1, v2);
                return (i == j) ? k + _noise + 1e-8 : k;
            }

            /// d k / d (log hyper-parameters); with optimize_noise one more entry: 2 sigma_n^2 delta_ij
            Eigen::VectorXd grad(const Eigen::VectorXd& x1, const Eigen::VectorXd& x2, int i = -1, int j = -2) const
            {
                Eigen::VectorXd g = derived().gradient(x1, x2);
                if (Params::kernel::optimize_noise()) {
                    Eigen::VectorXd gn(g.size() + 1);
                    for (int q = 0; q < (int)g.size(); ++q)
                        gn(q) = g(q);
                    gn(g.size()) = (i == j) ? 2.0 * _noise : 0.0;
                    return gn;
                }
                return g;
            }

            size_t h_params_size() const { return derived().params_size() + (Params::kernel::optimize_noise() ? 1 : 0); }

            /// hyper-parameters in log-space (noise last, as log sigma_n)
            Eigen::VectorXd h_params() const
            {
                Eigen::VectorXd p = derived().params();
                if (Params::kernel::optimize_noise()) {
                    Eigen::VectorXd pn(p.size() + 1);
                    for (int q = 0; q < (int)p.size(); ++q)
                        pn(q) = p(q);
                    pn(p.size()) = _noise_p;
                    return pn;
                }
                return p;
            }

            void set_h_params(const Eigen::VectorXd& p)
            {
                const int nk = (int)derived().params_size();
                assert((int)p.size() == (int)h_params_size());
                Eigen::VectorXd pk(nk);
                for (int q = 0; q < nk; ++q)
                    pk(q) = p(q);
                derived().set_params(pk);
                if (Params::kernel::optimize_noise()) {
                    _noise_p = p(nk);
                    _noise = std::exp(2.0 * _noise_p);
                }
            }

            double noise() const { return _noise; }

        protected:
            double _noise;
            double _noise_p;

            Kernel& derived() { return *static_cast<Kernel*>(this); }
            const Kernel& derived() const { return *static_cast<const Kernel*>(this); }
        };
    } // namespace kernel
} // namespace limbo

namespace limbo_amd {
    /// engine kinds, numerically equal to gpe_kernel_kind in include/gpe.h
    enum { KIND_SE_ARD = 0, KIND_MATERN52 = 1, KIND_MATERN32 = 2, KIND_EXP = 3, KIND_HOST_K = 4 };
    /// default: no device code — K is built by the functor on the host and uploaded
    template <typename Kernel>
    struct device_kernel {
        static constexpr int kind = KIND_HOST_K;
    };
} // namespace limbo_amd

#endif
