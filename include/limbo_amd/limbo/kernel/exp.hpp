// limbo/kernel/exp.hpp — isotropic squared exponential  k = sigma_f^2 exp(-|x-y|^2 / (2 l^2))
// hyper-parameters (log-space): [log l, log sigma_f]   (contract: src/limbo/kernel/exp.hpp:73-122)
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_KERNEL_EXP_HPP
#define LIMBO_KERNEL_EXP_HPP
#include <limbo/kernel/kernel.hpp>
namespace limbo {
    namespace defaults {
        struct kernel_exp {
            BO_PARAM(double, sigma_sq, 1);
            BO_PARAM(double, l, 1);
        };
    } // namespace defaults
    namespace kernel {
        template <typename Params>
        struct Exp : public BaseKernel<Params, Exp<Params>> {
            Exp(size_t /*dim*/ = 1) : _sf2(Params::kernel_exp::sigma_sq()), _l(Params::kernel_exp::l()), _h_params(2)
            {
                _h_params(0) = std::log(_l);
                _h_params(1) = std::log(std::sqrt(_sf2));
            }
            size_t params_size() const { return 2; }
            Eigen::VectorXd params() const { return _h_params; }
            void set_params(const Eigen::VectorXd& p)
            {
                _h_params = p;
                _l = std::exp(p(0));
                _sf2 = std::exp(2.0 * p(1));
            }
            double kernel(const Eigen::VectorXd& a, const Eigen::VectorXd& b) const { return _sf2 * std::exp(-0.5 * _r(a, b)); }
            Eigen::VectorXd gradient(const Eigen::VectorXd& a, const Eigen::VectorXd& b) const
            {
                const double r = _r(a, b), k = _sf2 * std::exp(-0.5 * r);
                Eigen::VectorXd g(2);
                g(0) = r * k;
                g(1) = 2.0 * k;
                return g;
            }

        protected:
            double _sf2, _l;
            Eigen::VectorXd _h_params;
            double _r(const Eigen::VectorXd& a, const Eigen::VectorXd& b) const
            {
                double s = 0.0;
                for (int i = 0; i < (int)a.size(); ++i)
                    s += (a(i) - b(i)) * (a(i) - b(i));
                return s / (_l * _l);
            }
        };
    } // namespace kernel
} // namespace limbo
namespace limbo_amd {
    template <typename Params>
    struct device_kernel<limbo::kernel::Exp<Params>> {
        static constexpr int kind = KIND_EXP;
    };
} // namespace limbo_amd
#endif
