// limbo/kernel/matern_three_halves.hpp — isotropic Matern 3/2
//   k = sigma_f^2 (1 + sqrt3 d/l) exp(-sqrt3 d/l);  hyper-parameters [log l, log sigma_f]
// (contract: src/limbo/kernel/matern_three_halves.hpp:80-132)
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_KERNEL_MATERN_THREE_HALVES_HPP
#define LIMBO_KERNEL_MATERN_THREE_HALVES_HPP
#include <limbo/kernel/kernel.hpp>
namespace limbo {
    namespace defaults {
        struct kernel_maternthreehalves {
            BO_PARAM(double, sigma_sq, 1);
            BO_PARAM(double, l, 1);
        };
    } // namespace defaults
    namespace kernel {
        template <typename Params>
        struct MaternThreeHalves : public BaseKernel<Params, MaternThreeHalves<Params>> {
            MaternThreeHalves(size_t /*dim*/ = 1) : _sf2(Params::kernel_maternthreehalves::sigma_sq()), _l(Params::kernel_maternthreehalves::l()), _h_params(2)
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
            double kernel(const Eigen::VectorXd& a, const Eigen::VectorXd& b) const
            {
                const double t = std::sqrt(3.0) * _d(a, b) / _l;
                return _sf2 * (1.0 + t) * std::exp(-t);
            }
            Eigen::VectorXd gradient(const Eigen::VectorXd& a, const Eigen::VectorXd& b) const
            {
                const double t = std::sqrt(3.0) * _d(a, b) / _l, r = std::exp(-t);
                Eigen::VectorXd g(2);
                g(0) = _sf2 * (-t * r + (1.0 + t) * t * r);
                g(1) = 2.0 * _sf2 * (1.0 + t) * r;
                return g;
            }

        protected:
            double _sf2, _l;
            Eigen::VectorXd _h_params;
            static double _d(const Eigen::VectorXd& a, const Eigen::VectorXd& b)
            {
                double s = 0.0;
                for (int i = 0; i < (int)a.size(); ++i)
                    s += (a(i) - b(i)) * (a(i) - b(i));
                return std::sqrt(s);
            }
        };
    } // namespace kernel
} // namespace limbo
namespace limbo_amd {
    template <typename Params>
    struct device_kernel<limbo::kernel::MaternThreeHalves<Params>> {
        static constexpr int kind = KIND_MATERN32;
    };
} // namespace limbo_amd
#endif
