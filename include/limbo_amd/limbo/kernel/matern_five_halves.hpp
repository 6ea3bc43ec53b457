// limbo/kernel/matern_five_halves.hpp — isotropic Matern 5/2
//   k(x, y) = sigma_f^2 (1 + sqrt5 d/l + 5 d^2 / (3 l^2)) exp(-sqrt5 d/l),  d = |x - y|
// hyper-parameters (log-space): [log l, log sigma_f]
// (policy contract and formulas: src/limbo/kernel/matern_five_halves.hpp:83-139).
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_KERNEL_MATERN_FIVE_HALVES_HPP
#define LIMBO_KERNEL_MATERN_FIVE_HALVES_HPP

#include <limbo/kernel/kernel.hpp>

namespace limbo {
    namespace defaults {
        struct kernel_maternfivehalves {
            BO_PARAM(double, sigma_sq, 1);
            BO_PARAM(double, l, 1);
        };
    } // namespace defaults

    namespace kernel {
        template <typename Params>
        struct MaternFiveHalves : public BaseKernel<Params, MaternFiveHalves<Params>> {
            MaternFiveHalves(size_t /*dim*/ = 1) : _sf2(Params::kernel_maternfivehalves::sigma_sq()), _l(Params::kernel_maternfivehalves::l()), _h_params(2)
            {
                _h_params(0) = std::log(_l);
                _h_params(1) = std::log(std::sqrt(_sf2));
            }

            size_t params_size() const { return 2; }
            Eigen::VectorXd params() const { return _h_params; }

            void set_params(const Eigen::VectorXd& p)
            {
                assert(p.size() == 2);
                _h_params = p;
                _l = std::exp(p(0));
                _sf2 = std::exp(2.0 * p(1));
            }

            double kernel(const Eigen::VectorXd& v1, const Eigen::VectorXd& v2) const
            {
                const double d = _dist(v1, v2);
                const double t1 = std::sqrt(5.0) * d / _l;
                const double t2 = 5.0 * d * d / (3.0 * _l * _l);
                return _sf2 * (1.0 + t1 + t2) * std::exp(-t1);
            }

            Eigen::VectorXd gradient(const Eigen::VectorXd& x1, const Eigen::VectorXd& x2) const
            {
                const double d = _dist(x1, x2);
                const double t1 = std::sqrt(5.0) * d / _l;
                const double t2 = 5.0 * d * d / (3.0 * _l * _l);
                const double r = std::exp(-t1);
                Eigen::VectorXd g(2);
                // d t1/d log l = -t1, d t2/d log l = -2 t2, d exp(-t1)/d log l = t1 exp(-t1)
                g(0) = _sf2 * (r * t1 * (1.0 + t1 + t2) + (-t1 - 2.0 * t2) * r);
                g(1) = 2.0 * _sf2 * (1.0 + t1 + t2) * r;
                return g;
            }

        protected:
            double _sf2, _l;
            Eigen::VectorXd _h_params;

            static double _dist(const Eigen::VectorXd& a, const Eigen::VectorXd& b)
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
    struct device_kernel<limbo::kernel::MaternFiveHalves<Params>> {
        static constexpr int kind = KIND_MATERN52;
    };
} // namespace limbo_amd

#endif
