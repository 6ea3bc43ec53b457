// limbo/kernel/squared_exp_ard.hpp — squared exponential with automatic relevance determination
//   k(x, y) = sigma_f^2 exp(-1/2 (x-y)^T M (x-y)),  M = Lambda Lambda^T + diag(ell)^-2
// hyper-parameters (log-space): [log ell_1..D, (Lambda column-major, k columns), log sigma_f]
// (policy contract and formulas: src/limbo/kernel/squared_exp_ard.hpp:81-161).
// The MI355X engine evaluates it for every k (GPE_KERNEL_SE_ARD; the number of Lambda columns follows from
// the parameter count): (x-y)^T M (x-y) is a sum of squares over the D inputs scaled by 1/ell and the k
// projections Lambda^T x, which the engine keeps as k extra rows of its sample matrix.  Engine limits:
// k <= D and D + D k + 1 <= 64 parameters (gpe_compute returns GPE_ERR_ARG beyond them).
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_KERNEL_SQUARED_EXP_ARD_HPP
#define LIMBO_KERNEL_SQUARED_EXP_ARD_HPP

#include <limbo/kernel/kernel.hpp>

namespace limbo {
    namespace defaults {
        struct kernel_squared_exp_ard {
            BO_PARAM(int, k, 0);
            BO_PARAM(double, sigma_sq, 1);
        };
    } // namespace defaults

    namespace kernel {
        template <typename Params>
        struct SquaredExpARD : public BaseKernel<Params, SquaredExpARD<Params>> {
            SquaredExpARD(int dim = 1) : _dim(dim), _nk(Params::kernel_squared_exp_ard::k()), _ell(dim), _A(dim, Params::kernel_squared_exp_ard::k())
            {
                Eigen::VectorXd p = Eigen::VectorXd::Zero(params_size());
                p(p.size() - 1) = std::log(std::sqrt(Params::kernel_squared_exp_ard::sigma_sq()));
                set_params(p);
            }

            size_t params_size() const { return _dim + _dim * _nk + 1; }
            Eigen::VectorXd params() const { return _h_params; }

            void set_params(const Eigen::VectorXd& p)
            {
                assert((size_t)p.size() == params_size());
                _h_params = p;
                for (int i = 0; i < _dim; ++i)
                    _ell(i) = std::exp(p(i));
                for (int j = 0; j < _nk; ++j)
                    for (int i = 0; i < _dim; ++i)
                        _A(i, j) = p((j + 1) * _dim + i);
                _sf2 = std::exp(2.0 * p(p.size() - 1));
            }

            double kernel(const Eigen::VectorXd& x1, const Eigen::VectorXd& x2) const
            {
                assert((int)x1.size() == _dim && (int)x2.size() == _dim);
                return _sf2 * std::exp(-0.5 * _quad(x1, x2));
            }

            Eigen::VectorXd gradient(const Eigen::VectorXd& x1, const Eigen::VectorXd& x2) const
            {
                Eigen::VectorXd g(params_size());
                const double k = _sf2 * std::exp(-0.5 * _quad(x1, x2));
                for (int i = 0; i < _dim; ++i) {
                    const double z = (x1(i) - x2(i)) / _ell(i);
                    g(i) = z * z * k; // d/d log ell_i
                }
                for (int j = 0; j < _nk; ++j) { // d/d Lambda(:, j) = -((x-y)^T Lambda_j) (x-y) k
                    double proj = 0.0;
                    for (int i = 0; i < _dim; ++i)
                        proj += (x1(i) - x2(i)) * _A(i, j);
                    for (int i = 0; i < _dim; ++i)
                        g((j + 1) * _dim + i) = -proj * (x1(i) - x2(i)) * k;
                }
                g(g.size() - 1) = 2.0 * k; // d/d log sigma_f
                return g;
            }

            const Eigen::VectorXd& ell() const { return _ell; }

        protected:
            int _dim, _nk;
            double _sf2;
            Eigen::VectorXd _ell;
            Eigen::MatrixXd _A;
            Eigen::VectorXd _h_params;

            double _quad(const Eigen::VectorXd& x1, const Eigen::VectorXd& x2) const
            {
                double z = 0.0;
                for (int i = 0; i < _dim; ++i) {
                    const double d = (x1(i) - x2(i)) / _ell(i);
                    z += d * d;
                }
                for (int j = 0; j < _nk; ++j) {
                    double proj = 0.0;
                    for (int i = 0; i < _dim; ++i)
                        proj += (x1(i) - x2(i)) * _A(i, j);
                    z += proj * proj;
                }
                return z;
            }
        };
    } // namespace kernel
} // namespace limbo

namespace limbo_amd {
    template <typename Params>
    struct device_kernel<limbo::kernel::SquaredExpARD<Params>> {
        static constexpr int kind = KIND_SE_ARD;
    };
} // namespace limbo_amd

#endif
