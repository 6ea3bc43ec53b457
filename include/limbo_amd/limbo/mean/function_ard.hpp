// limbo/mean/function_ard.hpp — affine transform of another mean function with learnable
// coefficients (contract: src/limbo/mean/function_ard.hpp:55-129):
//     m'(x) = T [m(x); 1],   T: dim_out x (dim_out + 1), initialised to [I | 0] (:64-66);
// h_params = [T row-major | inner mean's h_params] (:74-92).  Written with element loops only, so it
// compiles against Eigen and against the minimal eigen_shim alike.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_MEAN_FUNCTION_ARD_HPP
#define LIMBO_MEAN_FUNCTION_ARD_HPP
#include <limbo/mean/mean.hpp>
namespace limbo {
    namespace mean {
        template <typename Params, typename MeanFunction>
        struct FunctionARD : public BaseMean<Params> {
            FunctionARD(size_t dim_out = 1) : _mean_function(dim_out), _tr(dim_out, dim_out + 1)
            {
                const size_t nin = _mean_function.h_params_size();
                Eigen::VectorXd h = Eigen::VectorXd::Zero(dim_out * (dim_out + 1) + nin);
                for (size_t i = 0; i < dim_out; i++)
                    h[i * (dim_out + 2)] = 1;
                if (nin > 0) {
                    const Eigen::VectorXd hin = _mean_function.h_params();
                    for (size_t k = 0; k < nin; ++k)
                        h[dim_out * (dim_out + 1) + k] = hin[k];
                }
                this->set_h_params(h);
            }

            size_t h_params_size() const { return _tr.rows() * _tr.cols() + _mean_function.h_params_size(); }

            Eigen::VectorXd h_params() const
            {
                const int nt = _tr.rows() * _tr.cols(), nin = _mean_function.h_params_size();
                Eigen::VectorXd params(nt + nin);
                for (int k = 0; k < nt; ++k)
                    params[k] = _h_params[k];
                if (nin > 0) {
                    const Eigen::VectorXd hin = _mean_function.h_params();
                    for (int k = 0; k < nin; ++k)
                        params[nt + k] = hin[k];
                }
                return params;
            }

            void set_h_params(const Eigen::VectorXd& p)
            {
                const int nt = _tr.rows() * _tr.cols(), nin = _mean_function.h_params_size();
                _h_params = Eigen::VectorXd(nt);
                for (int k = 0; k < nt; ++k)
                    _h_params[k] = p[k];
                for (int c = 0; c < _tr.cols(); c++)
                    for (int r = 0; r < _tr.rows(); r++)
                        _tr(r, c) = p[r * _tr.cols() + c];
                if (nin > 0) {
                    Eigen::VectorXd hin(nin);
                    for (int k = 0; k < nin; ++k)
                        hin[k] = p[nt + k];
                    _mean_function.set_h_params(hin);
                }
            }

            template <typename GP>
            Eigen::MatrixXd grad(const Eigen::VectorXd& x, const GP& gp) const
            {
                const int R = _tr.rows(), C = _tr.cols(), nin = _mean_function.h_params_size();
                Eigen::MatrixXd grad = Eigen::MatrixXd::Zero(R, h_params_size());
                Eigen::VectorXd m = _mean_function(x, gp);
                for (int i = 0; i < R; i++) { // d m'_i / d T(i, :) = [m(x); 1]   (:100-103)
                    for (int c = 0; c < C - 1; ++c)
                        grad(i, i * C + c) = m[c];
                    grad(i, (i + 1) * C - 1) = 1;
                }
                if (nin > 0) { // chain rule through the inner mean: T [d m / d h; 0]   (:104-109)
                    Eigen::MatrixXd gin = _mean_function.grad(x, gp);
                    for (int i = 0; i < R; ++i)
                        for (int k = 0; k < nin; ++k) {
                            double s = 0.0;
                            for (int c = 0; c < R; ++c)
                                s += _tr(i, c) * gin(c, k);
                            grad(i, R * C + k) = s;
                        }
                }
                return grad;
            }

            template <typename GP>
            Eigen::VectorXd operator()(const Eigen::VectorXd& x, const GP& gp) const
            {
                Eigen::VectorXd m = _mean_function(x, gp);
                Eigen::VectorXd out(_tr.rows());
                for (int r = 0; r < _tr.rows(); ++r) {
                    double s = _tr(r, _tr.cols() - 1);
                    for (int c = 0; c < _tr.cols() - 1; ++c)
                        s += _tr(r, c) * m[c];
                    out[r] = s;
                }
                return out;
            }

        protected:
            MeanFunction _mean_function;
            Eigen::MatrixXd _tr;
            Eigen::VectorXd _h_params;
        };
    } // namespace mean
} // namespace limbo
#endif
