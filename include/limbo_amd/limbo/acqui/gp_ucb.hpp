// limbo/acqui/gp_ucb.hpp — GP-UCB: mu + kappa sigma, kappa = sqrt(2 log(n^(D/2+2) pi^2 / (3 delta)))
// (contract: src/limbo/acqui/gp_ucb.hpp:83-107) plus batch() over GP::query_batch.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_ACQUI_GP_UCB_HPP
#define LIMBO_ACQUI_GP_UCB_HPP
#include <cmath>
#include <tuple>
#include <vector>
#include <Eigen/Core>
#include <limbo/opt/optimizer.hpp>
#include <limbo/tools/macros.hpp>
namespace limbo {
    namespace defaults {
        struct acqui_gpucb {
            BO_PARAM(double, delta, 0.1);
        };
    } // namespace defaults
    namespace acqui {
        template <typename Params, typename Model>
        class GP_UCB {
        public:
            GP_UCB(const Model& model, int iteration) : _model(model)
            {
                const double nt = std::pow(iteration, dim_in() / 2.0 + 2.0);
                _beta = std::sqrt(2.0 * std::log(nt * M_PI * M_PI / (Params::acqui_gpucb::delta() * 3)));
            }
            size_t dim_in() const { return _model.dim_in(); }
            size_t dim_out() const { return _model.dim_out(); }

            template <typename AggregatorFunction>
            opt::eval_t operator()(const Eigen::VectorXd& v, const AggregatorFunction& afun, bool gradient) const
            {
                assert(!gradient);
                (void)gradient;
                Eigen::VectorXd mu;
                double sigma;
                std::tie(mu, sigma) = _model.query(v);
                return opt::no_grad(afun(mu) + _beta * std::sqrt(sigma));
            }
            template <typename AggregatorFunction>
            std::vector<double> batch(const std::vector<Eigen::VectorXd>& points, const AggregatorFunction& afun) const
            {
                Eigen::MatrixXd mu;
                Eigen::VectorXd s2;
                _model.query_batch(points, mu, s2);
                std::vector<double> out(points.size());
                for (size_t m = 0; m < points.size(); ++m) {
                    Eigen::VectorXd row(mu.cols());
                    for (int p = 0; p < (int)mu.cols(); ++p)
                        row(p) = mu(m, p);
                    out[m] = afun(row) + _beta * std::sqrt(s2(m));
                }
                return out;
            }

        protected:
            const Model& _model;
            double _beta;
        };
    } // namespace acqui
} // namespace limbo
#endif
