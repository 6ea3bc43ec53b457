// limbo/acqui/ucb.hpp — UCB(x) = mu(x) + alpha sqrt(sigma^2(x))   (contract: src/limbo/acqui/ucb.hpp:71-95)
// plus batch(): the same value for M points through one GP::query_batch — row N1 of SURVEY.md §8f:
// the acquisition optimiser is the caller that turns per-point query() into the device batch.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_ACQUI_UCB_HPP
#define LIMBO_ACQUI_UCB_HPP
#include <cmath>
#include <tuple>
#include <vector>
#include <Eigen/Core>
#include <limbo/opt/optimizer.hpp>
#include <limbo/tools/macros.hpp>
namespace limbo {
    namespace defaults {
        struct acqui_ucb {
            BO_PARAM(double, alpha, 0.5);
        };
    } // namespace defaults
    namespace acqui {
        template <typename Params, typename Model>
        class UCB {
        public:
            UCB(const Model& model, int /*iteration*/ = 0) : _model(model) {}
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
                return opt::no_grad(afun(mu) + Params::acqui_ucb::alpha() * std::sqrt(sigma));
            }

            /// values[m] == (*this)(points[m], afun, false).first, one device batch
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
                    out[m] = afun(row) + Params::acqui_ucb::alpha() * std::sqrt(s2(m));
                }
                return out;
            }

        protected:
            const Model& _model;
        };
    } // namespace acqui
} // namespace limbo
#endif
