// limbo/acqui/ei.hpp — expected improvement  EI = (mu - f+ - xi) Phi(Z) + sigma phi(Z)
// (contract: src/limbo/acqui/ei.hpp:77-120: returns 0 when sigma < 1e-10 or there is no sample;
// f+ = best predicted mean over the training samples, cached per nb_samples) plus batch().
// The f+ scan, N calls of model.mu() in the reference (ei.hpp:100-103), is one query_batch here.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_ACQUI_EI_HPP
#define LIMBO_ACQUI_EI_HPP
#include <algorithm>
#include <cmath>
#include <tuple>
#include <vector>
#include <Eigen/Core>
#include <limbo/opt/optimizer.hpp>
#include <limbo/tools/macros.hpp>
namespace limbo {
    namespace defaults {
        struct acqui_ei {
            BO_PARAM(double, jitter, 0.0);
        };
    } // namespace defaults
    namespace acqui {
        template <typename Params, typename Model>
        class EI {
        public:
            EI(const Model& model, int /*iteration*/ = 0) : _model(model), _nb_samples(-1), _f_max(0) {}
            size_t dim_in() const { return _model.dim_in(); }
            size_t dim_out() const { return _model.dim_out(); }

            template <typename AggregatorFunction>
            opt::eval_t operator()(const Eigen::VectorXd& v, const AggregatorFunction& afun, bool gradient)
            {
                assert(!gradient);
                (void)gradient;
                Eigen::VectorXd mu;
                double sigma_sq;
                std::tie(mu, sigma_sq) = _model.query(v);
                return opt::no_grad(_value(afun(mu), sigma_sq, afun));
            }
            template <typename AggregatorFunction>
            std::vector<double> batch(const std::vector<Eigen::VectorXd>& points, const AggregatorFunction& afun)
            {
                Eigen::MatrixXd mu;
                Eigen::VectorXd s2;
                _model.query_batch(points, mu, s2);
                std::vector<double> out(points.size());
                for (size_t m = 0; m < points.size(); ++m) {
                    Eigen::VectorXd row(mu.cols());
                    for (int p = 0; p < (int)mu.cols(); ++p)
                        row(p) = mu(m, p);
                    out[m] = _value(afun(row), s2(m), afun);
                }
                return out;
            }

        protected:
            const Model& _model;
            int _nb_samples;
            double _f_max;

            template <typename AggregatorFunction>
            double _value(double fmu, double sigma_sq, const AggregatorFunction& afun)
            {
                const double sigma = std::sqrt(sigma_sq);
                if (sigma < 1e-10 || _model.samples().size() < 1)
                    return 0.0;
                if (_nb_samples != _model.nb_samples()) { // best predicted observation so far
                    Eigen::MatrixXd mu;
                    Eigen::VectorXd s2;
                    _model.query_batch(_model.samples(), mu, s2);
                    double best = -std::numeric_limits<double>::infinity();
                    for (int i = 0; i < (int)mu.rows(); ++i) {
                        Eigen::VectorXd row(mu.cols());
                        for (int p = 0; p < (int)mu.cols(); ++p)
                            row(p) = mu(i, p);
                        best = std::max(best, (double)afun(row));
                    }
                    _nb_samples = _model.nb_samples();
                    _f_max = best;
                }
                const double X = fmu - _f_max - Params::acqui_ei::jitter();
                const double Z = X / sigma;
                const double phi = std::exp(-0.5 * Z * Z) / std::sqrt(2.0 * M_PI);
                const double Phi = 0.5 * std::erfc(-Z / std::sqrt(2));
                return X * Phi + sigma * phi;
            }
        };
    } // namespace acqui
} // namespace limbo
#endif
