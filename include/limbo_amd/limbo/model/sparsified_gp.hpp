// limbo/model/sparsified_gp.hpp — GP on a density-thinned subset of the samples
// (contract: src/limbo/model/sparsified_gp.hpp:56-205).
//
// Same class shape as the reference: a GP subclass whose compute()/add_sample() first thin the data
// to at most Params::model_sparse_gp::max_points() samples.  The thinning itself
// (_sparsify/_get_most_dense_point, :124-183: N - max_points removals, each a partial sort of every
// row of an N x N distance matrix) runs on the device: gpe_sparsify (limbo_amd/csrc/sparsify.hip)
// keeps the distance matrix in HBM and re-scans only the rows a removal invalidates.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_MODEL_SPARSIFIED_GP_HPP
#define LIMBO_MODEL_SPARSIFIED_GP_HPP
#include <stdexcept>
#include <limbo/model/gp.hpp>
namespace limbo {
    namespace defaults {
        struct model_sparse_gp {
            BO_PARAM(int, max_points, 200);
        };
    } // namespace defaults
    namespace model {
        template <typename Params, typename KernelFunction = kernel::MaternFiveHalves<Params>, typename MeanFunction = mean::Data<Params>, typename HyperParamsOptimizer = gp::NoLFOpt<Params>>
        class SparsifiedGP : public GP<Params, KernelFunction, MeanFunction, HyperParamsOptimizer> {
        public:
            using base_gp_t = GP<Params, KernelFunction, MeanFunction, HyperParamsOptimizer>;

            SparsifiedGP() : base_gp_t() {}
            SparsifiedGP(int dim_in, int dim_out) : base_gp_t(dim_in, dim_out) {}

            /// sparsified_gp.hpp:83-102
            void compute(const std::vector<Eigen::VectorXd>& samples, const std::vector<Eigen::VectorXd>& observations, bool compute_kernel = true)
            {
                if ((int)samples.size() <= Params::model_sparse_gp::max_points())
                    base_gp_t::compute(samples, observations, compute_kernel);
                else {
                    std::vector<Eigen::VectorXd> samp, obs;
                    std::tie(samp, obs) = _sparsify(samples, observations);
                    base_gp_t::compute(samp, obs, compute_kernel);
                }
            }

            /// sparsified_gp.hpp:104-120: add, and re-thin + recompute once over the limit
            void add_sample(const Eigen::VectorXd& sample, const Eigen::VectorXd& observation)
            {
                base_gp_t::add_sample(sample, observation);
                if ((int)this->_samples.size() > Params::model_sparse_gp::max_points()) {
                    std::vector<Eigen::VectorXd> observations;
                    for (size_t i = 0; i < this->_samples.size(); i++) {
                        Eigen::VectorXd row(this->_observations.cols());
                        for (int p = 0; p < (int)this->_observations.cols(); ++p)
                            row(p) = this->_observations(i, p);
                        observations.push_back(row);
                    }
                    const std::vector<Eigen::VectorXd> samples = this->_samples; // compute() overwrites _samples
                    compute(samples, observations, true);
                }
            }

        protected:
            /// sparsified_gp.hpp:157-183, on the device
            std::pair<std::vector<Eigen::VectorXd>, std::vector<Eigen::VectorXd>> _sparsify(const std::vector<Eigen::VectorXd>& samples, const std::vector<Eigen::VectorXd>& observations) const
            {
                const int64_t N = samples.size();
                const int D = samples[0].size();
                std::vector<double> X((size_t)N * D);
                for (int64_t i = 0; i < N; ++i)
                    for (int d = 0; d < D; ++d)
                        X[(size_t)i * D + d] = samples[i](d);
                std::vector<int64_t> keep((size_t)N);
                int64_t n_keep = 0;
                const int rc = gpe_sparsify(this->_eng.device(), X.data(), N, D, Params::model_sparse_gp::max_points(), keep.data(), &n_keep);
                if (rc != GPE_OK)
                    throw std::runtime_error("gpe_sparsify failed with status " + std::to_string(rc));
                std::vector<Eigen::VectorXd> samp, obs;
                samp.reserve(n_keep);
                obs.reserve(n_keep);
                for (int64_t k = 0; k < n_keep; ++k) {
                    samp.push_back(samples[keep[k]]);
                    obs.push_back(observations[keep[k]]);
                }
                return std::make_pair(samp, obs);
            }
        };
    } // namespace model
} // namespace limbo
#endif
