// limbo/model/multi_gp.hpp — dim_out independent single-output GPs behind the model concept
// (contract: src/limbo/model/multi_gp.hpp:56-397).  The reference runs every member as a TBB task
// (tools::par::loop at :124,172,191,210,226,263); here every output GP owns a device handle and a
// HIP stream, so the same loop makes their kernels overlap on one MI355X (BASELINE config 4 within
// a GPU; across GPUs the outputs are sharded by limbo_amd/parallel.py).
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_MODEL_MULTI_GP_HPP
#define LIMBO_MODEL_MULTI_GP_HPP
#include <cassert>
#include <string>
#include <tuple>
#include <vector>
#include <limbo/mean/null_function.hpp>
#include <limbo/model/gp.hpp>
#include <limbo/tools/parallel.hpp>
namespace limbo {
    namespace model {
        template <typename Params, template <typename, typename, typename, typename> class GPClass, typename KernelFunction, typename MeanFunction, class HyperParamsOptimizer = limbo::model::gp::NoLFOpt<Params>>
        class MultiGP {
        public:
            using GP_t = GPClass<Params, KernelFunction, limbo::mean::NullFunction<Params>, limbo::model::gp::NoLFOpt<Params>>;

            MultiGP() : _dim_in(-1), _dim_out(-1) {}
            MultiGP(int dim_in, int dim_out) : _dim_in(dim_in), _dim_out(dim_out), _mean_function(dim_out) { _make_models(); }

            void compute(const std::vector<Eigen::VectorXd>& samples, const std::vector<Eigen::VectorXd>& observations, bool compute_kernel = true)
            {
                assert(samples.size() != 0 && samples.size() == observations.size());
                _dim_in = samples[0].size();
                if (_dim_out != (int)observations[0].size()) {
                    _dim_out = observations[0].size();
                    _mean_function = MeanFunction(_dim_out);
                }
                if ((int)_gp_models.size() != _dim_out)
                    _make_models();
                _observations = observations;
                _update_mean_observation();
                // each member GP sees obs - m(x) and has a null mean of its own (:104-121)
                std::vector<std::vector<Eigen::VectorXd>> obs(_dim_out);
                for (size_t j = 0; j < observations.size(); j++) {
                    Eigen::VectorXd mv = _mean_function(samples[j], *this);
                    for (int i = 0; i < _dim_out; i++)
                        obs[i].push_back(limbo::tools::make_vector(observations[j](i) - mv(i)));
                }
                // ingest per member (host work), then all factorisations as one batched launch sequence per device —
                // the tools::par::loop of multi_gp.hpp:124-126 with the device doing the looping
                limbo::tools::par::loop(0, _dim_out, [&](size_t i) { _gp_models[i].compute(samples, obs[i], false); });
                if (compute_kernel)
                    _compute_kernels();
            }

            void optimize_hyperparams() { _hp_optimize(*this); }
            const MeanFunction& mean_function() const { return _mean_function; }
            MeanFunction& mean_function() { return _mean_function; }

            void add_sample(const Eigen::VectorXd& sample, const Eigen::VectorXd& observation)
            {
                if (_gp_models.size() == 0) {
                    _dim_in = sample.size();
                    _dim_out = observation.size();
                    _mean_function = MeanFunction(_dim_out);
                    _make_models();
                }
                else {
                    assert((int)sample.size() == _dim_in && (int)observation.size() == _dim_out);
                }
                _observations.push_back(observation);
                _update_mean_observation();
                Eigen::VectorXd mv = _mean_function(sample, *this);
                limbo::tools::par::loop(0, _dim_out, [&](size_t i) { _gp_models[i].add_sample(sample, limbo::tools::make_vector(observation(i) - mv(i))); });
            }

            /// (mu, one sigma^2 per output)
            std::tuple<Eigen::VectorXd, Eigen::VectorXd> query(const Eigen::VectorXd& v) const
            {
                Eigen::VectorXd mu(_dim_out), sigma(_dim_out);
                Eigen::VectorXd mv = _mean_function(v, *this);
                limbo::tools::par::loop(0, _dim_out, [&](size_t i) {
                    Eigen::VectorXd tmp;
                    double s;
                    std::tie(tmp, s) = _gp_models[i].query(v);
                    sigma(i) = s;
                    mu(i) = tmp(0) + mv(i);
                });
                return std::make_tuple(mu, sigma);
            }
            Eigen::VectorXd mu(const Eigen::VectorXd& v) const
            {
                Eigen::VectorXd mu(_dim_out);
                Eigen::VectorXd mv = _mean_function(v, *this);
                limbo::tools::par::loop(0, _dim_out, [&](size_t i) { mu(i) = _gp_models[i].mu(v)(0) + mv(i); });
                return mu;
            }
            Eigen::VectorXd sigma(const Eigen::VectorXd& v) const
            {
                Eigen::VectorXd sigma(_dim_out);
                limbo::tools::par::loop(0, _dim_out, [&](size_t i) { sigma(i) = _gp_models[i].sigma(v); });
                return sigma;
            }

            int dim_in() const
            {
                assert(_dim_in != -1);
                return _dim_in;
            }
            int dim_out() const
            {
                assert(_dim_out != -1);
                return _dim_out;
            }
            int nb_samples() const { return _observations.size(); }

            void recompute(bool update_obs_mean = true, bool update_full_kernel = true)
            {
                if (_gp_models.size() == 0)
                    return;
                if (update_obs_mean) { // a new mean changes every member's observations: full compute
                    const std::vector<Eigen::VectorXd> samples = _gp_models[0].samples();
                    return compute(samples, std::vector<Eigen::VectorXd>(_observations), update_full_kernel);
                }
                if (update_full_kernel)
                    _compute_kernels();
                else
                    limbo::tools::par::loop(0, _dim_out, [&](size_t i) { _gp_models[i].recompute(false, false); });
            }

            const std::vector<Eigen::VectorXd>& samples() const
            {
                assert(_gp_models.size());
                return _gp_models[0].samples();
            }
            const std::vector<Eigen::VectorXd>& observations() const { return _observations; }
            Eigen::MatrixXd observations_matrix() const
            {
                Eigen::MatrixXd m(_observations.size(), _dim_out);
                for (size_t i = 0; i < _observations.size(); i++)
                    for (int p = 0; p < _dim_out; ++p)
                        m(i, p) = _observations[i](p);
                return m;
            }
            Eigen::VectorXd mean_observation() const
            {
                assert(_dim_out > 0);
                return _observations.size() > 0 ? _mean_observation : Eigen::VectorXd::Zero(_dim_out);
            }
            std::vector<GP_t> gp_models() const { return _gp_models; }
            std::vector<GP_t>& gp_models() { return _gp_models; }

            template <typename A>
            void save(const std::string& directory) const
            {
                A archive(directory);
                save(archive);
            }
            template <typename A>
            void save(const A& archive) const
            {
                Eigen::VectorXd dims(2);
                dims(0) = _dim_in;
                dims(1) = _dim_out;
                archive.save(dims, "dims");
                archive.save(_observations, "observations");
                if (_mean_function.h_params_size() > 0)
                    archive.save(_mean_function.h_params(), "mean_params");
                for (int i = 0; i < _dim_out; i++)
                    _gp_models[i].template save<A>(archive.directory() + "/gp_" + std::to_string(i));
            }
            template <typename A>
            void load(const std::string& directory, bool recompute = true)
            {
                A archive(directory);
                load(archive, recompute);
            }
            template <typename A>
            void load(const A& archive, bool recompute = true)
            {
                _observations.clear();
                archive.load(_observations, "observations");
                Eigen::VectorXd dims;
                archive.load(dims, "dims");
                _dim_in = static_cast<int>(dims(0));
                _dim_out = static_cast<int>(dims(1));
                _update_mean_observation();
                _mean_function = MeanFunction(_dim_out);
                if (_mean_function.h_params_size() > 0) {
                    Eigen::VectorXd h_params;
                    archive.load(h_params, "mean_params");
                    _mean_function.set_h_params(h_params);
                }
                _make_models();
                for (int i = 0; i < _dim_out; i++) // members are not recomputed on their own (:385-388)
                    _gp_models[i].template load<A>(archive.directory() + "/gp_" + std::to_string(i), false);
                if (recompute)
                    this->recompute(true, true);
            }

        protected:
            std::vector<GP_t> _gp_models;
            int _dim_in, _dim_out;
            HyperParamsOptimizer _hp_optimize;
            MeanFunction _mean_function;
            std::vector<Eigen::VectorXd> _observations;
            Eigen::VectorXd _mean_observation;

            void _compute_kernels()
            {
                std::vector<GP_t*> ptrs;
                for (auto& g : _gp_models)
                    ptrs.push_back(&g);
                GP_t::compute_full_kernels_batched(ptrs);
            }
            void _make_models()
            {
                _gp_models.clear();
                _gp_models.reserve(_dim_out);
                for (int i = 0; i < _dim_out; i++) {
                    _gp_models.emplace_back(_dim_in, 1);
                    // independent GPs: dealt over the node's MI355Xs (one device: all of them there) — the
                    // tools::par::loop of multi_gp.hpp:124-126 then drives one device per host thread
                    _gp_models.back().set_device(limbo_amd::deal_device<Params>((size_t)i));
                }
            }
            void _update_mean_observation()
            {
                _mean_observation = Eigen::VectorXd::Zero(_dim_out);
                for (auto& o : _observations)
                    for (int p = 0; p < _dim_out; ++p)
                        _mean_observation(p) += o(p) / static_cast<double>(_observations.size());
            }
        };
    } // namespace model
} // namespace limbo
#endif
