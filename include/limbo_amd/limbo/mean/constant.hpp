// limbo/mean/constant.hpp — a fixed constant mean (src/limbo/mean/constant.hpp:52-83)
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_AMD_MEAN_CONSTANT_HPP
#define LIMBO_AMD_MEAN_CONSTANT_HPP
// With limbo's own tree on the include path BEHIND this directory this file steps aside: limbo's
// <limbo/mean/constant.hpp> is the one that gets compiled — the policy's body IS its interface, there is nothing of the engine's in it.
// Stand-alone (no limbo tree: this repository's own tests on a box without the reference) the definition below provides the name.
#if defined(__has_include_next)
#if __has_include_next(<limbo/mean/constant.hpp>)
#define LIMBO_AMD_MEAN_CONSTANT_HPP_FORWARDED 1
#include_next <limbo/mean/constant.hpp>
#endif
#endif
#ifndef LIMBO_AMD_MEAN_CONSTANT_HPP_FORWARDED
#ifndef LIMBO_MEAN_CONSTANT_HPP
#define LIMBO_MEAN_CONSTANT_HPP
#include <limbo/mean/mean.hpp>
namespace limbo {
    namespace defaults {
        struct mean_constant {
            BO_PARAM(double, constant, 1);
        };
    } // namespace defaults
    namespace mean {
        template <typename Params>
        struct Constant : public BaseMean<Params> {
            Constant(size_t dim_out = 1) : _dim_out(dim_out), _constant(Params::mean_constant::constant()) {}
            template <typename GP>
            Eigen::VectorXd operator()(const Eigen::VectorXd&, const GP&) const { return Eigen::VectorXd::Constant(_dim_out, _constant); }
            template <typename GP>
            Eigen::MatrixXd grad(const Eigen::VectorXd&, const GP&) const { return Eigen::MatrixXd::Ones(_dim_out, 1); }
            size_t h_params_size() const { return 1; }
            Eigen::VectorXd h_params() const
            {
                Eigen::VectorXd p(1);
                p(0) = _constant;
                return p;
            }
            void set_h_params(const Eigen::VectorXd& p) { _constant = p(0); }

        protected:
            size_t _dim_out;
            double _constant;
        };
    } // namespace mean
} // namespace limbo
#endif
#endif // LIMBO_AMD_MEAN_CONSTANT_HPP_FORWARDED
#endif // LIMBO_AMD_MEAN_CONSTANT_HPP
