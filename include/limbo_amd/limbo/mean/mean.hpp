// limbo/mean/mean.hpp — base of the mean functors (contract: src/limbo/mean/mean.hpp:61-77).
// Mean functors receive the GP itself and are evaluated on the HOST (gp.hpp:537-548): only
// obs_mean = Y - m(X) crosses to the device.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_AMD_MEAN_MEAN_HPP
#define LIMBO_AMD_MEAN_MEAN_HPP
// With limbo's own tree on the include path BEHIND this directory this file steps aside: limbo's
// <limbo/mean/mean.hpp> is the one that gets compiled — the policy's body IS its interface, there is nothing of the engine's in it.
// Stand-alone (no limbo tree: this repository's own tests on a box without the reference) the definition below provides the name.
#if defined(__has_include_next)
#if __has_include_next(<limbo/mean/mean.hpp>)
#define LIMBO_AMD_MEAN_MEAN_HPP_FORWARDED 1
#include_next <limbo/mean/mean.hpp>
#endif
#endif
#ifndef LIMBO_AMD_MEAN_MEAN_HPP_FORWARDED
#ifndef LIMBO_MEAN_MEAN_HPP
#define LIMBO_MEAN_MEAN_HPP
#include <Eigen/Core>
#include <cassert>
#include <limbo/tools/macros.hpp>
namespace limbo {
    namespace mean {
        template <typename Params>
        struct BaseMean {
            BaseMean(size_t /*dim_out*/ = 1) {}
            size_t h_params_size() const { return 0; }
            Eigen::VectorXd h_params() const { return Eigen::VectorXd(); }
            void set_h_params(const Eigen::VectorXd&) {}
            template <typename GP>
            Eigen::MatrixXd grad(const Eigen::VectorXd&, const GP&) const
            {
                assert(false && "this mean function has no hyper-parameters");
                return Eigen::MatrixXd();
            }
        };
    } // namespace mean
} // namespace limbo
#endif
#endif // LIMBO_AMD_MEAN_MEAN_HPP_FORWARDED
#endif // LIMBO_AMD_MEAN_MEAN_HPP
