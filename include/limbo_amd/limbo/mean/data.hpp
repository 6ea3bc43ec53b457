// limbo/mean/data.hpp — constant mean equal to the mean of the observations (src/limbo/mean/data.hpp:55-64)
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_AMD_MEAN_DATA_HPP
#define LIMBO_AMD_MEAN_DATA_HPP
// With limbo's own tree on the include path BEHIND this directory this file steps aside: limbo's
// <limbo/mean/data.hpp> is the one that gets compiled — the policy's body IS its interface, there is nothing of the engine's in it.
// Stand-alone (no limbo tree: this repository's own tests on a box without the reference) the definition below provides the name.
#if defined(__has_include_next)
#if __has_include_next(<limbo/mean/data.hpp>)
#define LIMBO_AMD_MEAN_DATA_HPP_FORWARDED 1
#include_next <limbo/mean/data.hpp>
#endif
#endif
#ifndef LIMBO_AMD_MEAN_DATA_HPP_FORWARDED
#ifndef LIMBO_MEAN_DATA_HPP
#define LIMBO_MEAN_DATA_HPP
#include <limbo/mean/mean.hpp>
namespace limbo {
    namespace mean {
        template <typename Params>
        struct Data : public BaseMean<Params> {
            Data(size_t /*dim_out*/ = 1) {}
            template <typename GP>
            Eigen::VectorXd operator()(const Eigen::VectorXd&, const GP& gp) const { return gp.mean_observation(); }
        };
    } // namespace mean
} // namespace limbo
#endif
#endif // LIMBO_AMD_MEAN_DATA_HPP_FORWARDED
#endif // LIMBO_AMD_MEAN_DATA_HPP
