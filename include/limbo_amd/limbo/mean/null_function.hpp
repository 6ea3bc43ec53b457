// limbo/mean/null_function.hpp — zero mean (src/limbo/mean/null_function.hpp:52-66)
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_AMD_MEAN_NULL_FUNCTION_HPP
#define LIMBO_AMD_MEAN_NULL_FUNCTION_HPP
// With limbo's own tree on the include path BEHIND this directory this file steps aside: limbo's
// <limbo/mean/null_function.hpp> is the one that gets compiled — the policy's body IS its interface, there is nothing of the engine's in it.
// Stand-alone (no limbo tree: this repository's own tests on a box without the reference) the definition below provides the name.
#if defined(__has_include_next)
#if __has_include_next(<limbo/mean/null_function.hpp>)
#define LIMBO_AMD_MEAN_NULL_FUNCTION_HPP_FORWARDED 1
#include_next <limbo/mean/null_function.hpp>
#endif
#endif
#ifndef LIMBO_AMD_MEAN_NULL_FUNCTION_HPP_FORWARDED
#ifndef LIMBO_MEAN_NULL_FUNCTION_HPP
#define LIMBO_MEAN_NULL_FUNCTION_HPP
#include <limbo/mean/mean.hpp>
namespace limbo {
    namespace mean {
        template <typename Params>
        struct NullFunction : public BaseMean<Params> {
            NullFunction(size_t dim_out = 1) : _dim_out(dim_out) {}
            template <typename GP>
            Eigen::VectorXd operator()(const Eigen::VectorXd&, const GP&) const { return Eigen::VectorXd::Zero(_dim_out); }

        protected:
            size_t _dim_out;
        };
    } // namespace mean
} // namespace limbo
#endif
#endif // LIMBO_AMD_MEAN_NULL_FUNCTION_HPP_FORWARDED
#endif // LIMBO_AMD_MEAN_NULL_FUNCTION_HPP
