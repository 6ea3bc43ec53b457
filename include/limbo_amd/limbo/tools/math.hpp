// limbo/tools/math.hpp — scalar helpers used by the optimisers (reference: src/limbo/tools/math.hpp)
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_AMD_TOOLS_MATH_HPP
#define LIMBO_AMD_TOOLS_MATH_HPP
// With limbo's own tree on the include path BEHIND this directory, this file steps aside: limbo's
// <limbo/tools/math.hpp> is the one that gets compiled, so every other limbo header keeps seeing exactly what it was
// written against.  Stand-alone (no limbo tree), the definitions below provide the same names.
#if defined(__has_include_next)
#if __has_include_next(<limbo/tools/math.hpp>)
#define LIMBO_AMD_TOOLS_MATH_HPP_FORWARDED 1
#include_next <limbo/tools/math.hpp>
#endif
#endif
#ifndef LIMBO_AMD_TOOLS_MATH_HPP_FORWARDED
#ifndef LIMBO_TOOLS_MATH_HPP
#define LIMBO_TOOLS_MATH_HPP
#include <Eigen/Core>
#include <cmath>
#include <type_traits>
namespace limbo {
    namespace tools {
        /// sign of x as -1, 0 or +1
        template <typename T>
        inline constexpr int signum(T x) { return (T(0) < x) - (x < T(0)); }

        /// a 1-D vector holding x
        inline Eigen::VectorXd make_vector(double x)
        {
            Eigen::VectorXd v(1);
            v(0) = x;
            return v;
        }

        /// true if v is nan or +-inf (scalars), or holds such a coefficient (vectors)
        template <typename T, typename std::enable_if<std::is_arithmetic<T>::value, int>::type = 0>
        inline bool is_nan_or_inf(T v) { return std::isinf(v) || std::isnan(v); }
        template <typename T, typename std::enable_if<!std::is_arithmetic<T>::value, int>::type = 0>
        inline bool is_nan_or_inf(const T& v)
        {
            for (int i = 0; i < (int)v.size(); ++i)
                if (std::isinf(v(i)) || std::isnan(v(i)))
                    return true;
            return false;
        }
    } // namespace tools
} // namespace limbo
#endif // LIMBO_TOOLS_MATH_HPP
#endif // stand-alone
#endif
