// limbo/tools/macros.hpp — the compile-time parameter system of limbo's policy API.
// These macro NAMES and their expansion contract are the configuration interface every user
// `Params` struct is written against (reference: src/limbo/tools/macros.hpp:53-110), so they are
// kept verbatim-compatible: BO_PARAM(T, name, v) yields `static constexpr T name()`, BO_DYN_PARAM
// a runtime-settable static, and so on.  New MI355X knobs live in limbo::defaults::gpu below.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_AMD_TOOLS_MACROS_HPP
#define LIMBO_AMD_TOOLS_MACROS_HPP
// With limbo's own tree on the include path BEHIND this directory, this file steps aside: limbo's
// <limbo/tools/macros.hpp> is the one that gets compiled, so every other limbo header keeps seeing exactly what it was
// written against.  Stand-alone (no limbo tree), the definitions below provide the same names.
#if defined(__has_include_next)
#if __has_include_next(<limbo/tools/macros.hpp>)
#define LIMBO_AMD_TOOLS_MACROS_HPP_FORWARDED 1
#include_next <limbo/tools/macros.hpp>
#endif
#endif
#ifndef LIMBO_AMD_TOOLS_MACROS_HPP_FORWARDED
#ifndef LIMBO_TOOLS_MACROS_HPP
#define LIMBO_TOOLS_MACROS_HPP

#include <Eigen/Core>
#include <cassert>
#include <cstddef>
#include <initializer_list>

#define BO_PARAM(Type, Name, Value) \
    static constexpr Type Name() { return Value; }

#define BO_DYN_PARAM(Type, Name)                 \
    static Type _##Name;                         \
    static Type Name() { return _##Name; }       \
    static void set_##Name(const Type& v) { _##Name = v; }

#define BO_DECLARE_DYN_PARAM(Type, Namespace, Name) Type Namespace::_##Name;

#define BO_PARAM_STRING(Name, Value) \
    static constexpr const char* Name() { return Value; }

// BO_PARAM_ARRAY(double, name, 1., 2., 3.) -> name(i), name_size(), name_t
#define BO_PARAM_ARRAY(Type, Name, ...)                                   \
    static Type Name(size_t i)                                            \
    {                                                                     \
        static constexpr Type _##Name[] = {__VA_ARGS__};                  \
        assert(i < sizeof(_##Name) / sizeof(_##Name[0]));                 \
        return _##Name[i];                                                \
    }                                                                     \
    static constexpr size_t Name##_size()                                 \
    {                                                                     \
        return std::initializer_list<Type>{__VA_ARGS__}.size();           \
    }                                                                     \
    using Name##_t = Type;

// BO_PARAM_VECTOR(double, name, 1., 2.) -> Eigen::VectorXd name()
#define BO_PARAM_VECTOR(Type, Name, ...)                                  \
    static const Eigen::VectorXd Name()                                   \
    {                                                                     \
        static constexpr Type _##Name[] = {__VA_ARGS__};                  \
        constexpr size_t n_ = sizeof(_##Name) / sizeof(_##Name[0]);       \
        Eigen::VectorXd v_(n_);                                           \
        for (size_t i_ = 0; i_ < n_; ++i_)                                \
            v_(i_) = _##Name[i_];                                         \
        return v_;                                                        \
    }
#endif // LIMBO_TOOLS_MACROS_HPP
#endif // stand-alone
namespace limbo {
    namespace defaults {
        /// placement of the model on the node's MI355Xs (not in the reference).  device >= 0: that HIP device;
        /// device = -1 (default): device 0 for a GP the user creates, and the CLONES that the parallel policies make
        /// (opt::ParallelRepeater restarts, model::MultiGP outputs, multi_gp::ParallelLFOpt fits) are dealt round-robin
        /// over the visible devices — with one visible device everything stays on it.
        struct gpu {
            BO_PARAM(int, device, -1);
        };
    } // namespace defaults
} // namespace limbo
#endif
