// limbo/model/gp/no_lf_opt.hpp — "do not optimise" policy (src/limbo/model/gp/no_lf_opt.hpp:55-66)
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_AMD_MODEL_GP_NO_LF_OPT_HPP
#define LIMBO_AMD_MODEL_GP_NO_LF_OPT_HPP
// With limbo's own tree on the include path BEHIND this directory this file steps aside: limbo's
// <limbo/model/gp/no_lf_opt.hpp> is the one that gets compiled — the policy's body IS its interface, there is nothing of the engine's in it.
// Stand-alone (no limbo tree: this repository's own tests on a box without the reference) the definition below provides the name.
#if defined(__has_include_next)
#if __has_include_next(<limbo/model/gp/no_lf_opt.hpp>)
#define LIMBO_AMD_MODEL_GP_NO_LF_OPT_HPP_FORWARDED 1
#include_next <limbo/model/gp/no_lf_opt.hpp>
#endif
#endif
#ifndef LIMBO_AMD_MODEL_GP_NO_LF_OPT_HPP_FORWARDED
#ifndef LIMBO_MODEL_GP_NO_LF_OPT_HPP
#define LIMBO_MODEL_GP_NO_LF_OPT_HPP
#include <cassert>
#include <iostream>
namespace limbo {
    namespace model {
        namespace gp {
            template <typename Params>
            struct NoLFOpt {
                template <typename GP>
                void operator()(GP&) const
                {
                    std::cerr << "'NoLFOpt' should never be called!" << std::endl;
                    assert(false);
                }
            };
        } // namespace gp
    } // namespace model
} // namespace limbo
#endif
#endif // LIMBO_AMD_MODEL_GP_NO_LF_OPT_HPP_FORWARDED
#endif // LIMBO_AMD_MODEL_GP_NO_LF_OPT_HPP
