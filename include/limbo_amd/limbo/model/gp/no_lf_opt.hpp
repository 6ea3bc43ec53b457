// limbo/model/gp/no_lf_opt.hpp — "do not optimise" policy (src/limbo/model/gp/no_lf_opt.hpp:55-66)
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
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
