// limbo/model/gp/no_lf_opt.hpp — "do not optimise" policy (src/limbo/model/gp/no_lf_opt.hpp:55-66)
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
