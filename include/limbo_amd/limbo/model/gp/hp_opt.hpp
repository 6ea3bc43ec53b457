// limbo/model/gp/hp_opt.hpp — base of the hyper-parameter optimisation policies
// (contract: src/limbo/model/gp/hp_opt.hpp:58-73: default/copy constructible, warns if never used)
#ifndef LIMBO_MODEL_GP_HP_OPT_HPP
#define LIMBO_MODEL_GP_HP_OPT_HPP
#include <iostream>
#include <limbo/opt/rprop.hpp>
namespace limbo {
    namespace model {
        namespace gp {
            template <typename Params, typename Optimizer = opt::Rprop<Params>>
            struct HPOpt {
            public:
                HPOpt() : _called(false) {}
                HPOpt(const HPOpt&) : _called(true) {} // copies never warn
                HPOpt& operator=(const HPOpt&) { return *this; }
                ~HPOpt()
                {
                    if (!_called)
                        std::cerr << "'HPOpt' was never called!" << std::endl;
                }

            protected:
                bool _called;
            };
        } // namespace gp
    } // namespace model
} // namespace limbo
#endif
