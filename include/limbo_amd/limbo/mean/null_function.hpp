// limbo/mean/null_function.hpp — zero mean (src/limbo/mean/null_function.hpp:52-66)
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
