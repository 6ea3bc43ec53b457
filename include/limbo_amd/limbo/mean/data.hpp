// limbo/mean/data.hpp — constant mean equal to the mean of the observations (src/limbo/mean/data.hpp:55-64)
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
