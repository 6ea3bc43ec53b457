// limbo/tools/math.hpp — scalar helpers used by the optimisers (reference: src/limbo/tools/math.hpp)
#ifndef LIMBO_TOOLS_MATH_HPP
#define LIMBO_TOOLS_MATH_HPP
#include <Eigen/Core>
#include <random>
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

        /// uniform random vector in [0, 1]^dim (bounded) or [-1, 1] scaled (unbounded is caller's business)
        inline Eigen::VectorXd random_vector(int dim, unsigned seed)
        {
            std::mt19937_64 g(seed);
            std::uniform_real_distribution<double> u(0.0, 1.0);
            Eigen::VectorXd v(dim);
            for (int i = 0; i < dim; ++i)
                v(i) = u(g);
            return v;
        }
    } // namespace tools
} // namespace limbo
#endif
