// limbo/opt/batch_search.hpp — batch-aware acquisition optimisers.
//
// limbo's inner optimisers see the acquisition function as a per-point functor
// F: (VectorXd, bool) -> eval_t (opt/optimizer.hpp:61-96), and opt::GridSearch evaluates it one
// point at a time (opt/grid_search.hpp:84-112); every evaluation is a GP::query, i.e. one N^2
// triangular solve (gp.hpp:618-624).  On the device the economical unit is the batch: M points =
// one cross-kernel build + one (N x M) MFMA triangular solve (gpe_query_batch).  The optimisers here
// keep limbo's optimiser signature — operator()(f, init, bounded) — and use f.batch(points) when
// the functor offers it (opt::make_batch_objective wraps an acquisition object, acqui/*.hpp
// batch()); with a plain functor they fall back to per-point calls and still return the same point.
//
//   BatchGridSearch   : the exact grid and tie-breaking of opt::GridSearch, one batch.
//   BatchRandomSearch : `points` uniform samples, then `refine_rounds` rounds of `points` samples in
//                       a box shrinking around the incumbent; every round is one batch.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_OPT_BATCH_SEARCH_HPP
#define LIMBO_OPT_BATCH_SEARCH_HPP
#include <algorithm>
#include <limits>
#include <random>
#include <type_traits>
#include <vector>
#include <limbo/opt/optimizer.hpp>
#include <limbo/tools/macros.hpp>
#if defined(__has_include)
#if __has_include(<limbo/opt/grid_search.hpp>)
#include <limbo/opt/grid_search.hpp> // limbo's tree is on the include path: its defaults::opt_gridsearch is THE definition
#endif
#endif
namespace limbo {
    namespace defaults {
#ifndef LIMBO_OPT_GRID_SEARCH_HPP // limbo's own opt/grid_search.hpp (included above when its tree is there) defines it
        struct opt_gridsearch {
            /// number of bins for each dimension (opt/grid_search.hpp:57-61)
            BO_PARAM(int, bins, 5);
        };
#endif
        struct opt_batchrandomsearch {
            BO_PARAM(int, points, 8192);
            BO_PARAM(int, refine_rounds, 2);
            BO_PARAM(double, shrink, 0.2); // half-width of the refinement box: shrink^round
            BO_PARAM(int, seed, -1); // < 0: std::random_device
        };
    } // namespace defaults
    namespace opt {
        /// acquisition object + aggregator as an optimiser objective that also evaluates batches
        /// (the per-point call is what bo_base's lambda does: bayes_opt/boptimizer.hpp:151-153)
        template <typename Acqui, typename Afun>
        struct BatchObjective {
            Acqui& acqui; // not const: acqui::EI caches f_max / nb_samples in operator() and batch() (acqui/ei.hpp:85-116)
            const Afun& afun;
            eval_t operator()(const Eigen::VectorXd& x, bool g) const { return acqui(x, afun, g); }
            std::vector<double> batch(const std::vector<Eigen::VectorXd>& pts) const { return acqui.batch(pts, afun); }
        };
        template <typename Acqui, typename Afun>
        BatchObjective<Acqui, Afun> make_batch_objective(Acqui& a, const Afun& f) { return BatchObjective<Acqui, Afun>{a, f}; }

        namespace detail {
            template <typename F>
            auto eval_many(const F& f, const std::vector<Eigen::VectorXd>& pts, int) -> decltype(f.batch(pts)) { return f.batch(pts); }
            template <typename F>
            std::vector<double> eval_many(const F& f, const std::vector<Eigen::VectorXd>& pts, long)
            {
                std::vector<double> v(pts.size());
                for (size_t i = 0; i < pts.size(); ++i)
                    v[i] = opt::eval(f, pts[i]);
                return v;
            }
            inline size_t first_argmax(const std::vector<double>& v)
            {
                size_t b = 0;
                for (size_t i = 1; i < v.size(); ++i)
                    if (v[i] > v[b])
                        b = i;
                return b;
            }
        } // namespace detail

        template <typename Params>
        struct BatchGridSearch {
            template <typename F>
            Eigen::VectorXd operator()(const F& f, const Eigen::VectorXd& init, bool bounded) const
            {
                assert(bounded); // grid_search.hpp:77
                (void)bounded;
                const size_t dim = init.size();
                // the coordinate list of grid_search.hpp:88-92, accumulated the same way
                const double step = 1.0 / (double)Params::opt_gridsearch::bins();
                std::vector<double> xs;
                for (double x = 0; x < 1.0 + step; x += step)
                    xs.push_back(x);
                size_t total = 1;
                for (size_t d = 0; d < dim; ++d)
                    total *= xs.size();
                // lexicographic order, dimension 0 slowest: the visiting order of the reference's
                // recursion, so "first strict maximum" picks the same point
                std::vector<Eigen::VectorXd> pts(total, Eigen::VectorXd(dim));
                for (size_t i = 0; i < total; ++i) {
                    size_t r = i;
                    for (size_t d = dim; d-- > 0;) {
                        pts[i](d) = xs[r % xs.size()];
                        r /= xs.size();
                    }
                }
                const std::vector<double> v = detail::eval_many(f, pts, 0);
                return pts[detail::first_argmax(v)];
            }
        };

        template <typename Params>
        struct BatchRandomSearch {
            template <typename F>
            Eigen::VectorXd operator()(const F& f, const Eigen::VectorXd& init, bool bounded) const
            {
                const int n = Params::opt_batchrandomsearch::points();
                const int rounds = Params::opt_batchrandomsearch::refine_rounds();
                const double shrink = Params::opt_batchrandomsearch::shrink();
                const size_t dim = init.size();
                const int seed = Params::opt_batchrandomsearch::seed();
                std::mt19937_64 g(seed < 0 ? std::random_device()() : (unsigned)seed);
                std::uniform_real_distribution<double> u01(0.0, 1.0);
                Eigen::VectorXd best = init;
                double best_val = -std::numeric_limits<double>::max();
                double half = bounded ? 0.5 : 1.0; // unbounded: a unit box around init, then shrinking
                Eigen::VectorXd centre = bounded ? Eigen::VectorXd::Constant(dim, 0.5) : init;
                for (int r = 0; r <= rounds; ++r) {
                    std::vector<Eigen::VectorXd> pts(n + 1, Eigen::VectorXd(dim));
                    pts[0] = best; // the incumbent (init in round 0) is always a candidate
                    for (int i = 1; i <= n; ++i)
                        for (size_t d = 0; d < dim; ++d) {
                            double x = centre(d) + (2.0 * u01(g) - 1.0) * half;
                            if (bounded)
                                x = std::min(1.0, std::max(0.0, x));
                            pts[i](d) = x;
                        }
                    const std::vector<double> v = detail::eval_many(f, pts, 0);
                    const size_t b = detail::first_argmax(v);
                    if (v[b] > best_val) {
                        best_val = v[b];
                        best = pts[b];
                    }
                    centre = best;
                    half = (r == 0 ? shrink : half * shrink);
                }
                return best;
            }
        };
    } // namespace opt
} // namespace limbo
#endif
