// limbo/opt/optimizer.hpp — the evaluation type shared by objective functors and optimisers
// (contract: src/limbo/opt/optimizer.hpp:61-96).  The reference spells the optional gradient
// boost::optional<Eigen::VectorXd>; boost is used when present, otherwise an equivalent minimal
// optional with the same accessors (is_initialized(), get(), operator bool).
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_AMD_OPT_OPTIMIZER_HPP
#define LIMBO_AMD_OPT_OPTIMIZER_HPP
// With limbo's own tree on the include path BEHIND this directory, this file steps aside: limbo's
// <limbo/opt/optimizer.hpp> is the one that gets compiled, so every other limbo header keeps seeing exactly what it was
// written against.  Stand-alone (no limbo tree), the definitions below provide the same names.
#if defined(__has_include_next)
#if __has_include_next(<limbo/opt/optimizer.hpp>)
#define LIMBO_AMD_OPT_OPTIMIZER_HPP_FORWARDED 1
#include_next <limbo/opt/optimizer.hpp>
#endif
#endif
#ifndef LIMBO_AMD_OPT_OPTIMIZER_HPP_FORWARDED
#ifndef LIMBO_OPT_OPTIMIZER_HPP
#define LIMBO_OPT_OPTIMIZER_HPP
#include <Eigen/Core>
#include <cassert>
#include <tuple>
#include <utility>
#if defined(__has_include)
#if __has_include(<boost/optional.hpp>)
#include <boost/optional.hpp>
#define LIMBO_AMD_HAVE_BOOST_OPTIONAL 1
#endif
#endif
namespace limbo {
    namespace opt {
#ifdef LIMBO_AMD_HAVE_BOOST_OPTIONAL
        using optional_grad_t = boost::optional<Eigen::VectorXd>;
#else
        class optional_grad_t {
        public:
            optional_grad_t() : _has(false) {}
            optional_grad_t(const Eigen::VectorXd& v) : _has(true), _v(v) {}
            bool is_initialized() const { return _has; }
            explicit operator bool() const { return _has; }
            const Eigen::VectorXd& get() const
            {
                assert(_has);
                return _v;
            }
            const Eigen::VectorXd& operator*() const { return get(); }

        private:
            bool _has;
            Eigen::VectorXd _v;
        };
#endif
        /// (value, optional gradient)
        using eval_t = std::pair<double, optional_grad_t>;
        inline eval_t no_grad(double x) { return eval_t{x, optional_grad_t{}}; }
        inline const Eigen::VectorXd& grad(const eval_t& fg)
        {
            assert(std::get<1>(fg).is_initialized());
            return std::get<1>(fg).get();
        }
        inline double fun(const eval_t& fg) { return std::get<0>(fg); }
        template <typename F>
        inline double eval(const F& f, const Eigen::VectorXd& x) { return std::get<0>(f(x, false)); }
        template <typename F>
        inline eval_t eval_grad(const F& f, const Eigen::VectorXd& x) { return f(x, true); }
    } // namespace opt
} // namespace limbo
#endif // LIMBO_OPT_OPTIMIZER_HPP
#endif // stand-alone
#endif
