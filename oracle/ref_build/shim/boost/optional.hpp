// boost::optional stand-in over std::optional (opt::eval_t, src/limbo/opt/optimizer.hpp:61) — TEST INFRASTRUCTURE.
#ifndef REFSHIM_BOOST_OPTIONAL
#define REFSHIM_BOOST_OPTIONAL
#include <optional>
namespace boost {
    template <class T> class optional {
        std::optional<T> o_;

    public:
        optional() {}
        optional(const T& v) : o_(v) {}
        optional(T&& v) : o_(std::move(v)) {}
        // anything T can be built from (e.g. a matrix expression for a VectorXd)
        template <class U, typename = typename std::enable_if<std::is_constructible<T, const U&>::value && !std::is_same<typename std::decay<U>::type, optional>::value>::type>
        optional(const U& u) : o_(T(u)) {}
        bool is_initialized() const { return o_.has_value(); }
        explicit operator bool() const { return o_.has_value(); }
        const T& get() const { return *o_; }
        T& get() { return *o_; }
        const T& operator*() const { return *o_; }
        T& operator*() { return *o_; }
    };
} // namespace boost
#endif
