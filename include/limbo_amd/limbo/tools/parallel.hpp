// limbo/tools/parallel.hpp — the task-parallel helpers limbo's model layer is written against
// (contract: src/limbo/tools/parallel.hpp:116-229, TBB or serial there).  Here a task is normally
// "drive one device GP": host threads only issue launches on independent HIP streams, so plain
// std::thread workers are enough — the parallelism that matters happens on the MI355X.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_TOOLS_PARALLEL_HPP
#define LIMBO_TOOLS_PARALLEL_HPP
#include <algorithm>
#include <cstddef>
#include <exception>
#include <thread>
#include <vector>
namespace limbo {
    namespace tools {
        namespace par {
            /// the container alias and conversion limbo's callers use (tools/parallel.hpp:77-113)
            template <typename X>
            using vector = std::vector<X>;
            template <typename V>
            inline V convert_vector(const V& v) { return v; }

            inline void init(int threads = -1) { (void)threads; }

            /// f(i) for i in [begin, end), one host thread per index (bounded by hardware_concurrency)
            template <typename F>
            inline void loop(size_t begin, size_t end, const F& f)
            {
                const size_t n = end > begin ? end - begin : 0;
                if (n <= 1) {
                    for (size_t i = begin; i < end; ++i)
                        f(i);
                    return;
                }
                const size_t nt = std::min<size_t>(n, std::max(1u, std::thread::hardware_concurrency()));
                // an exception thrown by f (the device GP throws std::runtime_error on any engine / HIP error) travels
                // to the caller as it would out of tbb::parallel_for: first one wins, the other workers finish
                std::vector<std::thread> th;
                std::vector<std::exception_ptr> err(nt);
                for (size_t t = 0; t < nt; ++t)
                    th.emplace_back([&, t]() {
                        try {
                            for (size_t i = begin + t; i < end; i += nt)
                                f(i);
                        }
                        catch (...) {
                            err[t] = std::current_exception();
                        }
                    });
                for (auto& x : th)
                    x.join();
                for (auto& e : err)
                    if (e)
                        std::rethrow_exception(e);
            }

            /// f(*it) for every element (tools/parallel.hpp:153-163)
            template <typename Iterator, typename F>
            inline void for_each(Iterator begin, Iterator end, const F& f)
            {
                std::vector<Iterator> its;
                for (Iterator i = begin; i != end; ++i)
                    its.push_back(i);
                loop(0, its.size(), [&](size_t i) { f(*its[i]); });
            }

            /// sort (tools/parallel.hpp:193-203) — host data, plain std::sort
            template <typename T1, typename T2, typename T3>
            inline void sort(T1 i1, T2 i2, T3 comp) { std::sort(i1, i2, comp); }

            /// f() nb times (tools/parallel.hpp:205-221)
            template <typename F>
            inline void replicate(size_t nb, const F& f)
            {
                loop(0, nb, [&](size_t) { f(); });
            }

            /// max over body(i), i in [0, num_steps), starting from init (the reduce of :169-191)
            template <typename T, typename F, typename C>
            inline T max(const T& init, int num_steps, const F& body, const C& comp)
            {
                std::vector<T> res((size_t)num_steps, init);
                loop(0, (size_t)num_steps, [&](size_t i) { res[i] = body((int)i); });
                T best = init;
                for (auto& r : res)
                    if (comp(r, best))
                        best = r;
                return best;
            }
        } // namespace par
    } // namespace tools
} // namespace limbo
#endif
