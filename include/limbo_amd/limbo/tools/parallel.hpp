// limbo/tools/parallel.hpp — the task-parallel helpers limbo's model layer is written against
// (contract: src/limbo/tools/parallel.hpp:116-229, TBB or serial there).  Here a task is normally
// "drive one device GP": host threads only issue launches on independent HIP streams, so plain
// std::thread workers are enough — the parallelism that matters happens on the MI355X.
#ifndef LIMBO_TOOLS_PARALLEL_HPP
#define LIMBO_TOOLS_PARALLEL_HPP
#include <algorithm>
#include <cstddef>
#include <thread>
#include <vector>
namespace limbo {
    namespace tools {
        namespace par {
            inline void init() {}

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
                std::vector<std::thread> th;
                for (size_t t = 0; t < nt; ++t)
                    th.emplace_back([&, t]() {
                        for (size_t i = begin + t; i < end; i += nt)
                            f(i);
                    });
                for (auto& x : th)
                    x.join();
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
