// limbo/model/gp/host_small.hpp — the exact-GP linear algebra for SMALL models on the host (an addition; not in limbo).
//
// Below Params::gpu::min_n_for_gpu() samples a GP is cheaper to keep on the host than on the device: one add_sample() or one
// single-point query() is O(n^2) flops — 11-18 us on one core at n = 200 — while ANY per-call device path pays a kernel launch
// and a PCIe round trip first (measured: 24-28 us through the one-launch kernels of csrc/small.hip).  limbo's
// stock acquisition optimisers call query() thousands of times per iteration on models of 10..200 samples
// (src/limbo/bayes_opt/boptimizer.hpp:148-161, src/benchmarks/limbo/bench.cpp:66-84), and its own test asserts that an
// incremental update beats a full recompute at n = 100 (src/tests/test_gp.cpp:625-628).
//
// What is here is what model::GP needs in that regime and nothing else, written against raw column-major storage (so that
// it compiles with Eigen3 and with the minimal stand-in alike):
//   llt_lower        K = L L^T in place (right-looking, column by column: contiguous axpys)           gp.hpp:565
//   append_row       the new row of L for one more sample (forward substitution + sqrt)                 gp.hpp:591-597
//   solve_lower / solve_lower_t   L y = b, L^T x = y for P right-hand sides                             gp.hpp:608-610, :620
// Everything above the threshold, every batched query, K^-1, the gradients and the LOO objectives stay on the device.
#ifndef LIMBO_MODEL_GP_HOST_SMALL_HPP
#define LIMBO_MODEL_GP_HOST_SMALL_HPP
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <limbo/tools/macros.hpp>

namespace limbo {
    namespace defaults {
        /// samples below which a model::GP keeps its factor on the host (0: always on the device).  256 = where one core's
        /// O(n^2) call meets the device's launch + PCIe floor, and the limit of the device's one-launch small path.
        struct gpu_host {
            BO_PARAM(int, min_n_for_gpu, 256);
        };
    } // namespace defaults
} // namespace limbo

namespace limbo_amd {
    /// Params::gpu::min_n_for_gpu() when the user's Params has it, defaults::gpu_host::min_n_for_gpu() otherwise;
    /// LIMBO_AMD_MIN_N_FOR_GPU in the environment overrides both (tests, benchmarks)
    template <typename Params, typename = void>
    struct param_min_n {
        static int get() { return limbo::defaults::gpu_host::min_n_for_gpu(); }
    };
    template <typename Params>
    struct param_min_n<Params, decltype((void)Params::gpu::min_n_for_gpu())> {
        static int get() { return Params::gpu::min_n_for_gpu(); }
    };
    template <typename Params>
    inline int min_n_for_gpu()
    {
        static const int env = [] {
            const char* e = std::getenv("LIMBO_AMD_MIN_N_FOR_GPU");
            return e ? std::atoi(e) : -1;
        }();
        return env >= 0 ? env : param_min_n<Params>::get();
    }

    /// query_batch() on a host-resident model goes to the device from this many (points x samples) on: below it the
    /// per-point host loop is cheaper than refreshing the device copy and a launch (LIMBO_AMD_HOST_BATCH_CROSSOVER)
    inline long host_batch_crossover()
    {
        static const long v = [] {
            const char* e = std::getenv("LIMBO_AMD_HOST_BATCH_CROSSOVER");
            return e ? std::atol(e) : 4096L;
        }();
        return v;
    }

    /// query_batch() of a device-resident model is dealt over ALL visible devices (one replica each, a host thread per
    /// device) from this many points on (LIMBO_AMD_MULTI_DEVICE_QUERY_MIN; 0: never) — the reference's parallel query
    /// (multi_gp.hpp:191-195, tools/parallel.hpp:138-201) with GPUs in place of cores
    inline long multi_device_query_min()
    {
        static const long v = [] {
            const char* e = std::getenv("LIMBO_AMD_MULTI_DEVICE_QUERY_MIN");
            return e ? std::atol(e) : 16384L;
        }();
        return v;
    }

    namespace host_small {
        /// A (n x n, column-major, leading dimension lda): lower triangle in, L out; the strict upper triangle is zeroed
        /// (gp.hpp:565: `Eigen::LLT<MatrixXd>(K).matrixL()` is dense with a zero upper part).  Returns 0, or the 1-based
        /// index of the first non-positive pivot (NaNs propagate from there on, as they do in the reference).
        inline int llt_lower(double* A, int64_t n, int64_t lda)
        {
            int bad = 0;
            for (int64_t j = 0; j < n; ++j) {
                double* cj = A + j * lda;
                const double d = cj[j];
                if (!(d > 0.0) && !bad)
                    bad = (int)(j + 1);
                const double l = std::sqrt(d), inv = 1.0 / l;
                cj[j] = l;
                for (int64_t i = j + 1; i < n; ++i)
                    cj[i] *= inv;
                for (int64_t k = j + 1; k < n; ++k) { // trailing columns: a_ik -= l_ij l_kj, i >= k
                    double* ck = A + k * lda;
                    const double lkj = cj[k];
                    for (int64_t i = k; i < n; ++i)
                        ck[i] -= cj[i] * lkj;
                }
                for (int64_t i = 0; i < j; ++i)
                    cj[i] = 0.0;
            }
            return bad;
        }
        /// L y = b in place for P right-hand sides (B: n x P, column-major, ldb)
        inline void solve_lower(const double* L, int64_t n, int64_t ldl, double* B, int64_t P, int64_t ldb)
        {
            for (int64_t p = 0; p < P; ++p) {
                double* b = B + p * ldb;
                for (int64_t j = 0; j < n; ++j) {
                    const double* cj = L + j * ldl;
                    const double y = b[j] / cj[j];
                    b[j] = y;
                    for (int64_t i = j + 1; i < n; ++i)
                        b[i] -= cj[i] * y;
                }
            }
        }
        /// L^T x = y in place
        inline void solve_lower_t(const double* L, int64_t n, int64_t ldl, double* B, int64_t P, int64_t ldb)
        {
            for (int64_t p = 0; p < P; ++p) {
                double* b = B + p * ldb;
                for (int64_t j = n - 1; j >= 0; --j) {
                    const double* cj = L + j * ldl;
                    double s = b[j];
                    for (int64_t i = j + 1; i < n; ++i)
                        s -= cj[i] * b[i];
                    b[j] = s / cj[j];
                }
            }
        }
        /// The row that one more sample appends to L (gp.hpp:591-597): kcol[0..n) = k(x_i, x_new), knn = k(x_new, x_new) +
        /// noise + 1e-8.  row[0..n) = L^-1 kcol, row[n] = sqrt(knn - |row|^2).  Returns 0, or n + 1 for a non-positive pivot.
        inline int append_row(const double* L, int64_t n, int64_t ldl, const double* kcol, double knn, double* row)
        {
            for (int64_t i = 0; i < n; ++i)
                row[i] = kcol[i];
            for (int64_t j = 0; j < n; ++j) {
                const double* cj = L + j * ldl;
                const double y = row[j] / cj[j];
                row[j] = y;
                for (int64_t i = j + 1; i < n; ++i)
                    row[i] -= cj[i] * y;
            }
            double s = 0.0;
            for (int64_t i = 0; i < n; ++i)
                s += row[i] * row[i];
            const double d = knn - s;
            row[n] = std::sqrt(d);
            return d > 0.0 ? 0 : (int)(n + 1);
        }
    } // namespace host_small
} // namespace limbo_amd
#endif
