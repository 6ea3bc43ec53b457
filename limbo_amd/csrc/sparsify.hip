// sparsify.hip — density-based thinning of a sample set (gfx950).
//
// Replaces SparsifiedGP::_sparsify / _get_most_dense_point (src/limbo/model/sparsified_gp.hpp:124-183):
// while more than max_points samples remain, remove the sample whose D nearest remaining neighbours
// are closest in total (D = input dimension).  The reference rebuilds and partial_sorts every row of
// an N x N distance matrix for EVERY removal (O(N^2 log) each, N - max_points removals).
//
// Device form (HBM-bound selection work, no matrix cores):
//   * the N x N Euclidean distance matrix is built once (134 MB at N = 4096) and stays in HBM;
//   * every alive row caches (sum of its D smallest alive distances, the D-th smallest);
//   * removing sample k only invalidates rows i with dist(i, k) <= kth[i] — on average ~D rows —
//     so a removal re-scans those rows only (one wave per row: per-lane sorted top-D in registers,
//     then D rounds of wave-wide minimum extraction, summed in ascending order like the
//     reference's `dist += neighbors[j]`), followed by a single-workgroup arg-min over the cached sums;
//   * the removed index travels through device memory, so the whole loop is enqueued without a
//     host round trip (2 launches per removal).
// Ties: the lowest index wins (the reference's serial loop takes the first strict minimum,
// sparsified_gp.hpp:146-149; its TBB variant is order-dependent).
#include "dev.h"
#include "../../include/gpe.h"
#include <cmath>
#include <vector>

#define ST 64

// Dm[i + j*ld] = || x_i - x_j ||  (sparsified_gp.hpp:160-166), +inf on the diagonal (the reference
// erases the self distance, :134).  X row-major N x D.  No contraction: plain mul/add like the CPU.
__global__ __launch_bounds__(256) void k_dist_matrix(const double* __restrict__ X, int64_t N, int D,
                                                     double* __restrict__ Dm, int64_t ld)
{
    extern __shared__ double sm[]; // xi[64][D] | xj[64][D]
    double* xi = sm;
    double* xj = sm + ST * D;
    const int64_t i0 = (int64_t)blockIdx.x * ST, j0 = (int64_t)blockIdx.y * ST;
    for (int e = threadIdx.x; e < ST * D; e += 256) {
        const int r = e / D, d = e - r * D;
        xi[e] = (i0 + r < N) ? X[(i0 + r) * D + d] : 0.0;
        xj[e] = (j0 + r < N) ? X[(j0 + r) * D + d] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t i = i0 + tx;
    if (i >= N)
        return;
    for (int c = ty; c < ST; c += 4) {
        const int64_t j = j0 + c;
        if (j >= N)
            break;
        double s = 0.0;
        for (int d = 0; d < D; ++d) {
            const double q = __dsub_rn(xi[tx * D + d], xj[c * D + d]);
            s = __dadd_rn(s, __dmul_rn(q, q));
        }
        Dm[i + j * ld] = (i == j) ? INFINITY : sqrt(s);
    }
}

struct SparseState {
    int last_removed; // -1 before the first removal: every row is computed
    int n_alive;
};

static __device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double t = __shfl_xor(v, o);
        v = (t < v) ? t : v;
    }
    return v;
}

// one wave per row: (re)compute sum[i] and kth[i] when the last removal touched the row's top-D set
template <int DM>
__global__ __launch_bounds__(256) void k_row_density(const double* __restrict__ Dm, int64_t ld, int64_t N, int D,
                                                     const unsigned char* __restrict__ alive,
                                                     const SparseState* __restrict__ st, double* __restrict__ sum,
                                                     double* __restrict__ kth)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N || !alive[i])
        return;
    const int kr = st->last_removed;
    if (kr >= 0 && !(Dm[i + (int64_t)kr * ld] <= kth[i]))
        return; // the removed sample was not among this row's D nearest: cached values stand
    double top[DM]; // ascending; this lane's DM smallest alive distances
#pragma unroll
    for (int q = 0; q < DM; ++q)
        top[q] = INFINITY;
    // four independent loads in flight per lane (the scan of one row is a single wave: latency-bound)
    for (int64_t j0 = lane; j0 < N; j0 += 256) {
        double vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = j0 + 64 * u;
            const int64_t jc = j < N ? j : N - 1;
            const double d = Dm[jc + i * ld]; // symmetric: column i is contiguous
            vv[u] = (j < N && alive[jc]) ? d : INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double v = vv[u];
            if (v < top[DM - 1]) {
#pragma unroll
                for (int q = 0; q < DM; ++q) { // insertion keeping the order
                    const bool sw = v < top[q];
                    const double t = top[q];
                    top[q] = sw ? v : t;
                    v = sw ? t : v;
                }
            }
        }
    }
    double s = 0.0, last = INFINITY;
    for (int r = 0; r < D; ++r) {
        const double m = wave_min(top[0]);
        // the lowest lane holding m gives it up
        const unsigned long long holders = __ballot(top[0] == m);
        const int owner = __ffsll((long long)holders) - 1;
        if (lane == owner) {
#pragma unroll
            for (int q = 0; q + 1 < DM; ++q)
                top[q] = top[q + 1];
            top[DM - 1] = INFINITY;
        }
        s = __dadd_rn(s, m); // ascending order, as `dist += neighbors[j]` (:141-143)
        last = m;
    }
    if (lane == 0) {
        sum[i] = s;
        kth[i] = last;
    }
}

// arg-min of sum over alive rows (lowest index on ties), remove it
__global__ __launch_bounds__(1024) void k_remove_densest(const double* __restrict__ sum, int64_t N,
                                                         unsigned char* __restrict__ alive, SparseState* __restrict__ st)
{
    __shared__ double sv[16];
    __shared__ int si[16];
    double bv = INFINITY;
    int bi = 0x7fffffff;
    for (int64_t i = threadIdx.x; i < N; i += 1024)
        if (alive[i]) {
            const double v = sum[i];
            if (v < bv || (v == bv && (int)i < bi)) {
                bv = v;
                bi = (int)i;
            }
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double tv = __shfl_xor(bv, o);
        const int ti = __shfl_xor(bi, o);
        if (tv < bv || (tv == bv && ti < bi)) {
            bv = tv;
            bi = ti;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = bv;
        si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] < bv || (sv[w] == bv && si[w] < bi)) {
                bv = sv[w];
                bi = si[w];
            }
        if (bi != 0x7fffffff) { // sanity check of sparsified_gp.hpp:172-173
            alive[bi] = 0;
            st->last_removed = bi;
            st->n_alive -= 1;
        }
    }
}

extern "C" int gpe_sparsify(int device_id, const double* X, int64_t N, int D, int64_t max_points, int64_t* keep,
                            int64_t* n_keep)
{
    if (!X || !keep || !n_keep || N <= 0 || D <= 0 || max_points <= 0)
        return GPE_ERR_ARG;
    if (N <= max_points) { // sparsified_gp.hpp:88-89: nothing to do
        for (int64_t i = 0; i < N; ++i)
            keep[i] = i;
        *n_keep = N;
        return GPE_OK;
    }
    if (D > 64 || max_points <= D) // the D nearest neighbours must exist among the remaining samples
        return GPE_ERR_UNSUPPORTED;
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device_id) != hipSuccess)
        return GPE_ERR_HIP;
    hipStream_t s = nullptr;
    double *dX = nullptr, *dD = nullptr, *dSum = nullptr, *dKth = nullptr;
    unsigned char* dAlive = nullptr;
    SparseState* dSt = nullptr;
    int rc = GPE_OK;
    const int64_t ld = (N + 15) / 16 * 16 + 16; // off the power-of-two stride
    std::vector<unsigned char> alive((size_t)N, 1);
#define SCHK(x)                 \
    do {                        \
        if ((x) != hipSuccess) { \
            rc = GPE_ERR_HIP;   \
            goto done;          \
        }                       \
    } while (0)
    SCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    SCHK(hipMalloc(&dX, sizeof(double) * (size_t)(N * D)));
    SCHK(hipMalloc(&dD, sizeof(double) * (size_t)(ld * N)));
    SCHK(hipMalloc(&dSum, sizeof(double) * (size_t)N));
    SCHK(hipMalloc(&dKth, sizeof(double) * (size_t)N));
    SCHK(hipMalloc(&dAlive, (size_t)N));
    SCHK(hipMalloc(&dSt, sizeof(SparseState)));
    {
        SparseState h{-1, (int)N};
        SCHK(hipMemcpyAsync(dX, X, sizeof(double) * (size_t)(N * D), hipMemcpyHostToDevice, s));
        SCHK(hipMemcpyAsync(dAlive, alive.data(), (size_t)N, hipMemcpyHostToDevice, s));
        SCHK(hipMemcpyAsync(dSt, &h, sizeof(h), hipMemcpyHostToDevice, s));
        const unsigned nt = (unsigned)((N + ST - 1) / ST);
        GPE_LAUNCH(k_dist_matrix, dim3(nt, nt), dim3(256), sizeof(double) * 2 * ST * D, s, dX, N, D, dD, ld);
        const dim3 rg((unsigned)((N + 3) / 4));
        for (int64_t it = 0; it < N - max_points; ++it) {
            if (D <= 8)
                GPE_LAUNCH((k_row_density<8>), rg, dim3(256), 0, s, dD, ld, N, D, dAlive, dSt, dSum, dKth);
            else if (D <= 16)
                GPE_LAUNCH((k_row_density<16>), rg, dim3(256), 0, s, dD, ld, N, D, dAlive, dSt, dSum, dKth);
            else if (D <= 32)
                GPE_LAUNCH((k_row_density<32>), rg, dim3(256), 0, s, dD, ld, N, D, dAlive, dSt, dSum, dKth);
            else
                GPE_LAUNCH((k_row_density<64>), rg, dim3(256), 0, s, dD, ld, N, D, dAlive, dSt, dSum, dKth);
            GPE_LAUNCH(k_remove_densest, dim3(1), dim3(1024), 0, s, dSum, N, dAlive, dSt);
        }
        SCHK(hipMemcpyAsync(alive.data(), dAlive, (size_t)N, hipMemcpyDeviceToHost, s));
        SCHK(hipStreamSynchronize(s));
        SCHK(hipGetLastError());
        int64_t n = 0;
        for (int64_t i = 0; i < N; ++i)
            if (alive[i])
                keep[n++] = i; // order preserved, as vector::erase does (:174-175)
        *n_keep = n;
    }
done:
#undef SCHK
    for (void* p : {(void*)dX, (void*)dD, (void*)dSum, (void*)dKth, (void*)dAlive, (void*)dSt})
        if (p)
            hipFree(p);
    if (s)
        hipStreamDestroy(s);
    hipSetDevice(prev);
    return rc;
}
