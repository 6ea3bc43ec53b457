// kbuild.hip — pairwise kernel-matrix build for SquaredExpARD / Matern / Exp (gfx950).
//
// Replaces the scalar double loop of limbo::model::GP::_compute_full_kernel
// (src/limbo/model/gp.hpp:556-562) and GP::_compute_k (gp.hpp:626-632), including
// BaseKernel::operator()'s "+ noise + 1e-8 iff i == j" (src/limbo/kernel/kernel.hpp:81-84).
//
// Layout: samples are SoA in HBM (Xt[d*ldx + i]) so a 64-sample tile is D coalesced 512-byte
// rows; a 64x64 output tile stages both X panels in LDS once, each thread keeps its own
// sample in registers and walks 16 columns (the j-sample is an LDS broadcast).  Output is
// column-major, so a wave's 64 lanes store 512 contiguous bytes per column.  Only the lower
// triangle is produced (N(N+1)/2 * 8 B of HBM writes: the HBM-write roofline of this kernel).
#include "dev.h"

#define TILE 64

__global__ void k_transpose_x(const double* __restrict__ Xrm, int64_t n, int D, double* __restrict__ Xt, int64_t ld,
                              int64_t col0)
{
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * D)
        return;
    int64_t i = idx / D;
    int d = (int)(idx - i * D);
    Xt[(int64_t)d * ld + col0 + i] = Xrm[idx];
}

void launch_transpose_x(hipStream_t s, const double* Xrm, int64_t n, int D, double* Xt, int64_t ld, int64_t col0)
{
    int64_t tot = n * D;
    if (tot <= 0)
        return;
    hipLaunchKernelGGL(k_transpose_x, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, Xrm, n, D, Xt, ld, col0);
}

__global__ void k_lambda_rows(double* __restrict__ Xt, int64_t ld, int64_t col0, int64_t n, LamParams lp)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    for (int j = 0; j < lp.k; ++j) {
        double f = 0.0; // (x^T Lambda)_j: the same left-to-right sum as `(x1 - x2).transpose() * _A.col(j)`
        for (int d = 0; d < lp.D; ++d)
            f = fma(Xt[(int64_t)d * ld + col0 + i], lp.A[d + j * lp.D], f);
        Xt[(int64_t)(lp.D + j) * ld + col0 + i] = f;
    }
}

void launch_lambda_rows(hipStream_t s, double* Xt, int64_t ld, int64_t col0, int64_t n, const LamParams& lp)
{
    if (n <= 0 || lp.k <= 0)
        return;
    hipLaunchKernelGGL(k_lambda_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Xt, ld, col0, n, lp);
}

// MODE 0: lower triangle of the symmetric training matrix (+diag_add on i==j)
// MODE 1: full symmetric training matrix (tests)
// MODE 2: rectangular cross matrix, no noise
template <int DMAX, int MODE, bool BATCH = false>
__global__ __launch_bounds__(256) void k_build(const double* __restrict__ Xt, int64_t ldx, int64_t N,
                                               const double* __restrict__ Qt, int64_t ldq, int64_t M, KParams kp_,
                                               double* __restrict__ A, int64_t lda, const BatchTab* __restrict__ bt)
{
    // batched (gridDim.z GPs): this GP's buffers and ITS kernel parameters (theta differs from GP to GP)
    if (BATCH) {
        Xt = bt_rebase(bt, (int)blockIdx.z, Xt);
        A = bt_rebase(bt, (int)blockIdx.z, A);
    }
#define KPF(field) (BATCH ? bt->kp[blockIdx.z].field : kp_.field)
    const int kp_D = KPF(D), kp_kind = KPF(kind);
    const double kp_sf2 = KPF(sf2), kp_diag_add = KPF(diag_add);
    double ie[DMAX]; // 1 / ell_d in registers (the loops over d are fully unrolled)
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
        ie[d] = d < kp_D ? KPF(inv_ell[d]) : 0.0;
#undef KPF
    extern __shared__ __attribute__((aligned(16))) double smem[]; // xj[D][TILE]
    int ti, tj;
    if (MODE == 0) {
        // linear block id -> (ti >= tj) lower-triangle tile
        long long b = blockIdx.x;
        long long t = (long long)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
        while ((t + 1) * (t + 2) / 2 <= b)
            ++t;
        while (t * (t + 1) / 2 > b)
            --t;
        ti = (int)t;
        tj = (int)(b - t * (t + 1) / 2);
    }
    else {
        ti = blockIdx.x;
        tj = blockIdx.y;
    }
    const int D = kp_D;
    const int tx = threadIdx.x & 63; // row inside the tile
    const int ty = threadIdx.x >> 6; // 4 column groups of 16
    const int64_t i = (int64_t)ti * TILE + tx;
    const int64_t j0 = (int64_t)tj * TILE;
    const double* Cs = (MODE == 2) ? Qt : Xt; // column samples
    const int64_t ldc = (MODE == 2) ? ldq : ldx;
    const int64_t ncol = (MODE == 2) ? M : N;

    // stage the column-sample panel: D x 64, coalesced
    for (int e = threadIdx.x; e < D * TILE; e += 256) {
        int d = e >> 6, c = e & 63;
        int64_t j = j0 + c;
        smem[e] = (j < ncol) ? Cs[(int64_t)d * ldc + j] : 0.0;
    }
    double xi[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
        xi[d] = (d < D && i < N) ? Xt[(int64_t)d * ldx + i] : 0.0;
    __syncthreads();
    if (i >= N)
        return;
#pragma unroll 4
    for (int c = 0; c < 16; ++c) {
        int cc = ty * 16 + c;
        int64_t j = j0 + cc;
        if (j >= ncol)
            break;
        if (MODE == 0 && j > i)
            break; // strictly upper part of a diagonal tile
        double z = 0.0;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
            if (d < D) {
                double q = (xi[d] - smem[d * TILE + cc]) * ie[d]; // cwiseQuotient(_ell)
                z = fma(q, q, z);
            }
        }
        double v = kfun(kp_kind, z, kp_sf2);
        if (MODE != 2 && i == j)
            v += kp_diag_add;
        A[i + j * lda] = v;
    }
}

template <int MODE>
static void launch_build(hipStream_t s, const double* Xt, int64_t ldx, int64_t N, const double* Qt, int64_t ldq,
                         int64_t M, const KParams& kp, double* A, int64_t lda)
{
    if (N <= 0)
        return;
    int64_t nt = (N + TILE - 1) / TILE;
    dim3 grid;
    if (MODE == 0)
        grid = dim3((unsigned)(nt * (nt + 1) / 2));
    else if (MODE == 1)
        grid = dim3((unsigned)nt, (unsigned)nt);
    else
        grid = dim3((unsigned)nt, (unsigned)((M + TILE - 1) / TILE));
    size_t sh = (size_t)kp.D * TILE * sizeof(double);
    int D = kp.D;
    const BatchTab* bt = (MODE == 0) ? g_batch.bt : nullptr;
    if (bt)
        grid.z = (unsigned)g_batch.G;
#define LB(DM)                                                                                                                  \
    do {                                                                                                                        \
        if (bt)                                                                                                                 \
            hipLaunchKernelGGL((k_build<DM, MODE, true>), grid, dim3(256), sh, s, Xt, ldx, N, Qt, ldq, M, kp, A, lda, bt);      \
        else                                                                                                                    \
            hipLaunchKernelGGL((k_build<DM, MODE, false>), grid, dim3(256), sh, s, Xt, ldx, N, Qt, ldq, M, kp, A, lda, bt);     \
    } while (0)
    if (D <= 4)
        LB(4);
    else if (D <= 8)
        LB(8);
    else if (D <= 16)
        LB(16);
    else if (D <= 32)
        LB(32);
    else
        LB(64);
#undef LB
}

void launch_build_K(hipStream_t s, const double* Xt, int64_t ldx, int64_t N, const KParams& kp, double* A, int64_t lda)
{
    launch_build<0>(s, Xt, ldx, N, nullptr, 0, 0, kp, A, lda);
}
void launch_build_K_full(hipStream_t s, const double* Xt, int64_t ldx, int64_t N, const KParams& kp, double* A,
                         int64_t lda)
{
    launch_build<1>(s, Xt, ldx, N, nullptr, 0, 0, kp, A, lda);
}
void launch_build_Ks(hipStream_t s, const double* Xt, int64_t ldx, int64_t N, const double* Qt, int64_t ldq, int64_t M,
                     const KParams& kp, double* Ks, int64_t ldk)
{
    if (M <= 0)
        return;
    launch_build<2>(s, Xt, ldx, N, Qt, ldq, M, kp, Ks, ldk);
}

// k(v, v) for every query point (gp.hpp:621 `_kernel_function(v, v)`, defaults i=-1,j=-2: no noise)
__global__ void k_kvv(int64_t M, KParams kp, double* __restrict__ kvv)
{
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M)
        kvv[m] = kfun(kp.kind, 0.0, kp.sf2);
}
void launch_kvv(hipStream_t s, const double* Qt, int64_t ldq, int64_t M, const KParams& kp, double* kvv)
{
    (void)Qt;
    (void)ldq;
    if (M <= 0)
        return;
    hipLaunchKernelGGL(k_kvv, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, M, kp, kvv);
}
