// potrf.hip — the serial pieces of the blocked Cholesky and of the triangular solves (gfx950).
//
// Together with gemm.hip these replace Eigen::LLT<MatrixXd>(K).matrixL()
// (src/limbo/model/gp.hpp:565) and the TriangularView solves (gp.hpp:260-261, :608-610, :620).
//
//   k_potf2       64x64 diagonal block, one 256-thread workgroup, block kept in registers
//                 (thread = (row, column class mod 4)), one LDS column broadcast + one barrier
//                 per column step.
//   k_trsm_right  rows below the diagonal block: X <- X L11^-T, one row per lane, the row's 64
//                 entries in registers, L11 read as LDS broadcasts.
//   k_trsm_left   64-row block times many columns: B <- L11^-1 B or L11^-T B, one column per lane.
#include "dev.h"

#define NB 64

// ---------------------------------------------------------------------------------------
// diagonal block
// ---------------------------------------------------------------------------------------
template <int J>
struct Potf2Steps {
    static __device__ __forceinline__ void run(double (&a)[16], double (*colbuf)[NB], int r, int cg, int& bad)
    {
        Potf2Steps<J - 1>::run(a, colbuf, r, cg, bad);
        constexpr int own = J & 3, mj = J >> 2, pb = J & 1;
        if (cg == own)
            colbuf[pb][r] = a[mj];
        __syncthreads();
        const double d = colbuf[pb][J];
        if (!(d > 0.0) && bad == 0)
            bad = J + 1;
        const double ljj = sqrt(d);
        const double inv = 1.0 / ljj;
        const double lr = colbuf[pb][r] * inv; // L[r, J] for r > J
        if (cg == own)
            a[mj] = (r == J) ? ljj : lr;
        // trailing columns c = cg + 4m > J
        if (cg > own) {
            const double lc = colbuf[pb][cg + 4 * mj] * inv;
            a[mj] = fma(-lr, lc, a[mj]);
        }
#pragma unroll
        for (int m = mj + 1; m < 16; ++m) {
            const double lc = colbuf[pb][cg + 4 * m] * inv;
            a[m] = fma(-lr, lc, a[m]);
        }
    }
};
template <>
struct Potf2Steps<-1> {
    static __device__ __forceinline__ void run(double (&)[16], double (*)[NB], int, int, int&) {}
};

__global__ __launch_bounds__(256) void k_potf2(double* __restrict__ A, int64_t lda, int jb, int* __restrict__ info,
                                               int64_t goff)
{
    __shared__ double colbuf[2][NB];
    const int r = threadIdx.x & 63, cg = threadIdx.x >> 6;
    double a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int c = cg + 4 * m;
        // only the lower triangle of A is meaningful; pad a short block with the identity
        a[m] = (r < jb && c < jb) ? ((c <= r) ? A[r + (int64_t)c * lda] : 0.0) : ((r == c) ? 1.0 : 0.0);
    }
    int bad = 0;
    Potf2Steps<NB - 1>::run(a, colbuf, r, cg, bad);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int c = cg + 4 * m;
        if (r < jb && c <= r)
            A[r + (int64_t)c * lda] = a[m];
    }
    if (threadIdx.x == 0 && bad != 0 && bad <= jb && *info == 0)
        *info = (int)(goff + bad);
}

void launch_potf2(hipStream_t s, double* A, int64_t lda, int jb, int* info, int64_t goff)
{
    hipLaunchKernelGGL(k_potf2, dim3(1), dim3(256), 0, s, A, lda, jb, info, goff);
}

// ---------------------------------------------------------------------------------------
// shared: stage the 64x64 diagonal block into LDS as Ls[j][k] (row j contiguous) with the
// reciprocal diagonal; TRANS stores U = L^T (U[j][k] = L[k][j]).  Short blocks padded with I.
// ---------------------------------------------------------------------------------------
template <bool TRANS>
__device__ __forceinline__ void stage_L11(const double* __restrict__ L11, int64_t ldl, int jb, double* Ls,
                                          double* invd, int nthreads)
{
    for (int e = threadIdx.x; e < NB * NB; e += nthreads) {
        const int r = e & 63, c = e >> 6; // coalesced along rows of the column-major block
        double v = 0.0;
        if (r < jb && c < jb)
            v = (c <= r) ? L11[r + (int64_t)c * ldl] : 0.0;
        else if (r == c)
            v = 1.0;
        if (!TRANS)
            Ls[r * NB + c] = v; // Ls[j=r][k=c] = L[r][c]
        else
            Ls[c * NB + r] = v; // Us[j=c][k=r] = L[r][c]
        if (r == c)
            invd[r] = 1.0 / v;
    }
}

// x[j] = (x[j] - sum_{k<j} Ls[j][k] x[k]) * invd[j]   (forward;  lower L)
template <int J>
struct FwdSub {
    static __device__ __forceinline__ void run(double (&x)[NB], const double* __restrict__ Ls,
                                               const double* __restrict__ invd)
    {
        FwdSub<J - 1>::run(x, Ls, invd);
        asm volatile("" ::: "memory"); // keep hipcc from hoisting all 2016 LDS reads (15 KB/lane of scratch)
        double s = x[J];
#pragma unroll
        for (int k = 0; k < J; ++k)
            s = fma(-Ls[J * NB + k], x[k], s);
        x[J] = s * invd[J];
    }
};
template <>
struct FwdSub<-1> {
    static __device__ __forceinline__ void run(double (&)[NB], const double*, const double*) {}
};
// x[j] = (x[j] - sum_{k>j} Us[j][k] x[k]) * invd[j]   (backward; U = L^T)
template <int J>
struct BwdSub {
    static __device__ __forceinline__ void run(double (&x)[NB], const double* __restrict__ Us,
                                               const double* __restrict__ invd)
    {
        BwdSub<J + 1>::run(x, Us, invd);
        asm volatile("" ::: "memory");
        double s = x[J];
#pragma unroll
        for (int k = J + 1; k < NB; ++k)
            s = fma(-Us[J * NB + k], x[k], s);
        x[J] = s * invd[J];
    }
};
template <>
struct BwdSub<NB> {
    static __device__ __forceinline__ void run(double (&)[NB], const double*, const double*) {}
};

// ---------------------------------------------------------------------------------------
// rows below the diagonal block:  X <- X * L11^-T   (row r: solve L11 x^T = a_r^T, forward)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_trsm_right(const double* __restrict__ L11, int64_t ldl, int jb,
                                                   double* __restrict__ A21, int64_t lda, int64_t m)
{
    __shared__ double Ls[NB * NB];
    __shared__ double invd[NB];
    __shared__ double Xs[NB * NB]; // Xs[j][lane]: the row panel, staged so the global loop stays rolled
    stage_L11<false>(L11, ldl, jb, Ls, invd, 64);
    const int lane = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * 64 + lane;
    for (int j = 0; j < NB; ++j)
        Xs[j * NB + lane] = (r < m && j < jb) ? A21[r + (int64_t)j * lda] : 0.0;
    __syncthreads();
    double x[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
        x[j] = Xs[j * NB + lane];
    FwdSub<NB - 1>::run(x, Ls, invd);
#pragma unroll
    for (int j = 0; j < NB; ++j)
        Xs[j * NB + lane] = x[j];
    if (r < m) {
        for (int j = 0; j < jb; ++j)
            A21[r + (int64_t)j * lda] = Xs[j * NB + lane];
    }
}

void launch_trsm_right(hipStream_t s, const double* L11, int64_t ldl, int jb, double* A21, int64_t lda, int64_t m)
{
    if (m <= 0)
        return;
    hipLaunchKernelGGL(k_trsm_right, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s, L11, ldl, jb, A21, lda, m);
}

// ---------------------------------------------------------------------------------------
// 64-row block, many right-hand sides:  B <- L11^-1 B   or   B <- L11^-T B
// one column per lane; the 64x64 tile of B goes through LDS so that global access is
// coalesced along the rows (column-major B).
// ---------------------------------------------------------------------------------------
template <bool TRANS>
__global__ __launch_bounds__(64) void k_trsm_left(const double* __restrict__ L11, int64_t ldl, int jb,
                                                  double* __restrict__ B, int64_t ldb, int64_t nrhs)
{
    __shared__ double Ls[NB * NB];
    __shared__ double invd[NB];
    __shared__ double Bs[NB * (NB + 1)];
    stage_L11<TRANS>(L11, ldl, jb, Ls, invd, 64);
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const int lane = threadIdx.x;
    for (int c = 0; c < NB; ++c) { // lane = row k: coalesced
        const int64_t col = c0 + c;
        Bs[c * (NB + 1) + lane] = (col < nrhs && lane < jb) ? B[lane + col * ldb] : 0.0;
    }
    __syncthreads();
    double x[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k)
        x[k] = Bs[lane * (NB + 1) + k];
    if (!TRANS)
        FwdSub<NB - 1>::run(x, Ls, invd);
    else
        BwdSub<0>::run(x, Ls, invd);
#pragma unroll
    for (int k = 0; k < NB; ++k)
        Bs[lane * (NB + 1) + k] = x[k];
    __syncthreads();
    for (int c = 0; c < NB; ++c) {
        const int64_t col = c0 + c;
        if (col < nrhs && lane < jb)
            B[lane + col * ldb] = Bs[c * (NB + 1) + lane];
    }
}

void launch_trsm_left(hipStream_t s, const double* L11, int64_t ldl, int jb, double* B, int64_t ldb, int64_t nrhs,
                      int trans)
{
    if (nrhs <= 0)
        return;
    dim3 grid((unsigned)((nrhs + 63) / 64));
    if (trans)
        hipLaunchKernelGGL((k_trsm_left<true>), grid, dim3(64), 0, s, L11, ldl, jb, B, ldb, nrhs);
    else
        hipLaunchKernelGGL((k_trsm_left<false>), grid, dim3(64), 0, s, L11, ldl, jb, B, ldb, nrhs);
}
