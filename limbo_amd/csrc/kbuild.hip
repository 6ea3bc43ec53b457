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
#include <cstdlib>

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
    GPE_LAUNCH(k_transpose_x, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, Xrm, n, D, Xt, ld, col0);
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
    GPE_LAUNCH(k_lambda_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Xt, ld, col0, n, lp);
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

    // stage the column-sample panel: DMAX x 64, coalesced (rows D .. DMAX - 1 are zeros, and so are their xi and 1 / ell: the pair
    // loop runs over DMAX dimensions without a condition, see k_build_wide)
    for (int e = threadIdx.x; e < DMAX * TILE; e += 256) {
        int d = e >> 6, c = e & 63;
        int64_t j = j0 + c;
        smem[e] = (d < D && j < ncol) ? Cs[(int64_t)d * ldc + j] : 0.0;
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
            double q = (xi[d] - smem[d * TILE + cc]) * ie[d]; // cwiseQuotient(_ell); a padded dimension adds an exact zero
            z = fma(q, q, z);
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
    int D = kp.D;
    const BatchTab* bt = (MODE == 0) ? g_batch.bt : nullptr;
    if (bt)
        grid.z = (unsigned)g_batch.G;
#define LB(DM)                                                                                                                  \
    do {                                                                                                                        \
        const size_t sh = (size_t)(DM) * TILE * sizeof(double);                                                                 \
        if (bt)                                                                                                                 \
            GPE_LAUNCH((k_build<DM, MODE, true>), grid, dim3(256), sh, s, Xt, ldx, N, Qt, ldq, M, kp, A, lda, bt);      \
        else                                                                                                                    \
            GPE_LAUNCH((k_build<DM, MODE, false>), grid, dim3(256), sh, s, Xt, ldx, N, Qt, ldq, M, kp, A, lda, bt);     \
    } while (0)
    if (D <= 2)
        LB(2);
    else if (D <= 4)
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


// ---------------------------------------------------------------------------------------------------------------------
// The training matrix's lower triangle, tuned for the HBM-write roofline (N (N + 1) / 2 * 8 B out, N * D * 8 B in).
// Differences of k_build_wide from the generic k_build above:
//   * specialised on the kernel functor: no switch inside the pair loop;
//   * branch-free pair loop: out-of-range rows/columns are clamped for the loads and masked at the store (diagonal
//     tiles compute the few upper-triangle pairs and drop them);
//   * exp(-h), h >= 0, without libm's special cases (kfun_fast.h): < 1 ulp from the correctly rounded value.
// Same pair formula and summation order as k_build: z = sum_d ((x_i,d - x_j,d) / ell_d)^2, d ascending, fma.
// (Round 2 measured a third form between the two — 64 x 64 tiles, column samples through the scalar unit, 8-byte stores:
// 32 us at N = 4096 like the other two, profiles/r02_kernel_build_trace.txt; removed in round 5 with its switch.)
// ---------------------------------------------------------------------------------------------------------------------
#include "kfun_fast.h"

// k_build_wide<KIND, DMAX, BATCH, CW> — shaped for the memory side: a workgroup owns 128 rows x CW
// columns, a lane owns TWO consecutive rows and stores them as one 16-byte double2 (1 KiB contiguous per wave store
// instead of 512 B), the CW column samples are staged once through LDS (wave-uniform LDS reads: broadcasts).
// Round 6, 32-34 -> 23-24 us at N = 4096 (2.8-2.9 TB/s of the 67 MB it writes; profiles/r06_kernel_build.log), three things the
// ISA showed: (1) `if (d < D)` inside the pair loop was a branch + an LDS read + a full wait per dimension per column — the loop now
// runs over DMAX dimensions unconditionally, the padded ones add exact zeros (DMAX 2, 4, 6, 8, 16 ..: no padding at D = 6);
// (2) with the triangle's masks in the loop the compiler split the 16-byte store into two predicated 8-byte ones and computed the
// second row's exp behind the first row's store — tiles strictly below the diagonal take a loop of their own without masks;
// (3) smaller workgroups (CW 32 instead of 64; 128 was slower still: 40 us).
// rt (single-GP launches): the workgroups from rt.first on do what k_cols_to_rows (solve.hip) does — obs_mean^T into the rows
// under the matrix, the backward sweep's output pre-filled with its sentinel — instead of a launch of its own behind this one
// (4.6 us + a launch boundary at the head of every evaluation).
template <int KIND, int DMAX, bool BATCH, int CW = 64>
__global__ __launch_bounds__(256) void k_build_wide(const double* __restrict__ Xt, int64_t ldx, int64_t N, KParams kp_,
                                                     double* __restrict__ A, int64_t lda, const BatchTab* __restrict__ bt,
                                                     BuildRowsTail rt)
{
    extern __shared__ __attribute__((aligned(16))) double smem[]; // xj[D][64]
    if (!BATCH && rt.V && (int64_t)blockIdx.x >= rt.first) {
        const int64_t i = ((int64_t)blockIdx.x - rt.first) * 256 + threadIdx.x;
        if (i < N)
            for (int p = 0; p < rt.P; ++p) {
                rt.Arows[p + i * lda] = rt.V[i + (int64_t)p * rt.ldv];
                if (rt.sent)
                    ((unsigned long long*)rt.sent)[i + (int64_t)p * rt.ldv] = ~0ull;
            }
        return;
    }
    if (BATCH) {
        Xt = bt_rebase(bt, (int)blockIdx.z, Xt);
        A = bt_rebase(bt, (int)blockIdx.z, A);
    }
#define KPF(field) (BATCH ? bt->kp[blockIdx.z].field : kp_.field)
    const int D = KPF(D);
    const double sf2 = KPF(sf2), diag_add = KPF(diag_add);
    // tile (ti: 128-row block, tj: CW-column block), live iff 128 ti + 127 >= CW tj  <=>  tj <= 2 ti + 1 (CW 64) / tj <= ti (CW 128)
    int ti, tj;
    {
        constexpr int R = 128 / CW; // column blocks per 128 rows: block-row ti has R (ti + 1) live ones, b = R ti (ti + 1) / 2 + tj
        const long long b = blockIdx.x;
        long long t = (long long)((sqrt(8.0 * (double)b / R + 1.0) - 1.0) * 0.5);
        while (R * (t + 1) * (t + 2) / 2 <= b)
            ++t;
        while (R * t * (t + 1) / 2 > b)
            --t;
        ti = (int)t;
        tj = (int)(b - R * t * (t + 1) / 2);
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)ti * 128 + 2 * tx; // rows i0, i0 + 1
    const int64_t j0 = (int64_t)tj * CW;
    // (rows D .. DMAX - 1 of the staged samples are zeros and so are their 1 / ell: the pair loop below runs over DMAX
    // dimensions without a condition — with `if (d < D)` inside it the compiler made a branch, an LDS read and a full wait of every
    // dimension of every column: 32 -> 28 us at N = 4096 went with the smaller workgroups, the rest with this)
    for (int e = threadIdx.x; e < DMAX * CW; e += 256) {
        const int d = e / CW, c = e % CW;
        const int64_t j = j0 + c;
        smem[e] = (d < D && j < N) ? Xt[(int64_t)d * ldx + j] : 0.0;
    }
    double xa[DMAX], xb[DMAX], ie[DMAX];
    const int64_t ia = i0 < N ? i0 : N - 1, ib = i0 + 1 < N ? i0 + 1 : N - 1;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
        ie[d] = d < D ? KPF(inv_ell[d]) : 0.0;
        xa[d] = d < D ? Xt[(int64_t)d * ldx + ia] : 0.0;
        xb[d] = d < D ? Xt[(int64_t)d * ldx + ib] : 0.0;
    }
#undef KPF
    __syncthreads();
    // A tile strictly below the diagonal and inside N (all but the 128 / CW last ones of a block row): no masks, no diagonal term, ONE
    // 16-byte store of the two rows per lane — 1 KiB contiguous per wave store.  (Written with the masks in the same loop the compiler
    // turned the 16-byte store into two predicated 8-byte ones and computed the second row's exp behind the first row's store.)
    const bool interior = (int64_t)ti * 128 >= j0 + CW && (int64_t)ti * 128 + 128 <= N;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    if (interior) {
#pragma unroll 4
        for (int c = 0; c < CW / 4; ++c) {
            const int cc = ty * (CW / 4) + c;
            double za = 0.0, zb = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                const double xj = smem[d * CW + cc];
                const double qa = (xa[d] - xj) * ie[d], qb = (xb[d] - xj) * ie[d];
                za = fma(qa, qa, za); // (a padded dimension adds an exact zero)
                zb = fma(qb, qb, zb);
            }
            d2_t v;
            v.x = kfun_fast<KIND>(za, sf2);
            v.y = kfun_fast<KIND>(zb, sf2);
            *reinterpret_cast<d2_t*>(A + i0 + (j0 + cc) * lda) = v; // (i0 and lda are even)
        }
        return;
    }
#pragma unroll 2
    for (int c = 0; c < CW / 4; ++c) {
        const int cc = ty * (CW / 4) + c;
        const int64_t j = j0 + cc; // wave-uniform
        if (j >= N)
            break;
        double za = 0.0, zb = 0.0;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
            const double xj = smem[d * CW + cc];
            const double qa = (xa[d] - xj) * ie[d], qb = (xb[d] - xj) * ie[d];
            za = fma(qa, qa, za);
            zb = fma(qb, qb, zb);
        }
        double va = kfun_fast<KIND>(za, sf2), vb = kfun_fast<KIND>(zb, sf2);
        if (i0 == j)
            va += diag_add;
        if (i0 + 1 == j)
            vb += diag_add;
        double* dst = A + i0 + j * lda;
        if (i0 < N && j <= i0)
            dst[0] = va;
        if (i0 + 1 < N && j <= i0 + 1)
            dst[1] = vb;
    }
}

template <int KIND>
static void launch_build_wide_kind(hipStream_t s, const double* Xt, int64_t ldx, int64_t N, const KParams& kp, double* A, int64_t lda,
                                   BuildRowsTail rt)
{
    const int64_t nt = (N + 127) / 128;
    // columns per workgroup: 32 for a single GP (2128 workgroups at N = 4096: measured 23.0-24.2 us against 24.2-24.4 with 64 and
    // 24.0-24.4 with 16), 64 for the members of a batch (gridDim.z fills the chip)
    const int cwx = g_batch.bt ? 64 : 32;
    // live tiles: sum over ti of min(2 ti + 2, column blocks)
    const int64_t ncb = (N + TILE - 1) / TILE;
    int64_t tiles = (128 / cwx) * nt * (nt + 1) / 2; // the last block-row may count a column block past N: those workgroups find j >= N and leave
    (void)ncb;
    rt.first = tiles;
    if (rt.V)
        tiles += (N + 255) / 256;
    dim3 grid((unsigned)tiles, 1, (unsigned)g_batch.G);
    const BatchTab* bt = g_batch.bt;
#define LBW(DM)                                                                                                     \
    do {                                                                                                            \
        const size_t sh = (size_t)(DM) * cwx * sizeof(double);                                                      \
        if (bt)                                                                                                     \
            GPE_LAUNCH((k_build_wide<KIND, DM, true>), grid, dim3(256), sh, s, Xt, ldx, N, kp, A, lda, bt, rt);  \
        else                                                                                                        \
            GPE_LAUNCH((k_build_wide<KIND, DM, false, 32>), grid, dim3(256), sh, s, Xt, ldx, N, kp, A, lda, bt, rt); \
    } while (0)
    if (kp.D <= 2)
        LBW(2);
    else if (kp.D <= 4)
        LBW(4);
    else if (kp.D <= 6)
        LBW(6);
    else if (kp.D <= 8)
        LBW(8);
    else if (kp.D <= 16)
        LBW(16);
    else if (kp.D <= 32)
        LBW(32);
    else
        LBW(64);
#undef LBW
}

bool launch_build_K(hipStream_t s, const double* Xt, int64_t ldx, int64_t N, const KParams& kp, double* A, int64_t lda,
                    const BuildRowsTail* tail)
{
    BuildRowsTail rt{};
    if (tail && !g_batch.bt && tail->P > 0)
        rt = *tail;
    if (N <= 0 || (lda & 1)) { // (the wide kernel stores pairs of rows: an odd leading dimension takes the generic one)
        launch_build<0>(s, Xt, ldx, N, nullptr, 0, 0, kp, A, lda);
        return false;
    }
    switch (kp.kind) {
    case 0:
    case 3: launch_build_wide_kind<0>(s, Xt, ldx, N, kp, A, lda, rt); break;
    case 1: launch_build_wide_kind<1>(s, Xt, ldx, N, kp, A, lda, rt); break;
    default: launch_build_wide_kind<2>(s, Xt, ldx, N, kp, A, lda, rt); break;
    }
    return rt.V != nullptr;
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
    GPE_LAUNCH(k_kvv, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, M, kp, kvv);
}
