// inv.hip — inverses of the diagonal outer-panel blocks of the Cholesky factor (gfx950).
//
// GP::compute_inv_kernel (src/limbo/model/gp.hpp:254-264) forms L^-1 by a triangular solve against
// the identity and then K^-1 = L^-T L^-1.  On the device L^-1 comes from a blocked forward
// substitution over outer panels of `nbo` columns (engine.hip:ensure_inv).  Inside a panel that
// substitution is a chain of tiny dependent products — 8 launches per panel, ~1 ms of launch floor
// at N = 4096, half of the whole inversion.  This kernel removes the chain: in ONE launch it inverts
// every diagonal nbo x nbo block of L (all panels at once, they are independent), from the 64 x 64
// block inverses the factorisation already left behind (potrf.hip:k_diag).  The substitution then
// needs two matrix-core launches per panel: Y_p = X_p Acc_p and the update of the rows below.
// The engine works on the transposes (U = L^-T, upper triangular): every product is then of the form
// C -= A B^T with both operands contiguous along their non-k index, which is what the LDS-direct matrix-core
// kernel (gemm.hip, k_gemm_glds) wants.  So each block inverse is written twice: X_p as it is (the B operand of
// Y_p^T = Acc_p^T X_p^T) and transposed (the diagonal block of U).
//
// Work per panel is ~10 products of 64^3 — far too little for the matrix cores to matter; the
// kernel is plain FMA from LDS, one workgroup per 32-column strip of a panel (8 x npanels
// workgroups), the strip's tiles of the result staying in LDS as the right-hand operand of the
// later ones:   Y_ss = X_s,   Y_is = -X_i * sum_{k=s}^{i-1} L_ik Y_ks   (i > s, 64-row blocks).
#include <type_traits>
#include "dev.h"

#define NB 64
#define SW 32  // strip width (columns per workgroup)
#define AST 65 // row stride of the staged left operand (k-major): conflict-free transposed writes

// acc[r][c] += sum_kk As[kk][4 ty + r] * Bs[kk][2 tx + c]     (64 x 64 times 64 x SW)
static __device__ __forceinline__ void strip_mm(const double* __restrict__ As, const double* __restrict__ Bs, int tx,
                                                int ty, double (&acc)[4][2])
{
#pragma unroll 8
    for (int kk = 0; kk < NB; ++kk) {
        const double* a = As + kk * AST + 4 * ty;
        const double* b = Bs + kk * SW + 2 * tx;
        const double b0 = b[0], b1 = b[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[r][0] = fma(a[r], b0, acc[r][0]);
            acc[r][1] = fma(a[r], b1, acc[r][1]);
        }
    }
}

// Out[o0 : o0+pw, o0 : o0+pw] = inv(L[o0 : o0+pw, o0 : o0+pw]) for every outer panel (blockIdx.y), full square
// (zeros above the diagonal); OutT (optional): the same blocks transposed.  ldo == 0: the blocks alone, one after the
// other (what the batched query path multiplies by: engine.hip, query_impl).
// Xt_all: Xt[k + 64 c] = (L_bb^-1)[c][k] per 64-block b, identity-padded.
__global__ __launch_bounds__(256) void k_inv_panels(const double* __restrict__ L, int64_t ld, int64_t N, int nbo,
                                                    const double* __restrict__ Xt_all, double* __restrict__ Out,
                                                    int64_t ldo, double* __restrict__ OutT, int64_t ldt, const BatchTab* bt,
                                                    int panel0)
{
    BT_REBASE(bt, L); // batched launch (gridDim.z GPs, dev.h): this GP's buffers
    BT_REBASE(bt, Xt_all);
    BT_REBASE(bt, Out);
    BT_REBASE(bt, OutT);
    __shared__ double Ys[4][NB * SW]; // the strip's result tiles, Ys[i - s][kk][col]
    __shared__ double Ss[NB * SW];    // sum_k L_ik Y_ks, as the right-hand operand of X_i
    __shared__ double As[NB * AST];   // left operand, k-major: As[kk][row]
    const int64_t pidx = (int64_t)blockIdx.y + panel0; // (panel0: one panel at a time, behind the factorisation: engine.hip)
    const int64_t o0 = pidx * nbo;
    const int pw = (int)((N - o0 < nbo) ? N - o0 : nbo);
    const int c0 = blockIdx.x * SW; // first column of the strip inside the panel
    if (c0 >= pw)
        return;
    const int nb = (pw + NB - 1) / NB; // 64-row blocks of this panel
    const int s = c0 / NB;             // the block holding the strip's diagonal part
    const int cs = c0 - s * NB;        // strip offset inside that block (0 or 32)
    const int64_t gb0 = o0 / NB;       // global index of the panel's first 64-block
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    auto store_tile = [&](int i, const double (&v)[4][2]) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int64_t col = o0 + c0 + 2 * tx + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = o0 + (int64_t)i * NB + 4 * ty + r;
                if (row < N && col < N) {
                    if (ldo == 0) // compact form: panel p's block alone, column-major nbo x nbo, at Out + p nbo^2
                        Out[(row - o0) + (col - o0) * (int64_t)nbo + pidx * nbo * nbo] = v[r][c];
                    else
                        Out[row + col * ldo] = v[r][c];
                    if (OutT)
                        OutT[col + row * ldt] = v[r][c];
                }
            }
        }
    };
    // left operand <- X_i (lower triangular, explicit zeros above the diagonal)
    // (Round 6: every load of a staging pass unconditional and in flight together, the zero written where the value is used —
    // `cond ? load : 0` made a predicated load with a wait of its own of each: sixteen memory round trips per 64 x 64 tile,
    // 73 us for the launch; profiles/r06_inv_panels.log.)
    auto stage_X = [&](int i) {
        const double* Xt = Xt_all + (gb0 + i) * (NB * NB);
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            v[q] = Xt[threadIdx.x + 256 * q];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = threadIdx.x + 256 * q;
            const int kk = e & 63, row = e >> 6; // Xt[kk + 64 row] = X[row][kk]
            As[kk * AST + row] = (row >= kk) ? v[q] : 0.0;
        }
    };
    // left operand <- L tile (block row i, block column k of the panel); rows past N read as zero (rows N.. of the
    // factor's buffer hold the appended right-hand sides)
    auto stage_L = [&](int i, int k) {
        const double* Lt = L + (o0 + (int64_t)i * NB) + (o0 + (int64_t)k * NB) * ld;
        const int64_t rmax = N - (o0 + (int64_t)i * NB); // valid rows in this block
        const int row = threadIdx.x & 63, k0 = threadIdx.x >> 6;
        const int rc = row < rmax ? row : (int)(rmax > 0 ? rmax - 1 : 0);
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            v[q] = Lt[rc + (int64_t)(k0 + 4 * q) * ld];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            As[(k0 + 4 * q) * AST + row] = (row < rmax) ? v[q] : 0.0;
    };
    const double zero[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (int i = 0; i < s; ++i)
        store_tile(i, zero); // above the strip's diagonal block
    // Y_ss: columns cs .. cs+SW-1 of X_s
    {
        const double* Xt = Xt_all + (gb0 + s) * (NB * NB);
        double v[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int row = 4 * ty + r, col = cs + 2 * tx + c;
                v[r][c] = (row >= col) ? Xt[col + NB * row] : 0.0;
                Ys[0][row * SW + 2 * tx + c] = v[r][c];
            }
        store_tile(s, v);
    }
    for (int i = s + 1; i < nb; ++i) {
        double acc[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        for (int k = s; k < i; ++k) {
            __syncthreads(); // As free (and Ys[k - s] written)
            stage_L(i, k);
            __syncthreads();
            strip_mm(As, Ys[k - s], tx, ty, acc);
        }
        __syncthreads();
        stage_X(i);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                Ss[(4 * ty + r) * SW + 2 * tx + c] = acc[r][c];
        __syncthreads();
        double y[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        strip_mm(As, Ss, tx, ty, y);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                y[r][c] = -y[r][c];
                if (i - s < 4)
                    Ys[i - s][(4 * ty + r) * SW + 2 * tx + c] = y[r][c];
            }
        store_tile(i, y);
    }
}

// nbo: outer panel width, a multiple of 64, at most 256
void launch_inv_panels(hipStream_t s, const double* L, int64_t ld, int64_t N, int nbo, const double* Xt_all, double* Out,
                       int64_t ldo, double* OutT, int64_t ldt)
{
    if (N <= 0)
        return;
    const unsigned np = (unsigned)((N + nbo - 1) / nbo);
    GPE_LAUNCH(k_inv_panels, dim3((unsigned)(nbo / SW), np, (unsigned)g_batch.G), dim3(256), 0, s, L, ld, N, nbo, Xt_all, Out, ldo,
                       OutT, ldt, g_batch.bt, 0);
}
__global__ void k_zero2d(double* __restrict__ A, int64_t lda, int64_t rows, int64_t cols, const BatchTab* bt)
{
    BT_REBASE(bt, A);
    const int64_t c = blockIdx.y;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x)
        A[r + c * lda] = 0.0;
}
void launch_zero2d(hipStream_t s, double* A, int64_t lda, int64_t rows, int64_t cols)
{
    if (rows <= 0 || cols <= 0)
        return;
    const unsigned gx = (unsigned)((rows + 1023) / 1024 > 8 ? 8 : (rows + 1023) / 1024);
    GPE_LAUNCH(k_zero2d, dim3(gx, (unsigned)cols, (unsigned)g_batch.G), dim3(256), 0, s, A, lda, rows, cols, g_batch.bt);
}

