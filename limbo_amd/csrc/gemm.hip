// gemm.hip — fp64 MFMA (v_mfma_f64_16x16x4_f64) rank-k update for gfx950.
//
//   C[m x n] -= opA[m x k] * opB[n x k]^T          (or C = opA opB^T when `overwrite`)
//
// This one kernel is the contraction engine of the whole path:
//   * Cholesky trailing update A22 -= L21 L21^T  (replaces the rankUpdate inside Eigen::LLT,
//     call-site src/limbo/model/gp.hpp:565),
//   * the block forward substitution for L^-1 and for batched query variances
//     (gp.hpp:260-261, :620),
//   * K^-1 = L^-T L^-1 as X^T X with a triangular k range (gp.hpp:254-264).
//
// Tile: 128 x 128 x 16 per 256-thread workgroup (4 waves as 2 x 2, each wave a 64 x 64
// sub-tile = 4 x 4 MFMA tiles of 16x16, 128 accumulator VGPRs).  Operands are staged through
// LDS (double-buffered, register prefetch of the next k-tile).  The MFMA is issued
// "transposed" (A-operand <- B panel, B-operand <- A panel) so that the fp64 C/D layout
// (col = lane&15, row = (lane>>4) + 4*reg — NOT the f32 map) puts 16 consecutive lanes on 16
// consecutive ROWS of column-major C: every C access is a 128-byte segment.
//
// LDS layouts (conflict-free ds_read_b64 per 32-lane half):
//   row-contiguous operand : S[kk][i], row stride 144 doubles  (144 = 16 mod 32)
//   k-contiguous operand   : S[i][kk], row stride 18 doubles   (18*i + h covers 32 banks)
#include "dev.h"

#define BM 128
#define BN 128
#define BKT 16
#define LDS_M 144 // stride of S[kk][i]
#define LDS_K 18  // stride of S[i][kk]
#define OPER_ELEMS 2304 // 16*144 == 128*18

template <bool KMAJOR>
struct Stager {
    // each thread moves 8 elements of a 128 x 16 operand tile
    double r[8];
    // global -> registers (guarded: rows < rows_left, kk < k_left; zero fill)
    __device__ __forceinline__ void load(const double* __restrict__ P, int64_t ld, int64_t rows_left, int64_t k_left)
    {
        const int t = threadIdx.x;
        if (!KMAJOR) {
            const int i = t & 127, kk0 = t >> 7;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int kk = kk0 + 2 * q;
                r[q] = (i < rows_left && kk < k_left) ? P[i + (int64_t)kk * ld] : 0.0;
            }
        }
        else {
            const int kk = t & 15, j0 = t >> 4;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int j = j0 + 16 * q;
                r[q] = (j < rows_left && kk < k_left) ? P[kk + (int64_t)j * ld] : 0.0;
            }
        }
    }
    __device__ __forceinline__ void store(double* __restrict__ S) const
    {
        const int t = threadIdx.x;
        if (!KMAJOR) {
            const int i = t & 127, kk0 = t >> 7;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                S[(kk0 + 2 * q) * LDS_M + i] = r[q];
        }
        else {
            const int kk = t & 15, j0 = t >> 4;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                S[(j0 + 16 * q) * LDS_K + kk] = r[q];
        }
    }
    // fragment element for MFMA lane l: operand row (off + (l&15)), k index (k0 + (l>>4))
    static __device__ __forceinline__ double frag(const double* __restrict__ S, int off, int k0, int lane)
    {
        if (!KMAJOR)
            return S[(k0 + (lane >> 4)) * LDS_M + off + (lane & 15)];
        else
            return S[(off + (lane & 15)) * LDS_K + k0 + (lane >> 4)];
    }
};

template <bool AK, bool BK>
__global__ __launch_bounds__(256, 2) void k_gemm_sub(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) double lds[2][2][OPER_ELEMS];
    // XCD-aware tile order: workgroups b, b+8, b+16.. land on the same XCD (private L2), so give
    // each XCD a contiguous band of tile rows: they share the same A row-panel.
    const int tiles_m = (int)((g.m + BM - 1) / BM);
    const int tiles_n = (int)((g.n + BN - 1) / BN);
    const int nwg = tiles_m * tiles_n;
    int wg = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ti = wg % tiles_m; // row tile fastest: neighbours share the B (column) panel
    const int tj = wg / tiles_m;
    const int64_t row0 = (int64_t)ti * BM, col0 = (int64_t)tj * BN;
    if (g.tri && (g.grow0 + row0 + BM - 1 < g.gcol0 + col0))
        return; // tile entirely above the diagonal
    int64_t kbeg = 0;
    if (g.ktri) { // X^T X with X lower triangular: rows of X below max(row0, col0) only
        kbeg = (row0 > col0 ? row0 : col0);
        kbeg -= kbeg % BKT;
    }
    const int64_t kend = g.k;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;

    const double* Ap = AK ? g.A + kbeg + row0 * g.lda : g.A + row0 + kbeg * g.lda;
    const double* Bp = BK ? g.B + kbeg + col0 * g.ldb : g.B + col0 + kbeg * g.ldb;
    const int64_t a_kstep = AK ? (int64_t)BKT : (int64_t)BKT * g.lda;
    const int64_t b_kstep = BK ? (int64_t)BKT : (int64_t)BKT * g.ldb;
    const int64_t mrows = g.m - row0, ncols = g.n - col0;

    d4_t acc[4][4]; // [nt][mt]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            acc[a][b] = d4_t{0.0, 0.0, 0.0, 0.0};

    Stager<AK> sa;
    Stager<BK> sb;
    int64_t kleft = kend - kbeg;
    if (kleft > 0) {
        sa.load(Ap, g.lda, mrows, kleft);
        sb.load(Bp, g.ldb, ncols, kleft);
        sa.store(lds[0][0]);
        sb.store(lds[0][1]);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BKT) {
        const bool more = (k0 + BKT < kend);
        if (more) { // prefetch next k-tile into registers while this one is consumed
            Ap += a_kstep;
            Bp += b_kstep;
            sa.load(Ap, g.lda, mrows, kend - k0 - BKT);
            sb.load(Bp, g.ldb, ncols, kend - k0 - BKT);
        }
        const double* As = lds[buf][0];
        const double* Bs = lds[buf][1];
#pragma unroll
        for (int ks = 0; ks < BKT; ks += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                af[x] = Stager<AK>::frag(As, wm + 16 * x, ks, lane);
                bf[x] = Stager<BK>::frag(Bs, wn + 16 * x, ks, lane);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = mfma_f64(bf[nt], af[mt], acc[nt][mt]);
        }
        if (more) {
            sa.store(lds[buf ^ 1][0]);
            sb.store(lds[buf ^ 1][1]);
        }
        __syncthreads();
        buf ^= 1;
    }

    // epilogue.  D layout (fp64): lane l, reg v -> (n = (l>>4) + 4v, m = l&15) of the 16x16 tile
    const int lm = lane & 15, ln = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t n = col0 + wn + nt * 16 + ln + 4 * v;
            if (n >= g.n)
                continue;
            double* Cc = g.C + n * g.ldc;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int64_t mrow = row0 + wm + mt * 16 + lm;
                if (mrow < g.m) {
                    if (g.overwrite)
                        Cc[mrow] = acc[nt][mt][v];
                    else
                        Cc[mrow] -= acc[nt][mt][v];
                }
            }
        }
    }
}

void launch_gemm_sub(hipStream_t s, const GemmArgs& g)
{
    if (g.m <= 0 || g.n <= 0)
        return;
    if (g.k <= 0 && !g.overwrite)
        return;
    int64_t tiles = ((g.m + BM - 1) / BM) * ((g.n + BN - 1) / BN);
    dim3 grid((unsigned)tiles), block(256);
    if (!g.a_kmajor && !g.b_kmajor)
        hipLaunchKernelGGL((k_gemm_sub<false, false>), grid, block, 0, s, g);
    else if (!g.a_kmajor && g.b_kmajor)
        hipLaunchKernelGGL((k_gemm_sub<false, true>), grid, block, 0, s, g);
    else if (g.a_kmajor && !g.b_kmajor)
        hipLaunchKernelGGL((k_gemm_sub<true, false>), grid, block, 0, s, g);
    else
        hipLaunchKernelGGL((k_gemm_sub<true, true>), grid, block, 0, s, g);
}

// algorithmic flops of one launch (2 m n k, lower-triangular tile skipping accounted for at
// ELEMENT level: tri -> only elements on/below the diagonal count; ktri -> k from max(i,j))
double gemm_flops(const GemmArgs& g)
{
    double m = (double)g.m, n = (double)g.n, k = (double)g.k;
    if (g.ktri) // sum_{i>=j} 2 (k - i)  over an n x n lower triangle with k == n
        return n * n * n / 3.0;
    if (g.tri) {
        // count elements (i, j) with grow0 + i >= gcol0 + j
        double off = (double)(g.grow0 - g.gcol0); // >= 0 in every call-site
        double cnt = 0.0;
        // columns j = 0..n-1: rows i >= j - off  -> m - max(0, j - off)
        double jfull = off < n ? off : n;             // columns with all m rows
        cnt += jfull * m;
        double rest = n - jfull;                      // columns j = off .. n-1: rows m - (j - off)
        if (rest > 0) {
            double first = m, last = m - (rest - 1);
            if (last < 0) {
                rest = m + 1;
                last = 0;
            }
            cnt += 0.5 * (first + last) * rest;
        }
        return 2.0 * cnt * k;
    }
    return 2.0 * m * n * k;
}
