// gemm.hip — fp64 matrix-core rank-k update for gfx950, built on v_mfma_f64_4x4x4_4b.
//
//   C[m x n] -= opA[m x k] * opB[n x k]^T          (or C = opA opB^T when `overwrite`)
//
// This one kernel is the contraction engine of the whole path:
//   * Cholesky trailing / in-panel updates A22 -= L21 L21^T  (replaces the rankUpdate inside
//     Eigen::LLT, call-site src/limbo/model/gp.hpp:565),
//   * the block forward substitution for L^-1 and for batched query variances
//     (gp.hpp:260-261, :620),
//   * K^-1 = L^-T L^-1 as X^T X with a triangular k range (gp.hpp:254-264).
//
// Why the 4x4x4 "4 blocks" instruction and not v_mfma_f64_16x16x4: measured on MI355X
// (tools/ubench2.hip, profiles/r01_ubench.log) the 16x16x4 form tops out at 48 TFLOP/s
// (one instruction per ~100-140 cycles per SIMD), the 4x4x4_4b form issues every 20 cycles and
// reaches 76-77 TFLOP/s = 97 % of the 78.6 TFLOP/s fp64 peak.  It computes four independent
// 4x4x4 products per instruction:
//     A operand lane l : block (l>>2)&3, row i = l&3,  k = l>>4
//     B operand lane l : block (l>>2)&3, col j = l&3,  k = l>>4
//     D result  lane l : block (l>>2)&3, col j = l&3,  row i = l>>4          (probed: tools/probe4.hip)
// Here the four blocks are four 4-row groups of a 16-row slab of opA, and the B operand is ONE
// 4(k) x 4(col) block of opB replicated over the blocks (the replication is free: the lanes of
// the four blocks read the same LDS address).  One instruction therefore produces a 16 x 4
// piece of C with k = 4; a wave holds RA x RB such pieces (RA 16-row slabs x RB 4-column
// groups) in RA*RB accumulator registers and needs RA + RB ds_read_b64 per RA*RB MFMAs.
// D rows for a fixed column are 16 consecutive rows of column-major C (scrambled lane order):
// every C access instruction touches four 128-byte segments.
//
// Workgroup tile TM x TN x 16 (256 threads, 4 waves as 2 x 2), operands staged through LDS
// (double-buffered, register prefetch of the next k-tile).  Three tile shapes are instantiated;
// launch_gemm_sub picks the largest one that still yields enough workgroups to cover the 256 CUs
// (the latency-critical in-panel updates of the blocked Cholesky are tiny).
//
// LDS layouts (conflict-free ds_read_b64 per 32-lane half):
//   row-contiguous operand : S[kk][i], row stride = ROWS + 16 doubles  (== 16 mod 32)
//   k-contiguous operand   : S[i][kk], row stride 18 doubles
#include "dev.h"
#include <cstdlib>

#define BKT 16
#define LDS_K 18 // stride of S[i][kk]

static __device__ __forceinline__ double mfma4(double a, double b, double c)
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

template <bool KMAJOR, int ROWS>
struct Stager {
    static constexpr int PER = ROWS * BKT / 256;              // elements per thread
    static constexpr int STRIDE = ROWS + 16;                  // S[kk][i] row stride
    static constexpr int ELEMS = KMAJOR ? ROWS * LDS_K : BKT * STRIDE;
    double r[PER];
    // global -> registers (guarded: rows < rows_left, kk < k_left; zero fill)
    __device__ __forceinline__ void load(const double* __restrict__ P, int64_t ld, int64_t rows_left, int64_t k_left)
    {
        const int t = threadIdx.x;
        if (!KMAJOR) {
            const int i = t % ROWS, kk0 = t / ROWS;
            constexpr int KS = 256 / ROWS;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int kk = kk0 + KS * q;
                r[q] = (i < rows_left && kk < k_left) ? P[i + (int64_t)kk * ld] : 0.0;
            }
        }
        else {
            const int kk = t & 15, j0 = t >> 4;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int j = j0 + 16 * q;
                r[q] = (j < rows_left && kk < k_left) ? P[kk + (int64_t)j * ld] : 0.0;
            }
        }
    }
    __device__ __forceinline__ void store(double* __restrict__ S) const
    {
        const int t = threadIdx.x;
        if (!KMAJOR) {
            const int i = t % ROWS, kk0 = t / ROWS;
            constexpr int KS = 256 / ROWS;
#pragma unroll
            for (int q = 0; q < PER; ++q)
                S[(kk0 + KS * q) * STRIDE + i] = r[q];
        }
        else {
            const int kk = t & 15, j0 = t >> 4;
#pragma unroll
            for (int q = 0; q < PER; ++q)
                S[(j0 + 16 * q) * LDS_K + kk] = r[q];
        }
    }
    // operand element: row `row` of the tile, k index kq
    static __device__ __forceinline__ double at(const double* __restrict__ S, int row, int kq)
    {
        if (!KMAJOR)
            return S[kq * STRIDE + row];
        else
            return S[row * LDS_K + kq];
    }
};

template <int TM, int TN, bool AK, bool BK>
__global__ __launch_bounds__(256) void k_gemm4(GemmArgs g)
{
    using SA = Stager<AK, TM>;
    using SB = Stager<BK, TN>;
    constexpr int RA = TM / 2 / 16; // 16-row slabs per wave
    constexpr int RB = TN / 2 / 4;  // 4-column groups per wave
    __shared__ __attribute__((aligned(16))) double lds[2][SA::ELEMS + SB::ELEMS];
    // XCD-aware tile order: workgroups b, b+8, b+16.. land on the same XCD (private L2), so give
    // each XCD a contiguous band of tiles: neighbours share operand panels in that L2.
    const int tiles_m = (int)((g.m + TM - 1) / TM);
    const int tiles_n = (int)((g.n + TN - 1) / TN);
    const int nwg = tiles_m * tiles_n;
    int wg = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ti = wg % tiles_m; // row tile fastest: neighbours share the B (column) panel
    const int tj = wg / tiles_m;
    const int64_t row0 = (int64_t)ti * TM, col0 = (int64_t)tj * TN;
    if (g.tri && (g.grow0 + row0 + TM - 1 < g.gcol0 + col0))
        return; // tile entirely above the diagonal
    int64_t kbeg = 0;
    if (g.ktri) { // X^T X with X lower triangular: rows of X below max(row0, col0) only
        kbeg = (row0 > col0 ? row0 : col0);
        kbeg -= kbeg % BKT;
    }
    const int64_t kend = g.k;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = (wave & 1) * (TM / 2), wn = (wave >> 1) * (TN / 2);
    const int arow = wm + (lane & 15); // + 16 m
    const int bcol = wn + (lane & 3);  // + 4 n
    const int kq = lane >> 4;

    const double* Ap = AK ? g.A + kbeg + row0 * g.lda : g.A + row0 + kbeg * g.lda;
    const double* Bp = BK ? g.B + kbeg + col0 * g.ldb : g.B + col0 + kbeg * g.ldb;
    const int64_t a_kstep = AK ? (int64_t)BKT : (int64_t)BKT * g.lda;
    const int64_t b_kstep = BK ? (int64_t)BKT : (int64_t)BKT * g.ldb;
    const int64_t mrows = g.m - row0, ncols = g.n - col0;

    double acc[RA][RB];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < RB; ++b)
            acc[a][b] = 0.0;

    SA sa;
    SB sb;
    int64_t kleft = kend - kbeg;
    if (kleft > 0) {
        sa.load(Ap, g.lda, mrows, kleft);
        sb.load(Bp, g.ldb, ncols, kleft);
        sa.store(lds[0]);
        sb.store(lds[0] + SA::ELEMS);
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
        const double* As = lds[buf];
        const double* Bs = lds[buf] + SA::ELEMS;
#pragma unroll
        for (int ks = 0; ks < BKT; ks += 4) {
            double af[RA], bf[RB];
#pragma unroll
            for (int x = 0; x < RA; ++x)
                af[x] = SA::at(As, arow + 16 * x, ks + kq);
#pragma unroll
            for (int x = 0; x < RB; ++x)
                bf[x] = SB::at(Bs, bcol + 4 * x, ks + kq);
#pragma unroll
            for (int n = 0; n < RB; ++n)
#pragma unroll
                for (int m = 0; m < RA; ++m)
                    acc[m][n] = mfma4(af[m], bf[n], acc[m][n]);
        }
        if (more) {
            sa.store(lds[buf ^ 1]);
            sb.store(lds[buf ^ 1] + SA::ELEMS);
        }
        __syncthreads();
        buf ^= 1;
    }

    // epilogue.  D lane l -> row 4*((l>>2)&3) + (l>>4) of the 16-row slab, column l&3 of the group
    const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4);
    const int dcol = lane & 3;
#pragma unroll
    for (int n = 0; n < RB; ++n) {
        const int64_t cn = col0 + wn + 4 * n + dcol;
        if (cn >= g.n)
            continue;
        double* Cc = g.C + cn * g.ldc;
#pragma unroll
        for (int m = 0; m < RA; ++m) {
            const int64_t rm = row0 + wm + 16 * m + drow;
            if (rm < g.m) {
                if (g.overwrite)
                    Cc[rm] = acc[m][n];
                else
                    Cc[rm] -= acc[m][n];
            }
        }
    }
}

template <int TM, int TN>
static void launch_tile(hipStream_t s, const GemmArgs& g)
{
    int64_t tiles = ((g.m + TM - 1) / TM) * ((g.n + TN - 1) / TN);
    dim3 grid((unsigned)tiles), block(256);
    if (!g.a_kmajor && !g.b_kmajor)
        hipLaunchKernelGGL((k_gemm4<TM, TN, false, false>), grid, block, 0, s, g);
    else if (!g.a_kmajor && g.b_kmajor)
        hipLaunchKernelGGL((k_gemm4<TM, TN, false, true>), grid, block, 0, s, g);
    else if (g.a_kmajor && !g.b_kmajor)
        hipLaunchKernelGGL((k_gemm4<TM, TN, true, false>), grid, block, 0, s, g);
    else
        hipLaunchKernelGGL((k_gemm4<TM, TN, true, true>), grid, block, 0, s, g);
}

// number of TM x TN tiles that do work (triangular skipping accounted for)
static int64_t live_tiles(const GemmArgs& g, int TM, int TN)
{
    const int64_t tm = (g.m + TM - 1) / TM, tn = (g.n + TN - 1) / TN;
    if (!g.tri)
        return tm * tn;
    int64_t cnt = 0;
    for (int64_t tj = 0; tj < tn; ++tj) {
        // tile (ti, tj) is live iff grow0 + ti*TM + TM - 1 >= gcol0 + tj*TN
        int64_t need = g.gcol0 + tj * TN - g.grow0 - (TM - 1); // ti*TM >= need
        int64_t ti0 = need <= 0 ? 0 : (need + TM - 1) / TM;
        if (ti0 < tm)
            cnt += tm - ti0;
    }
    return cnt;
}

void launch_gemm_sub(hipStream_t s, const GemmArgs& g)
{
    if (g.m <= 0 || g.n <= 0)
        return;
    if (g.k <= 0 && !g.overwrite)
        return;
    static int force = -1;
    if (force < 0) {
        const char* e = getenv("GPE_GEMM_TILE"); // debug/tuning: 128, 64 or 32
        force = e ? atoi(e) : 0;
    }
    int tile = g.tile ? g.tile : force;
    if (tile != 128 && tile != 64 && tile != 32) {
        // 256 CUs, 2 workgroups resident per CU at the 128 tile: want >= ~2 full rounds before
        // paying for the larger tile's longer per-workgroup latency
        if (live_tiles(g, 128, 128) >= 1024)
            tile = 128;
        else if (live_tiles(g, 64, 64) >= 512)
            tile = 64;
        else
            tile = 32;
    }
    if (tile == 128)
        launch_tile<128, 128>(s, g);
    else if (tile == 64)
        launch_tile<64, 64>(s, g);
    else
        launch_tile<32, 64>(s, g);
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
