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
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include <algorithm>

#ifdef GEMM_TIMING
__device__ long long g_gemm_ts[16];
#define GTS(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && lwg == (int)blockIdx.x) g_gemm_ts[i] = clock64(); } while (0)
__device__ long long g_gemm64_ts[24];
#define GTS64(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && lwg == (int)blockIdx.x) g_gemm64_ts[i] = clock64(); } while (0)
#else
#define GTS(i) do { } while (0)
#define GTS64(i) do { } while (0)
#endif
#ifdef GEMM_TIMING
#define GTS64_(i) GTS64(i)
#endif
#include "gemm_glds64.h"
#ifndef GEMM_ABL
#define GEMM_ABL 0 // debug ablations (tools/kbench): 1 = no global loads / LDS stores in the k loop, 2 = no MFMA
#endif


template <bool KMAJOR, int ROWS, int BK>
struct Stager {
    static constexpr int PER = ROWS * BK / 256;               // elements per thread
    static constexpr int STRIDE = ROWS + 16;                  // S[kk][i] row stride (== 16 mod 32)
    static constexpr int KSTR = BK + 2;                       // S[i][kk] row stride
    static constexpr int ELEMS = KMAJOR ? ROWS * KSTR : BK * STRIDE;
    double r[PER];
    // global -> registers.  No conditional loads: hipcc turns `cond ? P[..] : 0` into a branch
    // around every load with its own wait (dependent L2 round trips).  Instead the addresses are
    // clamped into the valid range; rows >= rl then carry a copy of row rl-1 and only feed output
    // rows/columns the epilogue never stores, and k >= kl is cancelled by multiplying the A
    // operand (MASKK) with an exact 0/1 factor.
    template <bool MASKK>
    __device__ __forceinline__ void load(const double* __restrict__ P, int64_t ld, int rl, int kl)
    {
        const int t = threadIdx.x;
        if (!KMAJOR) {
            const int i = t % ROWS, kk0 = t / ROWS;
            constexpr int KS = 256 / ROWS;
            const int ic = i < rl ? i : rl - 1;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int kk = kk0 + KS * q;
                const int kc = kk < kl ? kk : kl - 1;
                const double v = P[ic + (int64_t)kc * ld];
                r[q] = MASKK ? v * (kk < kl ? 1.0 : 0.0) : v;
            }
        }
        else {
            const int kk = t % BK, j0 = t / BK;
            constexpr int JS = 256 / BK;
            const int kc = kk < kl ? kk : kl - 1;
            const double km = kk < kl ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int j = j0 + JS * q;
                const int jc = j < rl ? j : rl - 1;
                const double v = P[kc + (int64_t)jc * ld];
                r[q] = MASKK ? v * km : v;
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
            const int kk = t % BK, j0 = t / BK;
            constexpr int JS = 256 / BK;
#pragma unroll
            for (int q = 0; q < PER; ++q)
                S[(j0 + JS * q) * KSTR + kk] = r[q];
        }
    }
    // operand element: row `row` of the tile, k index kq
    static __device__ __forceinline__ double at(const double* __restrict__ S, int row, int kq)
    {
        if (!KMAJOR)
            return S[kq * STRIDE + row];
        else
            return S[row * KSTR + kq];
    }
};

// NBUF = 2: double-buffered k loop.  NBUF = 1: the whole k range (<= BKT) is staged at once — the
// latency-critical one-shot form used by the panel steps of the Cholesky (k = 64): every global
// load of the workgroup, including its C tile, is in flight before the first wait.
template <int TM, int TN, int BKT, int NBUF, bool AK, bool BK, bool BATCH = false>
__global__ __launch_bounds__(256) void k_gemm4(GemmArgs g_)
{
    // BATCH (gridDim.z GPs): a rebased copy of the arguments; otherwise the kernel argument itself (the round-1 kernel)
    GemmArgs gb;
    if (BATCH) {
        gb = g_;
        gemm_rebase(gb);
    }
    const GemmArgs& g = BATCH ? gb : g_;
    using SA = Stager<AK, TM, BKT>;
    using SB = Stager<BK, TN, BKT>;
    constexpr int RA = TM / 2 / 16; // 16-row slabs per wave
    constexpr int RB = TN / 2 / 4;  // 4-column groups per wave
    __shared__ __attribute__((aligned(16))) double lds[NBUF][SA::ELEMS + SB::ELEMS];
    // Workgroup -> tile.  Two concerns:
    //  * triangular problems: only tiles touching the lower triangle are enumerated, and column
    //    tj is folded with column tiles_n-1-tj so that every "super column" has the same number of
    //    live tiles — a plain 2-D grid with early exits leaves the XCDs that draw the right-hand
    //    columns idle while XCD 0 works through three rounds;
    //  * XCD-aware order: workgroups b, b+8, b+16.. land on the same XCD (private L2), so each XCD
    //    gets a contiguous band of the enumeration: neighbours share operand panels in that L2.
    const int tiles_m = (int)((g.m + TM - 1) / TM);
    const int tiles_n = (int)((g.n + TN - 1) / TN);
    int wg = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int ti, tj;
    if (g.tri) {
        const int sc = wg / g.fold_len;
        int rr = wg % g.fold_len;
        const int t0 = first_live_tile<TM, TN>(g, sc), c0 = tiles_m - t0;
        if (rr < c0) {
            tj = sc;
            ti = t0 + rr;
        }
        else {
            tj = tiles_n - 1 - sc;
            const int t1 = first_live_tile<TM, TN>(g, tj);
            rr -= c0;
            if (tj == sc || rr >= tiles_m - t1)
                return;
            ti = t1 + rr;
        }
    }
    else {
        ti = wg % tiles_m; // row tile fastest: neighbours share the B (column) panel
        tj = wg / tiles_m;
    }
    const int64_t row0 = (int64_t)ti * TM, col0 = (int64_t)tj * TN;
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
    const int mr = (int)(mrows < TM ? mrows : TM), nc = (int)(ncols < TN ? ncols : TN);

    double acc[RA][RB];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < RB; ++b)
            acc[a][b] = 0.0;

    // C tile of this wave in the lane = row layout (WaveTileC); rows/columns of the wave tile inside the matrix
    using WT = WaveTileC<RA, RB>;
    double* Cw = g.C + (col0 + wn) * g.ldc + row0 + wm;
    const int rlim = mr - wm < WT::R ? mr - wm : WT::R, clim = nc - wn < WT::CN ? nc - wn : WT::CN;
    const bool live = rlim > 0 && clim > 0; // a ragged tile can leave a wave without any output
    constexpr bool CPRE = (NBUF == 1); // small one-shot tiles: fetch C together with the operands
    double cv[WT::NIT];
    if (CPRE && live && g.overwrite != 1)
        WT::load(cv, Cw, g.ldc, rlim, clim, lane);

    SA sa;
    SB sb;
    int64_t kleft = kend - kbeg;
    if (kleft > 0) {
        const int kl = (int)(kleft < BKT ? kleft : BKT);
        sa.template load<true>(Ap, g.lda, mr, kl);
        sb.template load<false>(Bp, g.ldb, nc, kl);
        sa.store(lds[0]);
        sb.store(lds[0] + SA::ELEMS);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BKT) {
#if GEMM_ABL != 1
        const bool more = (NBUF == 2) && (k0 + BKT < kend);
#endif
#if GEMM_ABL == 1
        const bool more = false;
#endif
        if (more) { // prefetch next k-tile into registers while this one is consumed
            Ap += a_kstep;
            Bp += b_kstep;
            const int64_t kleft2 = kend - k0 - BKT;
            const int kl = (int)(kleft2 < BKT ? kleft2 : BKT);
            sa.template load<true>(Ap, g.lda, mr, kl);
            sb.template load<false>(Bp, g.ldb, nc, kl);
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
#if GEMM_ABL == 2
                    acc[m][n] += af[m] * bf[n] * (ks == 0 && m == 0 && n == 0 ? 1.0 : 0.0);
#else
                    acc[m][n] = mfma4(af[m], bf[n], acc[m][n]);
#endif
        }
        if (NBUF == 2) {
            if (more) {
                sa.store(lds[buf ^ 1]);
                sb.store(lds[buf ^ 1] + SA::ELEMS);
            }
            __syncthreads();
            buf ^= 1;
        }
    }

    static_assert(4 * WT::SCRATCH <= NBUF * (SA::ELEMS + SB::ELEMS), "transposition scratch must fit in the operand stages");
    if (!CPRE && live && g.overwrite != 1)
        WT::load(cv, Cw, g.ldc, rlim, clim, lane);
    // the operand stages are dead (the double-buffered k loop ends with a barrier, the one-shot form
    // needs one here): each wave takes a private slice of them as its transposition scratch
    if (NBUF == 1)
        __syncthreads();
    if (live)
        WT::store(acc, cv, &lds[0][0] + wave * WT::SCRATCH, Cw, g.ldc, rlim, clim, g.overwrite, lane);
}

// Launch with the dispatch's own completion signal as an event when the caller asked for one
// (GemmArgs::stop_event): a separate hipEventRecord is a marker packet of its own on the stream and
// costs ~6 us on a dependent chain (rocprofv3 timeline of the factorisation).
template <typename K>
static void launch_k_named(const char* name, K kern_single, K kern_batched, dim3 grid, dim3 block, hipStream_t s, const GemmArgs& g0)
{
    GemmArgs g = g0;
    g.bt = g_batch.bt; // batched launch: gridDim.z GPs, pointers rebased per GP in the kernel (dev.h)
    grid.z = (unsigned)g_batch.G;
    K kern = g.bt ? kern_batched : kern_single;
    if (g.stop_event)
        GPE_LAUNCH_STOP(name, kern, grid, block, 0, s, (hipEvent_t)g.stop_event, g);
    else
        GPE_LAUNCH_NAMED(name, kern, grid, block, 0, s, g);
}
// (the kernel's name for the launch trace: the first argument as written)
#define launch_k(ks, ...) launch_k_named(#ks, ks, __VA_ARGS__)
template <int TM, int TN, int BKT, int NBUF>
static void launch_tile(hipStream_t s, const GemmArgs& g0)
{
    GemmArgs g = g0;
    const int tiles_m = (int)((g.m + TM - 1) / TM), tiles_n = (int)((g.n + TN - 1) / TN);
    int64_t tiles = (int64_t)tiles_m * tiles_n;
    if (g.tri) { // folded enumeration of the live tiles (see k_gemm4)
        int fold = 1;
        const int nsup = (tiles_n + 1) / 2;
        for (int sc = 0; sc < nsup; ++sc) {
            const int t2 = tiles_n - 1 - sc;
            int len = tiles_m - first_live_tile<TM, TN>(g, sc);
            if (t2 != sc)
                len += tiles_m - first_live_tile<TM, TN>(g, t2);
            fold = len > fold ? len : fold;
        }
        g.fold_len = fold;
        tiles = (int64_t)nsup * fold;
    }
    dim3 grid((unsigned)tiles), block(256);
    if (!g.a_kmajor && !g.b_kmajor)
        launch_k(k_gemm4<TM, TN, BKT, NBUF, false, false>, k_gemm4<TM, TN, BKT, NBUF, false, false, true>, grid, block, s, g);
    else if (!g.a_kmajor && g.b_kmajor)
        launch_k(k_gemm4<TM, TN, BKT, NBUF, false, true>, k_gemm4<TM, TN, BKT, NBUF, false, true, true>, grid, block, s, g);
    else if (g.a_kmajor && !g.b_kmajor)
        launch_k(k_gemm4<TM, TN, BKT, NBUF, true, false>, k_gemm4<TM, TN, BKT, NBUF, true, false, true>, grid, block, s, g);
    else
        launch_k(k_gemm4<TM, TN, BKT, NBUF, true, true>, k_gemm4<TM, TN, BKT, NBUF, true, true, true>, grid, block, s, g);
}

// ---------------------------------------------------------------------------------------------
// Fast path: both operands row-contiguous (the Cholesky updates), k a multiple of BKT.
// Operand k-rows go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave
// instruction = one k-row of a 128-row operand tile), no staging VGPRs, no ds_write pass and no
// per-load address arithmetic; NST LDS stages, counted vmcnt and raw s_barrier so that the next
// stage's loads stay in flight across the barriers; NWV waves (2 per SIMD at 8) so that one wave's
// LDS reads and waits sit under the other's MFMAs.
// ---------------------------------------------------------------------------------------------

template <int TM, int TN, int WM, int WN, int BKT, int NST, int EPC = 0, int MINB = 1, bool BATCH = false>
__global__ __launch_bounds__(WM* WN * 64, MINB) void k_gemm_glds(GemmArgs g_)
{
    GemmArgs gb;
    if (BATCH) {
        gb = g_;
        gemm_rebase(gb);
    }
    const GemmArgs& g = BATCH ? gb : g_;
    static_assert(TM == 128 && TN == 128, "one k-row of an operand tile = one 1 KiB glds instruction");
    constexpr int NWV = WM * WN;
    constexpr int SA = TM + 16, SB = TN + 16; // k-row strides (doubles), == 16 mod 32
    constexpr int STAGE = BKT * (SA + SB);
    constexpr int RA = TM / WM / 16, RB = TN / WN / 4;
    constexpr int LPW = 2 * BKT / NWV; // glds instructions per wave per stage
    __shared__ __attribute__((aligned(16))) double lds[NST * STAGE];

    const int tiles_m = (int)((g.m + TM - 1) / TM);
    const int tiles_n = (int)((g.n + TN - 1) / TN);
    if (g.rhs_rows > 0) // the right-hand-side rows under the matrix: this workgroup's n / G columns of them (gemm_glds64.h)
        gemm_rhs_rows<64 * NWV, MINB == 1>(g, g.rhs_rows, (int)gridDim.x, lds);
    // g.total logical workgroups; a launch with fewer physical ones (gridDim.x < g.total: the
    // look-ahead update, which must leave CUs free for the panel on the other stream) loops.
    for (int lwg = blockIdx.x; lwg < g.total; lwg += gridDim.x) {
    int wg = lwg;
    {
        const int nwg = g.total;
        const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int ti, tj;
    if (g.tile_map) { // (host-built order: tri_tile_map below)
        const int pk = g.tile_map[lwg];
        if (pk < 0)
            continue;
        ti = pk & 0xffff;
        tj = pk >> 16;
    }
    else if (g.tri) {
        const int sc = wg / g.fold_len;
        int rr = wg % g.fold_len;
        const int t0 = first_live_tile<TM, TN>(g, sc), c0 = tiles_m - t0;
        if (rr < c0) {
            tj = sc;
            ti = t0 + rr;
        }
        else {
            tj = tiles_n - 1 - sc;
            const int t1 = first_live_tile<TM, TN>(g, tj);
            rr -= c0;
            if (tj == sc || rr >= tiles_m - t1)
                continue;
            ti = t1 + rr;
        }
    }
    else {
        ti = wg % tiles_m;
        tj = wg / tiles_m;
    }
    const int64_t row0 = (int64_t)ti * TM, col0 = (int64_t)tj * TN;
    const int64_t mrows = g.m - row0, ncols = g.n - col0;
    const int mr = (int)(mrows < TM ? mrows : TM), nc = (int)(ncols < TN ? ncols : TN);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave % WM) * (TM / WM), wn = (wave / WM) * (TN / WN);
    const int arow = wm + (lane & 15), bcol = wn + (lane & 3), kq = lane >> 4;

    // this lane's 16-byte piece of every k-row: rows (2 lane, 2 lane + 1), clamped into the tile's
    // valid rows (copies of valid rows only feed outputs the epilogue does not store)
    int ra = 2 * lane, rb = 2 * lane;
    {
        const int ma = (mr - 1) & ~1, mb = (nc - 1) & ~1;
        ra = ra < ma ? ra : ma;
        rb = rb < mb ? rb : mb;
    }
    // ktri (U U^T with U upper triangular, K^-1 = L^-T L^-1): the k range of a tile starts at max(row0, col0), which is a
    // multiple of the tile edge and hence of BKT
    const int64_t kbeg = g.ktri ? (row0 > col0 ? row0 : col0) : 0;
    const double* pa = g.A + row0 + ra + ((int64_t)wave + kbeg) * g.lda;
    const double* pb = g.B + col0 + rb + ((int64_t)wave + kbeg) * g.ldb;
    const int64_t astep = (int64_t)NWV * g.lda, bstep = (int64_t)NWV * g.ldb;

    auto issue = [&](int stage) {
        double* sa = lds + stage * STAGE + wave * SA;
        double* sb = lds + stage * STAGE + BKT * SA + wave * SB;
#pragma unroll
        for (int q = 0; q < BKT / NWV; ++q) {
            __builtin_amdgcn_global_load_lds(pa, (lds_void_t*)(sa + q * NWV * SA), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(pb, (lds_void_t*)(sb + q * NWV * SB), 16, 0, 0);
            pa += astep;
            pb += bstep;
        }
    };

    double acc[RA][RB];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < RB; ++b)
            acc[a][b] = 0.0;

    const int nk = (int)((g.k - kbeg) / BKT);
    GTS(0);
    issue(0);
    for (int t = 0; t < nk; ++t) {
        const int st = t % NST;
        if (t + 1 < nk) {
            issue((t + 1) % NST);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
        }
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier(); // every wave's pieces of stage st have landed
        if (t == 0)
            GTS(1);
        if (t == 1)
            GTS(2);
        if (t == 2)
            GTS(3);
        const double* As = lds + st * STAGE;
        const double* Bs = As + BKT * SA;
#pragma unroll
        for (int ks = 0; ks < BKT; ks += 4) {
            double af[RA], bf[RB];
#pragma unroll
            for (int x = 0; x < RA; ++x)
                af[x] = As[(ks + kq) * SA + arow + 16 * x];
#pragma unroll
            for (int x = 0; x < RB; ++x)
                bf[x] = Bs[(ks + kq) * SB + bcol + 4 * x];
#pragma unroll
            for (int n = 0; n < RB; ++n)
#pragma unroll
                for (int m = 0; m < RA; ++m)
                    acc[m][n] = mfma4(af[m], bf[n], acc[m][n]);
        }
        if (t == 1)
            GTS(6);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier(); // stage st may be refilled
        if (t == 1)
            GTS(7);
    }
    GTS(4);

    {
        // C traffic in the lane = row layout (WaveTileC).  Fetching the tile under the k loop instead
        // was measured neutral to slightly worse (its loads sit in front of the stage loads in the
        // in-order vector-memory counter).
        using WT = WaveTileC<RA, RB>;
        constexpr int SCR = EPC > 0 ? WT::SCRATCH / EPC : WT::SCRATCH;
        static_assert(NWV * SCR <= NST * STAGE, "transposition scratch must fit in the operand stages");
        double* Cw = g.C + (col0 + wn) * g.ldc + row0 + wm;
        const int rlim = mr - wm < WT::R ? mr - wm : WT::R, clim = nc - wn < WT::CN ? nc - wn : WT::CN;
        if (rlim > 0 && clim > 0) {
            // the k loop ended with a barrier: the stages are free, each wave uses a private slice
            if constexpr (EPC > 0)
                WT::template rmw_chunked<EPC>(acc, lds + wave * SCR, Cw, g.ldc, rlim, clim, g.overwrite, lane);
            else {
                double cv[WT::NIT];
                if (g.overwrite != 1)
                    WT::load(cv, Cw, g.ldc, rlim, clim, lane);
                WT::store(acc, cv, lds + wave * WT::SCRATCH, Cw, g.ldc, rlim, clim, g.overwrite, lane);
            }
        }
    }
    __syncthreads(); // the next tile's prologue overwrites the LDS stages
    GTS(5);
    } // logical workgroups
}
#ifdef GEMM_TIMING
void dump_gemm_timing()
{
    {
        long long q[24];
        (void)hipMemcpyFromSymbol(q, HIP_SYMBOL(g_gemm64_ts), sizeof(q));
        printf("k_gemm_glds64 WG0 first tile, cycles: start->k-tile 0 begins %lld | k-tiles:", q[1] - q[0]);
        for (int t = 1; t < 16; ++t)
            printf(" %lld", q[1 + t] - q[t]);
        printf(" %lld | epilogue %lld | total %lld\n", q[17] - q[16], q[18] - q[17], q[18] - q[0]);
    }
    long long h[16];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_ts), sizeof(h));
    printf("k_gemm_glds WG0 first tile, cycles: prologue (first stage landed) %lld | k-tile 0 %lld | k-tile 1 %lld (mfma part %lld, closing barrier %lld) | whole k loop %lld | epilogue %lld | total %lld\n",
           h[1] - h[0], h[2] - h[1], h[3] - h[2], h[6] - h[2], h[7] - h[6], h[4] - h[1], h[5] - h[4], h[5] - h[0]);
    printf("  epilogue: issue 32 loads %lld | loads complete %lld | issue 32 stores %lld | stores complete %lld | tail %lld\n", h[8] - h[4],
           h[9] - h[8], h[10] - h[9], h[11] - h[10], h[5] - h[11]);
}
#endif

template <int BKT, int NST, int MINB, int NWV, bool BATCH = false>
__global__ __launch_bounds__(64 * NWV, MINB) void k_gemm_glds64(GemmArgs g_)
{
    GemmArgs gb;
    if (BATCH) {
        gb = g_;
        gemm_rebase(gb);
    }
    const GemmArgs& g = BATCH ? gb : g_;
    __shared__ __attribute__((aligned(16))) double lds[NST * Glds64Shape<BKT>::STAGE];
    gemm_glds64_body<BKT, NST, NWV>(g, lds, (int)blockIdx.x, (int)gridDim.x, false);
}

static void launch_glds64(hipStream_t s, const GemmArgs& g0)
{
    constexpr int TM = 64, TN = 64;
    GemmArgs g = g0;
    const int tiles_m = (int)((g.m + TM - 1) / TM), tiles_n = (int)((g.n + TN - 1) / TN);
    int64_t tiles = (int64_t)tiles_m * tiles_n;
    if (g.tri) {
        int fold = 1;
        const int nsup = (tiles_n + 1) / 2;
        for (int sc = 0; sc < nsup; ++sc) {
            const int t2 = tiles_n - 1 - sc;
            int len = tiles_m - first_live_tile<TM, TN>(g, sc);
            if (t2 != sc)
                len += tiles_m - first_live_tile<TM, TN>(g, t2);
            fold = len > fold ? len : fold;
        }
        g.fold_len = fold;
        tiles = (int64_t)nsup * fold;
    }
    g.total = (int)tiles;
    if (g.grid_limit > 0 && tiles > g.grid_limit)
        tiles = g.grid_limit;
    // Two shapes.  With more tiles than CUs, two workgroups share a CU (BKT 16, 4 stages, 64 KB) and fill each
    // other's bubbles.  With at most one tile per CU — the next-panel update on the critical path of the
    // factorisation — a workgroup is alone with one wave per SIMD and every k-tile pays its two barriers and the
    // LDS read latency in full (in-kernel stamps: 1944 cycles per 16-deep k-tile for 1024 cycles of MFMA work;
    // halving the number of k-tiles with BKT 32 changed nothing: it is the fragment-read latency before every
    // group of 16 MFMAs).  Eight waves on the tile (two per SIMD) cover each other.
    if ((int64_t)g.total * g_batch.G > 256 || g.grid_limit > 0)
        launch_k(k_gemm_glds64<16, 4, 2, 4>, k_gemm_glds64<16, 4, 2, 4, true>, dim3((unsigned)tiles), dim3(256), s, g);
    else
        launch_k(k_gemm_glds64<16, 4, 1, 8>, k_gemm_glds64<16, 4, 1, 8, true>, dim3((unsigned)tiles), dim3(512), s, g);
}

static bool glds_ok(const GemmArgs& g)
{
    // (ktri — the U U^T product of K^-1 — is served by the register-staged kernel: through this one, which supports it, the
    // product measured 12 % slower: tiles with k ranges from N down to 128 do not balance over two-per-CU workgroups)
    return !g.a_kmajor && !g.b_kmajor && !g.ktri && g.k >= 32 && g.k % 32 == 0 && (g.lda % 2) == 0 && (g.ldb % 2) == 0
        && ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0;
}

// ---- which workgroup takes which tile of a triangular update (round 6) ------------------------------------------------------
// Workgroup b runs on XCD b mod 8, each XCD has its own L2, and the workgroups resident on an XCD walk their k ranges together:
// what an XCD fetches from memory is the set of DISTINCT operand panels (128 rows x k) of its resident tiles.  The folded
// column enumeration (k_gemm4) gives an XCD a tile column and a half — one B panel, some thirty A panels: 139 panel fetches
// for the 253 tiles of N = 4096's update (the PMC's 308 MB, 3.35 x the algorithmic bytes).  Here: the live tiles in bands of
// TRI_BAND tile rows, column by column inside a band, the sequence cut into eight runs, run x dealt to the workgroups
// x, x + 8, .. — 32 consecutive tiles are a 8 x 4 block, 12 panels, fewer where the band meets the diagonal (A and B are the
// same matrix there): 91 panel fetches.  The table is built once per shape and device and kept.
#define TRI_BAND 8
struct TriMapKey {
    int dev, tiles_m, tiles_n;
    int64_t off; // (gcol0 - grow0: where the diagonal runs)
    bool operator<(const TriMapKey& o) const
    {
        return std::tie(dev, tiles_m, tiles_n, off) < std::tie(o.dev, o.tiles_m, o.tiles_n, o.off);
    }
};
// host part (also the test hook gpe_debug_tri_tile_map): the table for a launch of shape g; returns its length
static int tri_tile_map_host(const GemmArgs& g, std::vector<int>& tab)
{
    constexpr int TM = 128, TN = 128;
    const int tiles_m = (int)((g.m + TM - 1) / TM), tiles_n = (int)((g.n + TN - 1) / TN);
    std::vector<int> first(tiles_n);
    for (int tj = 0; tj < tiles_n; ++tj)
        first[tj] = first_live_tile<TM, TN>(g, tj);
    std::vector<int> seq;
    for (int r0 = 0; r0 < tiles_m; r0 += TRI_BAND) {
        const int r1 = std::min(tiles_m, r0 + TRI_BAND);
        for (int tj = 0; tj < tiles_n; ++tj)
            for (int ti = std::max(r0, first[tj]); ti < r1; ++ti)
                seq.push_back(ti | (tj << 16));
    }
    const int T = (int)seq.size(), q = T / 8, r = T % 8, slots = q + (r ? 1 : 0);
    tab.assign((size_t)slots * 8, -1);
    int pos = 0;
    for (int x = 0; x < 8; ++x) {
        const int len = q + (x < r ? 1 : 0);
        for (int i = 0; i < len; ++i)
            tab[(size_t)i * 8 + x] = seq[pos + i];
        pos += len;
    }
    return (int)tab.size();
}
int debug_tri_tile_map(int64_t m, int64_t n, int64_t grow0, int64_t gcol0, int* out, int cap)
{
    GemmArgs g{};
    g.m = m;
    g.n = n;
    g.tri = 1;
    g.grow0 = grow0;
    g.gcol0 = gcol0;
    std::vector<int> tab;
    const int len = tri_tile_map_host(g, tab);
    for (int i = 0; i < len && i < cap; ++i)
        out[i] = tab[i];
    return len;
}
static const int* tri_tile_map(const GemmArgs& g, int* total)
{
    static std::mutex mu;
    static std::map<TriMapKey, std::pair<int*, int>> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const TriMapKey key{dev, (int)((g.m + 127) / 128), (int)((g.n + 127) / 128), g.gcol0 - g.grow0};
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it == cache.end()) {
        std::vector<int> tab;
        const int len = tri_tile_map_host(g, tab);
        int* d = nullptr;
        if (hipMalloc(&d, sizeof(int) * (size_t)len) != hipSuccess)
            return nullptr;
        if (hipMemcpy(d, tab.data(), sizeof(int) * (size_t)len, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d);
            return nullptr;
        }
        it = cache.emplace(key, std::make_pair(d, len)).first;
    }
    *total = it->second.second;
    return it->second.first;
}

static void launch_glds128(hipStream_t s, const GemmArgs& g0)
{
    constexpr int TM = 128, TN = 128;
    GemmArgs g = g0;
    const int tiles_m = (int)((g.m + TM - 1) / TM), tiles_n = (int)((g.n + TN - 1) / TN);
    int64_t tiles = (int64_t)tiles_m * tiles_n;
    g.tile_map = nullptr;
    static const bool tri_map = !(getenv("GPE_TRI_MAP") && atoi(getenv("GPE_TRI_MAP")) == 0);
    int mapped = 0;
    // (not for batched launches: measured slower there — 8 x 2048 6.44 against 6.70 k evaluations/s, 64 x 2048 8.53 against 8.80:
    // the members' small updates are dealt over the chip by gridDim.z, a member's few tiles gain nothing from the order and lose
    // the padding to whole rounds of eight)
    if (g.tri && tri_map && !g_batch.bt && tiles_m < 65536 && tiles_n < 32768)
        g.tile_map = tri_tile_map(g, &mapped);
    if (g.tile_map)
        tiles = mapped;
    else if (g.tri) {
        int fold = 1;
        const int nsup = (tiles_n + 1) / 2;
        for (int sc = 0; sc < nsup; ++sc) {
            const int t2 = tiles_n - 1 - sc;
            int len = tiles_m - first_live_tile<TM, TN>(g, sc);
            if (t2 != sc)
                len += tiles_m - first_live_tile<TM, TN>(g, t2);
            fold = len > fold ? len : fold;
        }
        g.fold_len = fold;
        tiles = (int64_t)nsup * fold;
    }
    g.total = (int)tiles;
    if (g.grid_limit > 0 && tiles > g.grid_limit)
        tiles = g.grid_limit;
    // Two shapes of the same kernel.  BKT 32 / 147 KB of LDS: one workgroup per CU, fewest barriers — best
    // when there is at most one tile per CU, and the only choice for the look-ahead update (grid_limit:
    // the free CUs must stay free).  BKT 16 / 74 KB / <= 128 VGPRs: two workgroups per CU, so one tile's
    // prologue, epilogue and barrier stalls are covered by the other's MFMAs — 11 % faster on the
    // 465-tile update of N = 4096 (94 -> 84 us), 2-4 % slower when tiles <= CUs.
    const bool two_per_cu = g.grid_limit <= 0 && (int64_t)g.total * g_batch.G > 256;
    // (ONE workgroup of 16 waves per CU — 2 x 8 waves of 64 x 16, or 4 x 4 of 32 x 32: four waves per SIMD from one barrier
    // group — was measured too: a lone 128 x 128 x 256 tile takes 47.9 us against 47.3 with 8 waves, 4 x 4 is 15 % slower.
    // What reaches 97 % of the matrix-core peak in the k loop is two INDEPENDENT barrier groups per CU: profiles/r03_sk_study.md)
    if (two_per_cu)
        launch_k(k_gemm_glds<128, 128, 2, 4, 16, 2, 4, 2>, k_gemm_glds<128, 128, 2, 4, 16, 2, 4, 2, true>, dim3((unsigned)tiles), dim3(512), s, g);
    else
        launch_k(k_gemm_glds<128, 128, 2, 4, 32, 2>, k_gemm_glds<128, 128, 2, 4, 32, 2, 0, 1, true>, dim3((unsigned)tiles), dim3(512), s, g);
}

// ---------------------------------------------------------------------------------------------
// A LIST of 128 x 128 tile products, each with its own operands and depth (dev.h: GemmItem) — the recursive K^-1
// (inv2.hip).  The k loop is k_gemm_glds's in its two-workgroups-per-CU shape (BKT 16, two LDS stages, 74 KB, <= 128
// VGPRs: two independent barrier groups per CU run the loop at the matrix-core rate, profiles/r03_sk_study.md); what
// differs is where the work comes from: workgroup b walks items bin_start[b] .. bin_start[b + 1] of a host-built
// list (longest first, shares of equal length), every tile is full, C is overwritten (never read).
// ---------------------------------------------------------------------------------------------
template <int BKT, int NST, int EPC, int MINB>
__global__ __launch_bounds__(512, MINB) void k_gemm_items(const GemmItem* __restrict__ items, const int32_t* __restrict__ bin_start,
                                                          int64_t ld, const BatchTab* __restrict__ bt, int* __restrict__ counters, int nslots,
                                                          int64_t pstride)
{
    constexpr int TM = 128, TN = 128, WM = 2, WN = 4, NWV = WM * WN;
    constexpr int SA = TM + 16, SB = TN + 16;
    constexpr int STAGE = BKT * (SA + SB);
    constexpr int RA = TM / WM / 16, RB = TN / WN / 4;
    constexpr int LPW = 2 * BKT / NWV;
    __shared__ __attribute__((aligned(16))) double lds[NST * STAGE];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave % WM) * (TM / WM), wn = (wave / WM) * (TN / WN);
    const int arow = wm + (lane & 15), bcol = wn + (lane & 3), kq = lane >> 4;
    const int i0 = bin_start[blockIdx.x], i1 = bin_start[blockIdx.x + 1];
    const int64_t step = (int64_t)NWV * ld;
    for (int it = i0; it < i1; ++it) {
        GemmItem item = items[it];
        if (bt) { // batched launch: the list holds member 0's pointers
            item.A = bt_rebase(bt, (int)blockIdx.z, item.A);
            item.B = bt_rebase(bt, (int)blockIdx.z, item.B);
            item.C = bt_rebase(bt, (int)blockIdx.z, item.C);
            if (item.T)
                item.T = bt_rebase(bt, (int)blockIdx.z, item.T);
            if (item.nch > 1) {
                item.D = bt_rebase(bt, (int)blockIdx.z, item.D);
                item.P1 = bt_rebase(bt, (int)blockIdx.z, item.P1);
            }
        }
        // this lane's 16-byte piece of every k-row: rows (2 lane, 2 lane + 1), clamped into the tile's valid rows (copies of
        // valid rows only feed outputs the epilogue does not store); wave w moves k-rows w, w + 8, ..
        const int mr = (item.flags >> 8) & 255, nc = (item.flags >> 16) & 255;
        int ra = 2 * lane, rb = 2 * lane;
        {
            const int ma = (mr - 1) & ~1, mb = (nc - 1) & ~1;
            ra = ra < ma ? ra : ma;
            rb = rb < mb ? rb : mb;
        }
        const double* pa = item.A + ra + (int64_t)wave * ld;
        const double* pb = item.B + rb + (int64_t)wave * ld;
        auto issue = [&](int stage) {
            double* sa = lds + stage * STAGE + wave * SA;
            double* sb = lds + stage * STAGE + BKT * SA + wave * SB;
#pragma unroll
            for (int q = 0; q < BKT / NWV; ++q) {
                __builtin_amdgcn_global_load_lds(pa, (lds_void_t*)(sa + q * NWV * SA), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(pb, (lds_void_t*)(sb + q * NWV * SB), 16, 0, 0);
                pa += step;
                pb += step;
            }
        };
        double acc[RA][RB];
#pragma unroll
        for (int a = 0; a < RA; ++a)
#pragma unroll
            for (int b = 0; b < RB; ++b)
                acc[a][b] = 0.0;
        const int nk = item.k / BKT;
        issue(0);
        for (int t = 0; t < nk; ++t) {
            const int st = t % NST;
            if (t + 1 < nk) {
                issue((t + 1) % NST);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
            }
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // every wave's pieces of stage st have landed
            const double* As = lds + st * STAGE;
            const double* Bs = As + BKT * SA;
#pragma unroll
            for (int ks = 0; ks < BKT; ks += 4) {
                double af[RA], bf[RB];
#pragma unroll
                for (int x = 0; x < RA; ++x)
                    af[x] = As[(ks + kq) * SA + arow + 16 * x];
#pragma unroll
                for (int x = 0; x < RB; ++x)
                    bf[x] = Bs[(ks + kq) * SB + bcol + 4 * x];
#pragma unroll
                for (int n = 0; n < RB; ++n)
#pragma unroll
                    for (int m = 0; m < RA; ++m)
                        acc[m][n] = mfma4(af[m], bf[n], acc[m][n]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // stage st may be refilled
        }
        if (item.flags & 1) {
#pragma unroll
            for (int a = 0; a < RA; ++a)
#pragma unroll
                for (int b = 0; b < RB; ++b)
                    acc[a][b] = -acc[a][b];
        }
        {
            using WT = WaveTileC<RA, RB>;
            constexpr int SCR = WT::SCRATCH / EPC;
            static_assert(NWV * SCR <= NST * STAGE, "transposition scratch must fit in the operand stages");
            // the k loop ended with a barrier: the stages are free, each wave uses a private slice
            const int rlim = mr - wm < WT::R ? mr - wm : WT::R, clim = nc - wn < WT::CN ? nc - wn : WT::CN;
            double* const Tw = item.T ? item.T + wn + (int64_t)wm * ld : nullptr;
            const bool cut = item.nch > 1; // one of the chunks of a cut k range (dev.h: GemmItem)
            if (rlim > 0 && clim > 0)
                WT::template store_item<EPC>(acc, lds + wave * SCR, item.C + (int64_t)wn * ld + wm, cut ? nullptr : Tw, ld, rlim, clim, cut, lane);
            if (cut) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's part of the chunk is acknowledged
                __syncthreads();
                if (threadIdx.x == 0)
                    s_last = (__hip_atomic_fetch_add(counters + item.slot + (int)blockIdx.z * nslots, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1)
                            % item.nch
                        == 0;
                __syncthreads();
                if (s_last && rlim > 0 && clim > 0) // every chunk of the tile is out: this workgroup adds them up
                    WT::template fold_item<EPC>(lds + wave * SCR, item.D + (int64_t)wn * ld + wm, item.P1 + (int64_t)wn * ld + wm, pstride,
                                                item.nch - 1, Tw, ld, rlim, clim, lane);
            }
        }
        __syncthreads(); // the next item's prologue overwrites the LDS stages
    }
}

// The same for 64 x 64 tiles: the low levels of the recursion have a few dozen 128 x 128 tiles per launch, each a serial k
// loop on one CU (17 us per 128 of depth) — four times as many tiles of a quarter the work each finish in a third of the time
// (a launch of 32 tile products of k = 256: 38 us with 128 x 128 tiles).  The k loop is gemm_glds64_body's with eight waves
// on the tile (2 x 4 waves of 32 x 16: two per SIMD cover each other's fragment reads), BKT 16, four LDS stages (74 KB).
template <int BKT, int NST, int NWV>
__global__ __launch_bounds__(64 * NWV, 4) void k_gemm_items64(const GemmItem* __restrict__ items, const int32_t* __restrict__ bin_start,
                                                               int64_t ld, const BatchTab* __restrict__ bt, int* __restrict__ counters, int nslots,
                                                               int64_t pstride)
{
    static_assert(NWV == 8, "2 x 4 waves of 32 x 16");
    constexpr int PAIR = 144, OPER = (BKT / 2) * PAIR, STAGE = 2 * OPER;
    constexpr int RA = 2, RB = 64 / (NWV / 2) / 4;
    constexpr int LPW = 2 * (BKT / 2) / NWV;
    __shared__ __attribute__((aligned(16))) double lds[NST * STAGE];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * (4 * RB);
    const int arow = wm + (lane & 15), bcol = wn + (lane & 3);
    const int kq = lane >> 4, kperm = ((kq & 1) << 1) | (kq >> 1); // 0,2,1,3
    const int r2 = 2 * (lane & 31), khalf = lane >> 5;
    const int i0 = bin_start[blockIdx.x], i1 = bin_start[blockIdx.x + 1];
    const int64_t step8 = (int64_t)(2 * NWV) * ld;
    for (int it = i0; it < i1; ++it) {
        GemmItem item = items[it];
        if (bt) {
            item.A = bt_rebase(bt, (int)blockIdx.z, item.A);
            item.B = bt_rebase(bt, (int)blockIdx.z, item.B);
            item.C = bt_rebase(bt, (int)blockIdx.z, item.C);
            if (item.T)
                item.T = bt_rebase(bt, (int)blockIdx.z, item.T);
            if (item.nch > 1) {
                item.D = bt_rebase(bt, (int)blockIdx.z, item.D);
                item.P1 = bt_rebase(bt, (int)blockIdx.z, item.P1);
            }
        }
        // this lane's 16-byte piece: rows (2 l', 2 l' + 1) of k-row kk + (lane >> 5), l' = lane & 31, clamped into the tile's valid
        // rows; wave w moves k-row pairs w, w + NWV, ..
        const int mr = (item.flags >> 8) & 255, nc = (item.flags >> 16) & 255;
        const int ma = (mr - 1) & ~1, mb = (nc - 1) & ~1;
        const double* pa = item.A + (r2 < ma ? r2 : ma) + (int64_t)(2 * wave + khalf) * ld;
        const double* pb = item.B + (r2 < mb ? r2 : mb) + (int64_t)(2 * wave + khalf) * ld;
        auto issue = [&](int stage) {
            double* sa = lds + stage * STAGE + wave * PAIR;
            double* sb = sa + OPER;
#pragma unroll
            for (int q = 0; q < BKT / 2 / NWV; ++q) {
                __builtin_amdgcn_global_load_lds(pa, (lds_void_t*)(sa + NWV * q * PAIR), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(pb, (lds_void_t*)(sb + NWV * q * PAIR), 16, 0, 0);
                pa += step8;
                pb += step8;
            }
        };
        double acc[RA][RB];
#pragma unroll
        for (int a = 0; a < RA; ++a)
#pragma unroll
            for (int b = 0; b < RB; ++b)
                acc[a][b] = 0.0;
        const int nk = item.k / BKT;
        for (int t = 0; t < NST - 1 && t < nk; ++t)
            issue(t);
        for (int t = 0; t < nk; ++t) {
            const int st = t % NST;
            if (t + NST - 1 < nk)
                issue((t + NST - 1) % NST);
            const int ahead = nk - 1 - t < NST - 1 ? nk - 1 - t : NST - 1; // stages still allowed in flight once stage t has landed
            if (ahead >= 3)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPW) : "memory");
            else if (ahead == 2)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory");
            else if (ahead == 1)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * LPW) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // every wave's pieces of stage st have landed
            const double* As = lds + st * STAGE;
            const double* Bs = As + OPER;
#pragma unroll
            for (int ks = 0; ks < BKT; ks += 4) {
                const int k = ks + kperm;
                const int off = (k >> 1) * PAIR + (k & 1) * 64;
                double af[RA], bf[RB];
#pragma unroll
                for (int x = 0; x < RA; ++x)
                    af[x] = As[off + arow + 16 * x];
#pragma unroll
                for (int x = 0; x < RB; ++x)
                    bf[x] = Bs[off + bcol + 4 * x];
#pragma unroll
                for (int n = 0; n < RB; ++n)
#pragma unroll
                    for (int m = 0; m < RA; ++m)
                        acc[m][n] = mfma4(af[m], bf[n], acc[m][n]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // stage st may be refilled
        }
        if (item.flags & 1) {
#pragma unroll
            for (int a = 0; a < RA; ++a)
#pragma unroll
                for (int b = 0; b < RB; ++b)
                    acc[a][b] = -acc[a][b];
        }
        {
            using WT = WaveTileC<RA, RB>;
            static_assert(NWV * WT::SCRATCH <= NST * STAGE, "transposition scratch must fit in the operand stages");
            const int rlim = mr - wm < WT::R ? mr - wm : WT::R, clim = nc - wn < WT::CN ? nc - wn : WT::CN;
            double* const Tw = item.T ? item.T + wn + (int64_t)wm * ld : nullptr;
            const bool cut = item.nch > 1; // (as in k_gemm_items)
            if (rlim > 0 && clim > 0)
                WT::template store_item<1>(acc, lds + wave * WT::SCRATCH, item.C + (int64_t)wn * ld + wm, cut ? nullptr : Tw, ld, rlim, clim, cut, lane);
            if (cut) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0)
                    s_last = (__hip_atomic_fetch_add(counters + item.slot + (int)blockIdx.z * nslots, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1)
                            % item.nch
                        == 0;
                __syncthreads();
                if (s_last && rlim > 0 && clim > 0)
                    WT::template fold_item<1>(lds + wave * WT::SCRATCH, item.D + (int64_t)wn * ld + wm, item.P1 + (int64_t)wn * ld + wm, pstride,
                                              item.nch - 1, Tw, ld, rlim, clim, lane);
            }
        }
        __syncthreads(); // the next item's prologue overwrites the LDS stages
    }
}

void launch_gemm_items(hipStream_t s, const GemmItem* items, const int32_t* bin_start, int nbins, int64_t ld, int tile, int* counters, int nslots,
                       int64_t pstride)
{
    if (nbins > 0 && tile == 64) {
        GPE_LAUNCH_NAMED("k_gemm_items64", (k_gemm_items64<16, 4, 8>), dim3((unsigned)nbins, 1, (unsigned)g_batch.G), dim3(512), 0, s, items,
                         bin_start, ld, g_batch.bt, counters, nslots, pstride);
        return;
    }

    if (nbins <= 0)
        return;
    GPE_LAUNCH_NAMED("k_gemm_items", (k_gemm_items<16, 2, 4, 4>), dim3((unsigned)nbins, 1, (unsigned)g_batch.G), dim3(512), 0, s, items, bin_start,
                     ld, g_batch.bt, counters, nslots, pstride);
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

static void launch_gemm_sub_impl(hipStream_t s, const GemmArgs& g, int use_glds64, int force);
void launch_gemm_sub(hipStream_t s, const GemmArgs& g0)
{
    if (g0.m <= 0 || g0.n <= 0)
        return;
    if (g0.k <= 0 && !g0.overwrite)
        return;
    const int use_glds64 = 1, force = 0; // (the tile is picked by problem size below; GemmArgs::tile forces one)
    if (g0.rhs_rows > 0) {
        // Right-hand-side rows as FMAs inside the kernel (gemm_glds64.h) instead of a row of tiles: the FMAs cost every
        // workgroup ~4 us before its first tile, the tile row costs n / 128 extra workgroups — which only matters when
        // they push the launch over a round of the chip (tools/updbench: the fifth update of N = 4096 has 253 + 22
        // tiles for 256 CUs, 77.5 -> 52.4 us; every other launch of that factorisation is 3-5 us slower with the FMAs).
        // So: FMAs exactly when dropping the row brings the tile count down to a whole round.
        GemmArgs q = g0;
        q.m = g0.m - g0.rhs_rows;
        q.rhs_rows = 0; // (for the tile counts below)
        static const int rhs_fma = getenv("GPE_RHS_FMA") ? atoi(getenv("GPE_RHS_FMA")) : 1; // 0: never, 2: always
        bool fma = false;
        if (rhs_fma && q.m > 0 && !q.overwrite && glds_ok(q) && g_batch.G == 1 && g0.grid_limit <= 0) {
            auto crosses = [&](int T, int64_t round) { return live_tiles(q, T, T) <= round && live_tiles(g0, T, T) > round; };
            fma = rhs_fma == 2 || crosses(128, 256) || (live_tiles(g0, 128, 128) < 200 && (crosses(64, 512) || crosses(64, 256)));
        }
        if (fma) {
            q.rhs_rows = g0.rhs_rows;
            launch_gemm_sub_impl(s, q, use_glds64, force);
        }
        else {
            q = g0;
            q.rhs_rows = 0;
            launch_gemm_sub_impl(s, q, use_glds64, force);
        }
        return;
    }
    launch_gemm_sub_impl(s, g0, use_glds64, force);
}

static void launch_gemm_sub_impl(hipStream_t s, const GemmArgs& g, int use_glds64, int force)
{
    int tile = g.tile ? g.tile : force;
    if (tile != 128 && tile != 64 && tile != 32) {
        // measured at k = 256 (tools/kbench): one 128 x 128 glds workgroup per CU runs at the
        // LDS-fed MFMA rate (~48 us per tile); below ~200 live tiles too many CUs idle and the
        // 64 x 64 tile (4x the workgroups) wins; the 32 x 64 tile serves the tiny panel steps
        const int64_t GB = g_batch.G; // a batched launch runs GB problems of this shape at once
        const int64_t t128_min = 200;
        if (glds_ok(g) && GB * live_tiles(g, 128, 128) >= t128_min)
            tile = 128;
        else if (glds_ok(g) && !g.ktri && use_glds64 && GB * live_tiles(g, 64, 64) >= 96) // (the 64 x 64 body has no ktri form)
            tile = 64; // deep-prefetch 64 x 64 glds kernel: also the latency-critical next-panel update
        else if (GB * live_tiles(g, 64, 64) >= 512)
            tile = 64;
        else
            tile = 32;
    }
    if (g.rhs_rows > 0) {
        // right-hand-side rows are FMAs inside the direct-to-LDS kernels only; any other kernel takes them as ordinary rows
        const bool direct = glds_ok(g) && (tile == 128 || (tile == 64 && !g.ktri && use_glds64));
        if (!direct) {
            GemmArgs q = g;
            q.m += q.rhs_rows;
            q.rhs_rows = 0;
            launch_gemm_sub_impl(s, q, use_glds64, force);
            return;
        }
    }
    // (Half-height 64 x 128 tiles for launches with fewer 128 x 128 tiles than CUs — two co-resident workgroups per CU — were
    // built and measured no faster: profiles/r04_half_height_ab.log.)
    if (tile == 128 && glds_ok(g))
        launch_glds128(s, g);
    else if (tile == 128)
        launch_tile<128, 128, 32, 2>(s, g);
    else if (tile == 64 && glds_ok(g) && !g.ktri && use_glds64)
        launch_glds64(s, g);
    else if (tile == 64)
        launch_tile<64, 64, 32, 2>(s, g);
    else if (g.k <= 64 && !g.ktri)
        launch_tile<32, 64, 64, 1>(s, g); // one-shot panel-step form
    else
        launch_tile<32, 64, 32, 2>(s, g);
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
