// gemm_glds64.h — pieces of the fp64 MFMA GEMM shared by gemm.hip and potrf.hip (the fused next-panel update):
// triangular tile enumeration, the lane = row C-tile traffic, and the body of the 64 x 64 direct-to-LDS kernel.
#pragma once
#include "dev.h"

#ifndef GPE_MFMA4_DEFINED
#define GPE_MFMA4_DEFINED
static __device__ __forceinline__ double mfma4(double a, double b, double c)
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
#endif
typedef __attribute__((address_space(3))) void lds_void_t;
#ifndef GTS64_
#define GTS64_(i) do { } while (0)
#endif

// first row tile of tile column tj that touches the lower triangle (global row >= global col),
// clamped to tiles_m
template <int TM, int TN>
static __host__ __device__ __forceinline__ int first_live_tile(const GemmArgs& g, int tj)
{
    const int tiles_m = (int)((g.m + TM - 1) / TM);
    const int64_t need = g.gcol0 + (int64_t)tj * TN - g.grow0 - (TM - 1); // ti*TM >= need
    int64_t t = need <= 0 ? 0 : (need + TM - 1) / TM;
    return (int)(t < tiles_m ? t : tiles_m);
}

// ---- right-hand-side rows ------------------------------------------------------------------------------------
// The rows appended under the matrix (engine.hip: z = L^-1 obs_mean rides in the factorisation) are not tiles: as a tile
// row they cost every trailing update n / 128 extra workgroups with ONE live row each (30 of 495 in the first update of
// N = 4096; 22 of 275 in the fifth, which they push over 256 CUs into a second round: 78 us instead of 50).  Instead
// every workgroup of the launch updates its n / G columns of those rows with plain FMAs before its first tile:
//   C[m + p, c] -= sum_kk A[m + p, kk] B[c, kk],   g.m = the main rows, G = workgroups of the launch.
// 16 columns at a time, thread = (column tid & 15, k-group tid >> 4); every thread has 4 loads of B and 16 of the rows in
// flight together, then the NT / 16 k-groups are added in order through LDS (red: 4 x NT / 16 x 16 doubles): bitwise
// reproducible.  Four rows per pass.
static __device__ __forceinline__ void gemm_rhs_rows_mfma(const GemmArgs& g, int R, int G, double* lds, int b);
template <int NT, bool MANY = false> // MANY: the kernel shapes with one workgroup per CU (the other ones must stay within 128 VGPRs)
static __device__ __forceinline__ void gemm_rhs_rows(const GemmArgs& g, int rhs_rows, int G, double* red)
{
    if constexpr (MANY && NT == 512) {
        if (rhs_rows > 4 && rhs_rows <= 80) { // many rows (a ragged order's last ones): as a matrix-core product, below
            gemm_rhs_rows_mfma(g, rhs_rows, G, red, (int)blockIdx.x);
            return;
        }
    }
    constexpr int KG = NT / 16;
    const int tid = threadIdx.x, cl = tid & 15, kg = tid >> 4, b = blockIdx.x;
    const int nc = (int)((g.n + G - 1) / G);
    const int64_t c_lo = (int64_t)b * nc;
    int64_t c_hi = c_lo + nc;
    c_hi = c_hi < g.n ? c_hi : g.n;
    const int kper = (int)((g.k + KG - 1) / KG);
    const int kk0 = kg * kper;
    int kk1 = kk0 + kper;
    kk1 = kk1 < (int)g.k ? kk1 : (int)g.k;
    for (int64_t cb = c_lo; cb < c_hi; cb += 16) {
        const int64_t c = cb + cl;
        const int64_t cc = c < c_hi ? c : c_hi - 1;
        for (int p0 = 0; p0 < rhs_rows; p0 += 4) {
            const int pn = rhs_rows - p0 < 4 ? rhs_rows - p0 : 4;
            // the finishing threads (tid < 16 pn: row tid >> 4, column tid & 15) fetch their C element now
            const int pf = kg < pn ? kg : pn - 1;
            double* Cp = g.C + (g.m + p0 + pf) + cc * g.ldc;
            const double cold = *Cp;
            double a4[4] = {0.0, 0.0, 0.0, 0.0};
            const double* Bc = g.B + cc;
            const double* Ar = g.A + g.m + p0;
            for (int kb = kk0; kb < kk1; kb += 4) {
                double bv[4], zv[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kc = kb + u < kk1 ? kb + u : kk1 - 1;
                    bv[u] = Bc[(int64_t)kc * g.ldb];
#pragma unroll
                    for (int p = 0; p < 4; ++p) // rows beyond pn repeat the last one: summed, never stored
                        zv[p][u] = Ar[(p < pn ? p : pn - 1) + (int64_t)kc * g.lda];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double keep = kb + u < kk1 ? 1.0 : 0.0;
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        a4[p] += zv[p][u] * (bv[u] * keep);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
                red[(p * KG + kg) * 16 + cl] = a4[p];
            __syncthreads();
            if (tid < 16 * pn) {
                double sacc = 0.0;
#pragma unroll
                for (int q = 0; q < KG; ++q)
                    sacc += red[(kg * KG + q) * 16 + cl];
                if (c < c_hi)
                    *Cp = cold - sacc;
            }
            __syncthreads();
        }
    }
}

// ... and the same rows when there are MANY of them (round 6): a ragged order's last rows ride under the matrix like right-hand sides
// (up to 63 + P rows), and as FMAs they cost ~3 us a row in front of every tile (N = 4130: the update 214 -> 320 us).  Here the
// workgroup's share — R rows x <= 16 columns at a time x k — is a small matrix-core product: the k range in chunks of 64 through
// LDS (A: R x 64, lane = row: contiguous; B: 16 x 64), wave w takes k-rows 8 w .. 8 w + 7 of a chunk for all (16-row, 4-column)
// blocks, the eight waves' sums are added in order through LDS at the end (bitwise reproducible).  512 threads; LDS: 8448 doubles.
static __device__ __forceinline__ void gemm_rhs_rows_mfma(const GemmArgs& g, int R, int G, double* lds, int b)
{
    constexpr int RS = 80, BS = 20, KC = 64; // strides of the staged chunks: As[kk][row], Bs[kk][col]
    double* const As = lds;
    double* const Bs = lds + KC * RS;
    double* const red = Bs + KC * BS; // [8 waves][4][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4;
    const int MB = (R + 15) / 16; // 16-row blocks (<= 5)
    const int nc = (int)((g.n + G - 1) / G);
    const int64_t c_lo = (int64_t)b * nc;
    int64_t c_hi = c_lo + nc;
    c_hi = c_hi < g.n ? c_hi : g.n;
    const double* Ab = g.A + g.m; // the rows under the main ones
    // this thread's elements of a chunk: ten of A (row ap, k-rows ak + 6.4 i: idx = tid + 512 i -> row idx % 80, k idx / 80), two of B
    for (int64_t cb = c_lo; cb < c_hi; cb += 16) {
        double acc[5][4];
#pragma unroll
        for (int mb = 0; mb < 5; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                acc[mb][nb] = 0.0;
        double pa[10], pbv[2];
        auto fetch = [&](int64_t kb) { // (unconditional loads from clamped addresses; masks where the values are used)
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int idx = tid + 512 * i, p = idx % RS, kk = idx / RS;
                int64_t kc = kb + kk;
                kc = kc < g.k ? kc : g.k - 1;
                pa[i] = Ab[(p < R ? p : R - 1) + kc * g.lda];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + 512 * i, c = idx & 15, kk = idx >> 4;
                int64_t kc = kb + kk;
                kc = kc < g.k ? kc : g.k - 1;
                const int64_t cc = cb + c < c_hi ? cb + c : c_hi - 1;
                pbv[i] = g.B[cc + kc * g.ldb];
            }
        };
        fetch(0);
        for (int64_t kb = 0; kb < g.k; kb += KC) {
            __syncthreads(); // the previous chunk is consumed
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int idx = tid + 512 * i, p = idx % RS, kk = idx / RS;
                As[kk * RS + p] = (p < R && kb + kk < g.k) ? pa[i] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + 512 * i, c = idx & 15, kk = idx >> 4;
                Bs[kk * BS + c] = (kb + kk < g.k) ? pbv[i] : 0.0;
            }
            fetch(kb + KC < g.k ? kb + KC : kb); // (the last one fetches its own chunk again: unconditional, counted)
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 8; ks += 4) {
                const int k = 8 * wv + ks + kq;
                double bf[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    bf[nb] = Bs[k * BS + 4 * nb + (lane & 3)];
#pragma unroll
                for (int mb = 0; mb < 5; ++mb) {
                    if (mb < MB) { // (wave-uniform)
                        const double af = As[k * RS + 16 * mb + (lane & 15)];
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb)
                            acc[mb][nb] = mfma4(af, bf[nb], acc[mb][nb]);
                    }
                }
            }
        }
        // the eight waves' sums, 16 rows at a time, in order
        const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4), dcol = lane & 3;
#pragma unroll
        for (int mb = 0; mb < 5; ++mb) {
            if (mb < MB) {
                __syncthreads();
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    red[(wv * 4 + nb) * 64 + lane] = acc[mb][nb];
                __syncthreads();
                if (tid < 256) {
                    const int nb = tid >> 6;
                    double sacc = 0.0;
#pragma unroll
                    for (int w = 0; w < 8; ++w)
                        sacc += red[(w * 4 + nb) * 64 + lane];
                    const int row = 16 * mb + drow;
                    const int64_t c = cb + 4 * nb + dcol;
                    if (row < R && c < c_hi) {
                        double* Cp = g.C + (g.m + row) + c * g.ldc;
                        *Cp = *Cp - sacc;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- C tile traffic in the lane = row layout ---------------------------------------------------------
// The accumulator of v_mfma_f64_4x4x4 puts lane l on row 4*((l>>2)&3) + (l>>4), column l&3 of a 16 x 4
// fragment: one global load/store of a fragment touches 4 columns x 16 rows, adjacent lanes sit in
// different columns (ldc * 8 bytes apart).  Measured inside the kernel (tools/kbench_g): such a load
// takes ~150 cycles to ISSUE and a store ~450, and the read-modify-write of a 128 x 128 C tile was 29 %
// of the tile time.  Here the wave turns its (16 RA) x (4 RB) accumulator through a private LDS scratch
// (no barrier: one wave's LDS operations execute in order) and all C traffic uses lane = row:
// 64 / R whole columns of R = 16 RA consecutive rows per instruction (512 contiguous bytes for RA = 4).
template <int RA, int RB>
struct WaveTileC {
    static constexpr int R = 16 * RA, CN = 4 * RB, SW = R + 2, CPI = 64 / R, NIT = CN / CPI;
    static constexpr int SCRATCH = CN * SW; // doubles of LDS per wave
    // cv[it] = C[row, col(it)] for this lane's (row, column) pairs; addresses clamped into the valid
    // rlim x clim part of the tile (no branches; lanes outside simply do not store later)
    static __device__ __forceinline__ void load(double (&cv)[NIT], const double* __restrict__ Cw, int64_t ldc, int rlim,
                                                int clim, int lane)
    {
        const int row = lane % R, cl = lane / R;
        const int rr = row < rlim ? row : rlim - 1;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = it * CPI + cl;
            cv[it] = Cw[(int64_t)(col < clim ? col : clim - 1) * ldc + rr];
        }
    }
    // C = cv - acc (mode 0), C = acc (mode 1: overwrite), C = cv + acc (mode 2: accumulate)
    static __device__ __forceinline__ void store(const double (&acc)[RA][RB], const double (&cv)[NIT], double* __restrict__ W,
                                                 double* __restrict__ Cw, int64_t ldc, int rlim, int clim, int mode,
                                                 int lane)
    {
        const bool overwrite = mode == 1;
        const double sgn = mode == 2 ? 1.0 : -1.0; // fma(-1, t, cv) == cv - t exactly
        const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4), dcol = lane & 3;
#pragma unroll
        for (int n = 0; n < RB; ++n)
#pragma unroll
            for (int m = 0; m < RA; ++m)
                W[(4 * n + dcol) * SW + 16 * m + drow] = acc[m][n];
        const int row = lane % R, cl = lane / R;
        double t[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            t[it] = W[(it * CPI + cl) * SW + row];
        if (rlim == R && clim == CN) {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                Cw[(int64_t)(it * CPI + cl) * ldc + row] = overwrite ? t[it] : fma(sgn, t[it], cv[it]);
        }
        else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int col = it * CPI + cl;
                if (row < rlim && col < clim)
                    Cw[(int64_t)col * ldc + row] = overwrite ? t[it] : fma(sgn, t[it], cv[it]);
            }
        }
    }
    // The epilogue of a tile product of a LIST (dev.h: GemmItem; k_gemm_items / k_gemm_items64), in NCH passes over column chunks:
    // C = acc, never read; `through`: with device-scope (write-through) stores — a chunk of a cut k range, which another
    // workgroup may add up; Tw (null: no): the tile transposed as well, Tw[c + r * ld] = C[r + c * ld], lanes along c.
    template <int NCH>
    static __device__ __forceinline__ void store_item(const double (&acc)[RA][RB], double* __restrict__ W, double* __restrict__ Cw,
                                                      double* __restrict__ Tw, int64_t ld, int rlim, int clim, bool through, int lane)
    {
        static_assert(RB % NCH == 0 && (4 * RB / NCH) % CPI == 0, "chunking");
        constexpr int RBC = RB / NCH, CNC = 4 * RBC, NITC = CNC / CPI;
        const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4), dcol = lane & 3;
        const int row = lane % R, cl = lane / R;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            double t[NITC];
#pragma unroll
            for (int n = 0; n < RBC; ++n)
#pragma unroll
                for (int m = 0; m < RA; ++m)
                    W[(4 * n + dcol) * SW + 16 * m + drow] = acc[m][ch * RBC + n];
#pragma unroll
            for (int it = 0; it < NITC; ++it)
                t[it] = W[(it * CPI + cl) * SW + row];
#pragma unroll
            for (int it = 0; it < NITC; ++it) {
                const int col = ch * CNC + it * CPI + cl;
                if (row < rlim && col < clim) {
                    if (through)
                        __hip_atomic_store(Cw + (int64_t)col * ld + row, t[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        Cw[(int64_t)col * ld + row] = t[it];
                }
            }
            if (Tw) { // (wave-uniform)
#pragma unroll
                for (int q = 0; q < R * CNC / 64; ++q) {
                    const int e = q * 64 + lane, c = e % CNC, r = e / CNC;
                    if (r < rlim && ch * CNC + c < clim)
                        Tw[(int64_t)r * ld + ch * CNC + c] = W[c * SW + r];
                }
            }
        }
    }
    // ... and what the workgroup that counted last does for a cut k range: D = D + P1 + .. + P(nparts), in that order, every
    // operand read device-wide (the chunks went out write-through from wherever they ran); D stored, and D^T at Tw
    template <int NCH>
    static __device__ __forceinline__ void fold_item(double* __restrict__ W, double* __restrict__ Dw, const double* __restrict__ P1w,
                                                     int64_t pstride, int nparts, double* __restrict__ Tw, int64_t ld, int rlim,
                                                     int clim, int lane)
    {
        constexpr int RBC = RB / NCH, CNC = 4 * RBC, NITC = CNC / CPI;
        const int row = lane % R, cl = lane / R;
        const int rr = row < rlim ? row : rlim - 1;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            double v[NITC];
            int64_t off[NITC];
#pragma unroll
            for (int it = 0; it < NITC; ++it) {
                const int col = ch * CNC + it * CPI + cl;
                off[it] = (int64_t)(col < clim ? col : clim - 1) * ld + rr;
                v[it] = __hip_atomic_load(Dw + off[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // (every load of the pass in flight before the first sum: the device-wide loads are 2-3 us each way, and this workgroup
            // is the only one working on the tile)
            double u1[NITC], u2[NITC], u3[NITC];
            if (nparts > 0) {
#pragma unroll
                for (int it = 0; it < NITC; ++it)
                    u1[it] = __hip_atomic_load(P1w + off[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (nparts > 1) {
#pragma unroll
                for (int it = 0; it < NITC; ++it)
                    u2[it] = __hip_atomic_load(P1w + pstride + off[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (nparts > 2) {
#pragma unroll
                for (int it = 0; it < NITC; ++it)
                    u3[it] = __hip_atomic_load(P1w + 2 * pstride + off[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (nparts > 0) {
#pragma unroll
                for (int it = 0; it < NITC; ++it)
                    v[it] += u1[it];
            }
            if (nparts > 1) {
#pragma unroll
                for (int it = 0; it < NITC; ++it)
                    v[it] += u2[it];
            }
            if (nparts > 2) {
#pragma unroll
                for (int it = 0; it < NITC; ++it)
                    v[it] += u3[it];
            }
#pragma unroll
            for (int it = 0; it < NITC; ++it) {
                const int col = ch * CNC + it * CPI + cl;
                if (row < rlim && col < clim)
                    Dw[(int64_t)col * ld + row] = v[it];
            }
            if (Tw) {
#pragma unroll
                for (int it = 0; it < NITC; ++it)
                    W[(it * CPI + cl) * SW + row] = v[it];
#pragma unroll
                for (int q = 0; q < R * CNC / 64; ++q) {
                    const int e = q * 64 + lane, c = e % CNC, r = e / CNC;
                    if (r < rlim && ch * CNC + c < clim)
                        Tw[(int64_t)r * ld + ch * CNC + c] = W[c * SW + r];
                }
            }
        }
    }
    // the same in NCH passes over column chunks: 1/NCH of the registers and of the LDS scratch
    // (SCRATCH / NCH doubles per wave) — for kernels that keep two workgroups per CU
    template <int NCH>
    static __device__ __forceinline__ void rmw_chunked(const double (&acc)[RA][RB], double* __restrict__ W,
                                                       double* __restrict__ Cw, int64_t ldc, int rlim, int clim,
                                                       int mode, int lane)
    {
        const bool overwrite = mode == 1;
        const double sgn = mode == 2 ? 1.0 : -1.0;
        static_assert(RB % NCH == 0 && (4 * RB / NCH) % CPI == 0, "chunking");
        constexpr int RBC = RB / NCH, CNC = 4 * RBC, NITC = CNC / CPI;
        const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4), dcol = lane & 3;
        const int row = lane % R, cl = lane / R;
        const int rr = row < rlim ? row : rlim - 1;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            double cv[NITC], t[NITC];
            if (!overwrite) {
#pragma unroll
                for (int it = 0; it < NITC; ++it) {
                    const int col = ch * CNC + it * CPI + cl;
                    cv[it] = Cw[(int64_t)(col < clim ? col : clim - 1) * ldc + rr];
                }
            }
#pragma unroll
            for (int n = 0; n < RBC; ++n)
#pragma unroll
                for (int m = 0; m < RA; ++m)
                    W[(4 * n + dcol) * SW + 16 * m + drow] = acc[m][ch * RBC + n];
#pragma unroll
            for (int it = 0; it < NITC; ++it)
                t[it] = W[(it * CPI + cl) * SW + row];
#pragma unroll
            for (int it = 0; it < NITC; ++it) {
                const int col = ch * CNC + it * CPI + cl;
                if (row < rlim && col < clim)
                    Cw[(int64_t)col * ldc + row] = overwrite ? t[it] : fma(sgn, t[it], cv[it]);
            }
        }
    }
};


// 64 x 64 tile variant for the mid-size and latency-critical updates (a few hundred tiles, k = 256):
// 4 waves (32 x 32 each), BKT = 16, FOUR LDS stages (74 KB, two workgroups per CU) so that three
// k-tiles are always in flight — the register-staged 64 x 64 kernel it replaces exposes one
// global-load latency per k-tile.  One glds instruction moves TWO k-rows of a 64-row operand (lanes
// 0-31 / 32-63), so k-rows sit in LDS as pairs: row kk at (kk >> 1) * 144 + (kk & 1) * 64.  The two
// k values a 32-lane half reads per MFMA step are taken from different pairs (k permutation
// 0,2,1,3 — applied to both operands, the sum over k does not care), which keeps the reads
// conflict-free.
// skip00: leave tile (0, 0) alone (the fused next-panel update: that tile is the next diagonal block, which the
// extra workgroup of the same launch updates and factors — potrf.hip:k_upd_fused)
// hooks of the body (potrf.hip: the next-panel update that also does the panel's last step): `pre` runs once the tile is
// known, before the first operand stage is requested; `post` gets the accumulators after the k loop, before the epilogue
struct Glds64NoHook {
    __device__ __forceinline__ void pre(int, int, int64_t, int64_t, int, int) const {}
    template <int RA_, int RB_>
    __device__ __forceinline__ void post(double (&)[RA_][RB_], int, int) const {}
};
template <int BKT, int NST, int NWV, class HOOK = Glds64NoHook>
static __device__ __forceinline__ void gemm_glds64_body(const GemmArgs& g, double* __restrict__ lds, int first_wg, int n_wg,
                                                        bool skip00, const HOOK& hook = HOOK())
{
    constexpr int TM = 64, TN = 64;
    static_assert(NWV == 4 || NWV == 8, "2 x 2 waves of 32 x 32 or 2 x 4 waves of 32 x 16");
    constexpr int PAIR = 144;                 // doubles per k-row pair (2 x 64 + 16 pad)
    constexpr int OPER = (BKT / 2) * PAIR;    // one operand, one stage
    constexpr int STAGE = 2 * OPER;
    constexpr int RA = 2, RB = 64 / (NWV / 2) / 4; // wave tile 32 x (64 / (NWV / 2))
    constexpr int LPW = 2 * (BKT / 2) / NWV;  // glds instructions per wave per stage

    const int tiles_m = (int)((g.m + TM - 1) / TM);
    const int tiles_n = (int)((g.n + TN - 1) / TN);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (g.rhs_rows > 0 && !skip00) // (the fused next-panel update keeps its right-hand-side rows as a tile row)
        gemm_rhs_rows<64 * NWV, NWV == 8>(g, g.rhs_rows, n_wg, lds);
    for (int lwg = first_wg; lwg < g.total; lwg += n_wg) {
        int wg = lwg;
        {
            const int nwg = g.total;
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
                    continue;
                ti = t1 + rr;
            }
        }
        else {
            ti = wg % tiles_m;
            tj = wg / tiles_m;
        }
        if (skip00 && ti == 0 && tj == 0)
            continue;
        const int64_t row0 = (int64_t)ti * TM, col0 = (int64_t)tj * TN;
        const int64_t mrows = g.m - row0, ncols = g.n - col0;
        const int mr = (int)(mrows < TM ? mrows : TM), nc = (int)(ncols < TN ? ncols : TN);
        const int wm = (wave & 1) * 32, wn = (wave >> 1) * (4 * RB);
        const int arow = wm + (lane & 15), bcol = wn + (lane & 3);
        const int kq = lane >> 4, kperm = ((kq & 1) << 1) | (kq >> 1); // 0,2,1,3

        // this lane's 16-byte piece: rows (2 l', 2 l' + 1) of k-row kk + (lane >> 5), l' = lane & 31
        int ra = 2 * (lane & 31), rb = ra;
        {
            const int ma = (mr - 1) & ~1, mb = (nc - 1) & ~1;
            ra = ra < ma ? ra : ma;
            rb = rb < mb ? rb : mb;
        }
        const int khalf = lane >> 5;
        hook.pre(ti, tj, row0, col0, mr, nc);
        const double* pa = g.A + row0 + ra + (int64_t)(2 * wave + khalf) * g.lda;
        const double* pb = g.B + col0 + rb + (int64_t)(2 * wave + khalf) * g.ldb;
        // wave w moves pairs w, w + NWV, .. of each operand
        const int64_t astep8 = (int64_t)(2 * NWV) * g.lda, bstep8 = (int64_t)(2 * NWV) * g.ldb;
        auto issue = [&](int stage) {
            double* sa = lds + stage * STAGE + wave * PAIR;
            double* sb = sa + OPER;
#pragma unroll
            for (int q = 0; q < BKT / 2 / NWV; ++q) { // BKT / 2 pairs per operand and stage
                __builtin_amdgcn_global_load_lds(pa, (lds_void_t*)(sa + NWV * q * PAIR), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(pb, (lds_void_t*)(sb + NWV * q * PAIR), 16, 0, 0);
                pa += astep8;
                pb += bstep8;
            }
        };

        double acc[RA][RB];
#pragma unroll
        for (int a = 0; a < RA; ++a)
#pragma unroll
            for (int b = 0; b < RB; ++b)
                acc[a][b] = 0.0;

        const int nk = (int)(g.k / BKT);
        GTS64_(0);
        for (int t = 0; t < NST - 1 && t < nk; ++t)
            issue(t);
        for (int t = 0; t < nk; ++t) {
            const int st = t % NST;
            if (t < 16)
                GTS64_(1 + t);
            if (t + NST - 1 < nk)
                issue((t + NST - 1) % NST);
            // tiles still allowed in flight after tile t has landed
            const int ahead = nk - 1 - t < NST - 1 ? nk - 1 - t : NST - 1;
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
        GTS64_(17);
        hook.post(acc, wm, wn);

        {
            using WT = WaveTileC<RA, RB>;
            static_assert(NWV * WT::SCRATCH <= NST * STAGE, "transposition scratch must fit in the operand stages");
            double* Cw = g.C + (col0 + wn) * g.ldc + row0 + wm;
            const int rlim = mr - wm < WT::R ? mr - wm : WT::R, clim = nc - wn < WT::CN ? nc - wn : WT::CN;
            if (rlim > 0 && clim > 0) {
                double cv[WT::NIT];
                if (g.overwrite != 1)
                    WT::load(cv, Cw, g.ldc, rlim, clim, lane);
                WT::store(acc, cv, lds + wave * WT::SCRATCH, Cw, g.ldc, rlim, clim, g.overwrite, lane);
            }
        }
        GTS64_(18);
        __syncthreads();
    }
}


template <int BKT>
struct Glds64Shape {
    static constexpr int PAIR = 144, OPER = (BKT / 2) * PAIR, STAGE = 2 * OPER;
};
