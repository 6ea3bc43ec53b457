// potrf.hip — the serial piece of the blocked Cholesky (gfx950): the 64 x 64 diagonal block.
//
// Together with gemm.hip this replaces Eigen::LLT<MatrixXd>(K).matrixL()
// (src/limbo/model/gp.hpp:565) and the TriangularView solves (gp.hpp:260-261, :608-610, :620).
//
//   k_diag      one workgroup: factor the diagonal block D = L11 L11^T and invert its two 32 x 32 diagonal
//               half-blocks; writes L11 in place and X^T = L11^-T (as Xt[k + 64 c] = X[c][k]) to a side
//               buffer.  Since round 2 this is a data-flow of seven specialised waves (diag_flow.h, the
//               default, also inside k_panel_step / k_upd_fused); the barrier rounds described below are
//               what k_diag_full (ragged blocks, add_sample, load) still runs.
//   k_diag_inv  the inversion alone, batched over blocks (load(..., recompute = false) and
//               add_sample need the inverses of blocks they did not factor).
//
// Everything below the diagonal block then is a matrix-core product with X (gemm.hip):
// L21 = A21 X^T, and the triangular sweeps become 64 x 64 mat-vecs with X / X^T (solve.hip).
//
// Factorisation.  The critical path of a Cholesky is the chain of N pivots
// (pivot -> rsqrt -> scale -> update of the next column), ~170 cycles each on this chip
// (tools/ubench.hip); everything here is arranged around it.  Thread (r, w): lane r = row,
// wave w owns the four-column groups g = 4q + w (columns 4g .. 4g+3), 16 doubles per thread.
// Round g: the owner wave first applies round g-1's rank-4 update to its four columns, then
// factors the 64 x 4 panel entirely inside the wave (pivot and multipliers broadcast with
// v_readlane, 1/sqrt by v_rsq_f64 + two Newton steps — no IEEE sqrt/div sequence on the chain),
// and publishes the four scaled columns through LDS; ONE barrier per four columns.  The other
// three waves meanwhile apply the previous rank-4 update to all their later columns; the owner's
// own non-critical columns are caught up one round later (4 LDS buffers keep that legal).
#include "dev.h"
#include <hip/hip_ext.h>
#include <atomic>
#include <cstdio>

#define NB 64
#ifndef DIAG_ABL
#define DIAG_ABL 0 // debug ablations (tools/kbench): 1 = no factorisation rounds, 2 = no inversion
#endif
#ifdef DIAG_TIMING
__device__ long long g_diag_arr[16][8]; // per round: arrival of waves 0-3 and of the inversion wave (4) at the closing barrier
__device__ long long g_diag_ts[32];
#ifdef DIAG_NO_STAMPS // the arrays exist for tools/diagbench.hip, the kernel is the shipped one
#define ARR(G, w) do { } while (0)
#define TS(i) do { } while (0)
#else
#define ARR(G, w) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_diag_arr[G][w] = clock64(); } while (0)
#define TS(i) do { if (threadIdx.x == 0) g_diag_ts[i] = clock64(); } while (0)
#endif
#else
#define TS(i) do { } while (0)
#define ARR(G, w) do { } while (0)
#endif
#ifdef DIAG_TIMING
__device__ long long g_panel_ts[64];
#define PTS(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) g_panel_ts[(blockIdx.x == 0 ? 0 : 32) + (i)] = clock64(); } while (0)
__device__ long long g_tail_ts[64][4]; // k_tail, the diagonal workgroup of column c: updates done | solved | factoring | factored
__device__ long long g_tail_cyc[64][4]; // ... the same stamps in shader-clock cycles (clock64): cycles / wall time = the clock the CU ran at
#define TTS(c, i) do { if (threadIdx.x == 0 && (c) < 64) { g_tail_ts[(c)][(i)] = wall_clock64(); g_tail_cyc[(c)][(i)] = clock64(); } } while (0)
__device__ long long g_tail_ts2[64][12]; // ... inside its two-phase solve: X11/L21 seen | phase A done | X22 seen | X22 in LDS | Y2 written | done
#define TTS2(x, on, i) do { if ((on) && threadIdx.x == 0 && ((x).R0 - (x).p0) / NB < 64) g_tail_ts2[((x).R0 - (x).p0) / NB][(i)] = wall_clock64(); } while (0)
__device__ long long g_p256_ts[5][32]; // k_panel256: strips 0..3 and the last one; [6 S + i] = stamp i of step S, [30] start, [31] end
#define P2TS(i) do { if (threadIdx.x == 0 && (blockIdx.x < 4 || blockIdx.x == gridDim.x - 1)) g_p256_ts[blockIdx.x < 4 ? blockIdx.x : 4][(i)] = wall_clock64(); } while (0)
#else
#define PTS(i) do { } while (0)
#define P2TS(i) do { } while (0)
#define TTS(c, i) do { } while (0)
#define TTS2(x, on, i) do { } while (0)
#endif
#define XS 66 // LDS row stride (doubles) of the 64 x 64 work matrices: conflict-free MFMA operand reads

#include "gemm_glds64.h" // mfma4, and the 64 x 64 GEMM body for the fused next-panel update

static __device__ __forceinline__ double bcast_lane(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// Scaling a column by 1/sqrt(p) with the shortest dependent chain (every fp64 op costs ~32 cycles
// of latency on the pivot chain): y0 = v_rsq_f64(p) is a ~1e-8-accurate seed; with
// eh = 1/2 - (p/2) y0^2 (= half the relative residual) the Newton-corrected factor is
// y = y0 (1 + eh) and a * y = fma(a*y0, eh, a*y0).  Chain: rsq -> mul -> fma -> fma.
// A second correction step is applied off the chain only to the stored inverse pivot `y`.
struct RsqScale {
    double y0, eh;
    __device__ __forceinline__ explicit RsqScale(double p)
    {
        y0 = __builtin_amdgcn_rsq(p);
        const double t = (0.5 * p) * y0;
        eh = fma(-t, y0, 0.5);
    }
    __device__ __forceinline__ double scale(double a) const
    {
        const double l = a * y0;
        return fma(l, eh, l);
    }
    // 1/sqrt(p) itself, one more Newton step (not on the critical path)
    __device__ __forceinline__ double inv(double p) const
    {
        double y = fma(y0, eh, y0);
        const double t = p * y;
        const double e = fma(-t, y, 1.0);
        return fma(0.5 * y, e, y);
    }
};

// a[qq][*] -= sum_e Lt[r][e] * Lt[c][e] for this thread's column groups qq with 4 qq + w >= gmin
// (Lt: one round's four scaled columns, Lt[row * 4 + e])
static __device__ __forceinline__ void rank4_update(double (&a)[4][4], const double* __restrict__ Lt, int r, int w,
                                                    int gmin, int qlo)
{
    const double m0 = Lt[r * 4 + 0], m1 = Lt[r * 4 + 1], m2 = Lt[r * 4 + 2], m3 = Lt[r * 4 + 3];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        if (qq < qlo || 4 * qq + w < gmin)
            continue; // wave-uniform
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double* lc = Lt + (16 * qq + 4 * w + e) * 4; // wave-uniform address: LDS broadcast
            double v = a[qq][e];
            v = fma(-m0, lc[0], v);
            v = fma(-m1, lc[1], v);
            v = fma(-m2, lc[2], v);
            v = fma(-m3, lc[3], v);
            a[qq][e] = v;
        }
    }
}

template <int G>
struct DiagRound {
    static __device__ __forceinline__ void run(double (&a)[4][4], double* __restrict__ Ltb, double* __restrict__ invd,
                                               int* __restrict__ sbad, int r, int w, double* __restrict__ Ls)
    {
        DiagRound<G - 1>::run(a, Ltb, invd, sbad, r, w, Ls);
        constexpr int q = G >> 2, own = G & 3, c0 = 4 * G;
        double* Lt = Ltb + (G & 3) * (NB * 4);
        const double* Lp = Ltb + ((G + 3) & 3) * (NB * 4); // round G-1
        const double* Lpp = Ltb + ((G + 2) & 3) * (NB * 4); // round G-2
        if (w == own) {
            if (G > 0) {
                // critical: round G-1's update on this group's four columns only
                const double m0 = Lp[r * 4 + 0], m1 = Lp[r * 4 + 1], m2 = Lp[r * 4 + 2], m3 = Lp[r * 4 + 3];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double* lc = Lp + (c0 + e) * 4;
                    double v = a[q][e];
                    v = fma(-m0, lc[0], v);
                    v = fma(-m1, lc[1], v);
                    v = fma(-m2, lc[2], v);
                    v = fma(-m3, lc[3], v);
                    a[q][e] = v;
                }
            }
            double a0 = a[q][0], a1 = a[q][1], a2 = a[q][2], a3 = a[q][3];
            // column c0
            const double p0 = bcast_lane(a0, c0);
            const RsqScale s0(p0);
            const double l0 = s0.scale(a0);
            a1 = fma(-l0, bcast_lane(l0, c0 + 1), a1);
            // column c0+1
            const double p1 = bcast_lane(a1, c0 + 1);
            const RsqScale s1(p1);
            a2 = fma(-l0, bcast_lane(l0, c0 + 2), a2);
            a3 = fma(-l0, bcast_lane(l0, c0 + 3), a3);
            const double l1 = s1.scale(a1);
            a2 = fma(-l1, bcast_lane(l1, c0 + 2), a2);
            // column c0+2
            const double p2 = bcast_lane(a2, c0 + 2);
            const RsqScale s2(p2);
            a3 = fma(-l1, bcast_lane(l1, c0 + 3), a3);
            const double l2 = s2.scale(a2);
            a3 = fma(-l2, bcast_lane(l2, c0 + 3), a3);
            // column c0+3
            const double p3 = bcast_lane(a3, c0 + 3);
            const RsqScale s3(p3);
            const double l3 = s3.scale(a3);
            const double y0 = s0.inv(p0), y1 = s1.inv(p1), y2 = s2.inv(p2), y3 = s3.inv(p3);
            a[q][0] = l0;
            a[q][1] = l1;
            a[q][2] = l2;
            a[q][3] = l3;
            Lt[r * 4 + 0] = l0;
            Lt[r * 4 + 1] = l1;
            Lt[r * 4 + 2] = l2;
            Lt[r * 4 + 3] = l3;
            if (Ls) { // the finished columns, for the inversion pipeline (XPipe32)
                Ls[r * XS + c0 + 0] = l0;
                Ls[r * XS + c0 + 1] = l1;
                Ls[r * XS + c0 + 2] = l2;
                Ls[r * XS + c0 + 3] = l3;
            }
            if (r == 0) {
                invd[c0 + 0] = y0;
                invd[c0 + 1] = y1;
                invd[c0 + 2] = y2;
                invd[c0 + 3] = y3;
                // first non-positive pivot (the reference never checks LLT::info(), gp.hpp:565)
                int bad = 0;
                if (!(p3 > 0.0))
                    bad = c0 + 4;
                if (!(p2 > 0.0))
                    bad = c0 + 3;
                if (!(p1 > 0.0))
                    bad = c0 + 2;
                if (!(p0 > 0.0))
                    bad = c0 + 1;
                if (bad != 0 && *sbad == 0)
                    *sbad = bad;
            }
        }
        else {
            if (G > 1 && w == ((G - 1) & 3)) // last round's owner catches up on round G-2
                rank4_update(a, Lpp, r, w, G, 0);
            if (G > 0)
                rank4_update(a, Lp, r, w, G, 0);
        }
        ARR(G, w);
        __syncthreads();
        if (G < 16)
            TS(10 + G);
    }
};
template <>
struct DiagRound<-1> {
    static __device__ __forceinline__ void run(double (&)[4][4], double*, double*, int*, int, int, double*) {}
};

// (Round 2 also built 8-column rounds here — half the barriers and publishes, optionally with a 1/p update chain — and
// measured them neutral, profiles/r02_diag_rounds.log; the data-flow form in diag_flow.h replaced that line of attack and
// the code was removed.)

#define DIAG_COL(q, e, w) (16 * (q) + 4 * (w) + (e))
#define DIAG_LTB (4 * NB * 4)
#define DIAG_RUN(a, Ltb, invd, sbad, r, w, Ls) DiagRound<15>::run(a, Ltb, invd, sbad, r, w, Ls)
#include "diag_flow.h"
#include "kfun_fast.h"
#define DIAG_THREADS 512

// ---- inversion of the 64 x 64 lower-triangular L (in LDS, Ls[row * XS + col]) --------------------
// acc[n] += sum_{k < 16} P[i0 + i][pk0 + k] * Q[qk0 + k][j0 + 4 n + j]   (16 x 16 x 16, n in [n0, n1))
static __device__ __forceinline__ void mm16(const double* __restrict__ P, int i0, int pk0, const double* __restrict__ Q,
                                            int qk0, int j0, double (&acc)[4], int n0, int n1, int lane)
{
    const int ai = i0 + (lane & 15), kq = lane >> 4, bj = j0 + (lane & 3);
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
        const double av = P[ai * XS + pk0 + ks + kq];
#pragma unroll
        for (int n = 0; n < 4; ++n)
            if (n >= n0 && n < n1)
                acc[n] = mfma4(av, Q[(qk0 + ks + kq) * XS + bj + 4 * n], acc[n]);
    }
}
// D[i0 + row][j0 + col] = sign * acc   (result layout of v_mfma_f64_4x4x4_4b, see gemm.hip)
static __device__ __forceinline__ void st16(double* __restrict__ D, int i0, int j0, const double (&acc)[4], int n0,
                                            int n1, double sign, int lane)
{
    const int row = i0 + 4 * ((lane >> 2) & 3) + (lane >> 4), col = j0 + (lane & 3);
#pragma unroll
    for (int n = 0; n < 4; ++n)
        if (n >= n0 && n < n1)
            D[row * XS + col + 4 * n] = sign * acc[n];
}

static __device__ __forceinline__ void invert_level2(const double* __restrict__ Ls, double* __restrict__ Xs,
                                                     double* __restrict__ Ts);
// Ls: L (lower, zeros above).  invd[j] = 1 / L[j][j].  Xs <- L^-1 (zeros above).  Ts: scratch.
// All 256 threads; ends with a barrier.
static __device__ __forceinline__ void invert_L64(const double* __restrict__ Ls, const double* __restrict__ invd,
                                                  double* __restrict__ Xs, double* __restrict__ Ts)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < NB * XS; e += 256)
        Xs[e] = 0.0;
    __syncthreads();
    TS(4);
    { // level 0: the four 16 x 16 diagonal blocks, wave w -> block w, lane (mod 16) = column of X
        const int b0 = 16 * w, c = lane & 15;
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            x[j] = (j == c) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x[j] *= invd[b0 + j];
#pragma unroll
            for (int i = j + 1; i < 16; ++i)
                x[i] = fma(-Ls[(b0 + i) * XS + b0 + j], x[j], x[i]);
        }
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                Xs[(b0 + j) * XS + b0 + c] = x[j];
        }
    }
    __syncthreads();
    TS(5);
    { // level 1: blocks (1,0) and (3,2):  X_ib,jb = -X_ib,ib (L_ib,jb X_jb,jb); two waves per block
        const int t = w >> 1, h = w & 1, ib = 2 * t + 1, jb = 2 * t;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        mm16(Ls, 16 * ib, 16 * jb, Xs, 16 * jb, 16 * jb, acc, 2 * h, 2 * h + 2, lane);
        st16(Ts, 16 * ib, 16 * jb, acc, 2 * h, 2 * h + 2, 1.0, lane);
        __syncthreads();
        double acc2[4] = {0.0, 0.0, 0.0, 0.0};
        mm16(Xs, 16 * ib, 16 * ib, Ts, 16 * ib, 16 * jb, acc2, 2 * h, 2 * h + 2, lane);
        st16(Xs, 16 * ib, 16 * jb, acc2, 2 * h, 2 * h + 2, -1.0, lane);
    }
    __syncthreads();
    TS(6);
    invert_level2(Ls, Xs, Ts);
    TS(7);
}

// level 2: X[32:64, 0:32] = -X22 (L21 X11) given the two 32 x 32 diagonal inverses in Xs, one
// 16 x 16 block per wave; 256 threads, ends with a barrier
static __device__ __forceinline__ void invert_level2(const double* __restrict__ Ls, double* __restrict__ Xs,
                                                     double* __restrict__ Ts)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ib = 2 + (w >> 1), jb = w & 1;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int kb = jb; kb < 2; ++kb) // T = L_21 X_11, X_11 lower
        mm16(Ls, 16 * ib, 16 * kb, Xs, 16 * kb, 16 * jb, acc, 0, 4, lane);
    st16(Ts, 16 * ib, 16 * jb, acc, 0, 4, 1.0, lane);
    __syncthreads();
    double acc2[4] = {0.0, 0.0, 0.0, 0.0};
    for (int kb = 2; kb <= ib; ++kb) // X_21 = -X_22 T, X_22 lower
        mm16(Xs, 16 * ib, 16 * kb, Ts, 16 * kb, 16 * jb, acc2, 0, 4, lane);
    st16(Xs, 16 * ib, 16 * jb, acc2, 0, 4, -1.0, lane);
    __syncthreads();
}

// Xt[k + 64 c] = X[c][k]
static __device__ __forceinline__ void store_Xt(const double* __restrict__ Xs, double* __restrict__ Xt)
{
    for (int e = threadIdx.x; e < NB * NB; e += 256)
        Xt[e] = Xs[(e >> 6) * XS + (e & 63)];
}

// The block the fused panel steps start from: L11 in place and X^T = L11^-T into Xt (the quarter above the
// diagonal stays zero), by the data-flow form of diag_flow.h.  Full 64 x 64 blocks only.
static __device__ __forceinline__ void diag_body(double* __restrict__ A, int64_t lda, double* __restrict__ Xt,
                                                 int* __restrict__ info, int64_t goff)
{
    __shared__ __attribute__((aligned(16))) double Ls[NB * XS];
    __shared__ __attribute__((aligned(16))) double Ltb[DIAG_LTB];
    __shared__ __attribute__((aligned(16))) double invd[NB];
    const int r = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ DiagSync sy;
    __shared__ __attribute__((aligned(16))) double Xw[DIAG_XW_DOUBLES];
    static_assert(DIAG_H_DOUBLES <= DIAG_LTB, "H fits where the round buffers were");
    TS(0);
    for (int e = threadIdx.x; e < NB * NB; e += DIAG_THREADS)
        Ls[(e & 63) * XS + (e >> 6)] = A[(e & 63) + (int64_t)(e >> 6) * lda];
    diag_flow_init(&sy);
    __syncthreads();
    TS(1);
    diag_flow(Ls, Ltb, invd, &sy, A, lda, Xt, info, goff, w, r, Xw);
    TS(2);
    return;
}
// entry points: single GP (the round-1 kernel, unchanged) / batched (gridDim.z GPs, pointers rebased; dev.h)
__global__ __launch_bounds__(DIAG_THREADS) void k_diag(double* __restrict__ A, int64_t lda, double* __restrict__ Xt,
                                              int* __restrict__ info, int64_t goff)
{
    diag_body(A, lda, Xt, info, goff);
}
__global__ __launch_bounds__(DIAG_THREADS) void k_diag_b(double* __restrict__ A, int64_t lda, double* __restrict__ Xt,
                                                int* __restrict__ info, int64_t goff, const BatchTab* __restrict__ bt)
{
    BT_REBASE(bt, A);
    BT_REBASE(bt, Xt);
    BT_REBASE(bt, info);
    diag_body(A, lda, Xt, info, goff);
}

// Full-inverse form (any jb <= 64): the three-launch panel step and its GEMM consumers need all of X.
__global__ __launch_bounds__(256) void k_diag_full(double* __restrict__ A, int64_t lda, int jb, double* __restrict__ Xt,
                                              int* __restrict__ info, int64_t goff, const BatchTab* __restrict__ bt)
{
    BT_REBASE(bt, A);
    BT_REBASE(bt, Xt);
    BT_REBASE(bt, info);
    __shared__ __attribute__((aligned(16))) double Ls[NB * XS];
    __shared__ __attribute__((aligned(16))) double Xs[NB * XS];
    __shared__ __attribute__((aligned(16))) double Ts[NB * XS];
    __shared__ __attribute__((aligned(16))) double Ltb[DIAG_LTB];
    __shared__ double invd[NB];
    __shared__ int sbad;
    const int r = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0)
        sbad = 0;
    double a[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = DIAG_COL(q, e, w);
            // only the lower triangle of A is meaningful; a short block is padded with the identity
            a[q][e] = (r < jb && c < jb) ? ((c <= r) ? A[r + (int64_t)c * lda] : 0.0) : ((r == c) ? 1.0 : 0.0);
        }
    TS(0);
    __syncthreads();
    TS(1);
#if DIAG_ABL != 1
    DIAG_RUN(a, Ltb, invd, &sbad, r, w, nullptr);
#endif
    TS(2);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = DIAG_COL(q, e, w);
            const double v = (c <= r) ? a[q][e] : 0.0;
            Ls[r * XS + c] = v;
            if (r < jb && c <= r)
                A[r + (int64_t)c * lda] = v;
        }
    if (threadIdx.x == 0 && sbad != 0 && sbad <= jb && *info == 0)
        *info = (int)(goff + sbad);
    __syncthreads();
    TS(3);
#if DIAG_ABL != 2
    invert_L64(Ls, invd, Xs, Ts);
#endif
    store_Xt(Xs, Xt);
    TS(8);
}

#ifdef DIAG_TIMING
void dump_diag_timing()
{
    {
        long long a[16][8];
        (void)hipMemcpyFromSymbol(a, HIP_SYMBOL(g_diag_arr), sizeof(a));
        printf("arrival at the closing barrier of round G, cycles after the previous barrier's last arrival (waves 0-3 factor, X = inversion wave; * = owner):\n");
        long long prev = 0;
        for (int G = 0; G < 16; ++G) {
            long long last = 0;
            for (int w = 0; w < 5; ++w)
                last = a[G][w] > last ? a[G][w] : last;
            if (G > 0) {
                printf("  G=%2d:", G);
                for (int w = 0; w < 5; ++w)
                    printf(" %s%6lld%s", w == 4 ? "X" : "w", a[G][w] - prev, w == (G & 3) ? "*" : " ");
                printf("\n");
            }
            prev = last;
        }
    }
    long long h[32];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_diag_ts), sizeof(h));
    printf("k_diag cycles (factor waves): load %lld | rounds %lld | writeL %lld | total %lld\n", h[1] - h[0], h[2] - h[1],
           h[3] - h[2], h[3] - h[0]);
    printf("rounds:");
    for (int g = 0; g < 16; ++g)
        printf(" %lld", h[10 + g] - (g ? h[9 + g] : h[1]));
    printf("\n");
}
#endif
void launch_diag(hipStream_t s, double* A, int64_t lda, int jb, double* Xt, int* info, int64_t goff, int half_form)
{
    if (half_form && jb == NB)
    {
        if (g_batch.bt)
            GPE_LAUNCH(k_diag_b, dim3(1, 1, g_batch.G), dim3(DIAG_THREADS), 0, s, A, lda, Xt, info, goff, g_batch.bt);
        else
            GPE_LAUNCH(k_diag, dim3(1), dim3(DIAG_THREADS), 0, s, A, lda, Xt, info, goff);
    }
    else
        GPE_LAUNCH(k_diag_full, dim3(1, 1, g_batch.G), dim3(256), 0, s, A, lda, jb, Xt, info, goff, g_batch.bt);
}

// ---------------------------------------------------------------------------------------------
// k_panel_step — one 64-column step of the blocked factorisation below an already factored
// diagonal block, fused into ONE launch (a dependent launch costs ~3.4 us here, so the
// three-launch form [L21 = A21 X^T | in-panel update | next k_diag] pays 10 us of floor per step):
//   workgroup b owns rows R_b = [r0 + 64 b, +64) of the panel (r0 = j0 + 64):
//     1. L_b = A[R_b, j0:j0+64] X^T                                    (matrix cores, X = L11^-1)
//     2. for every remaining 64-column block t of the outer panel with t <= b:
//          A[R_b, block t] -= L_b L_t^T,  L_t = rows of block t of the same product.
//        L_t belongs to another workgroup; instead of an inter-workgroup hand-off it is recomputed
//        here (256 MFMAs per wave, hidden behind workgroup 0's serial diagonal factorisation).
//     3. workgroup 0 then holds the fully updated next diagonal block and factors + inverts it
//        (same code as k_diag), so the next step needs no separate diagonal launch.
// Requires full 64-column blocks (host falls back to the three-launch form otherwise).
// ---------------------------------------------------------------------------------------------
#define PS 80 // stride (doubles) of the [kk][i] operand tiles: == 16 mod 32

// acc[m][n] += sum_{k in [k0, k0+KLEN)} Aop[k][wm + 16 m + ..] * B(col, k)  — 8 waves.
//   wave tile: 32 rows x (4 RBN) columns starting at column wn
//   BKM = true : B stored k-contiguous, B(col, k) = Bop[col * XS + k]     (X or L blocks, [c][k])
//   BKM = false: B stored [kk][n],      B(col, k) = Bop[k * PS + col]
template <bool BKM, int KLEN, int RBN, int BS = XS> // BS: row stride of a k-contiguous B
static __device__ __forceinline__ void mmk(const double* __restrict__ Aop, int ak0, const double* __restrict__ Bop,
                                           int bk0, int wm, int wn, int lane, double (&acc)[2][RBN])
{
    const int ar = wm + (lane & 15), bc = wn + (lane & 3), kq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KLEN; ks += 4) {
        double af[2], bf[RBN];
#pragma unroll
        for (int x = 0; x < 2; ++x)
            af[x] = Aop[(ak0 + ks + kq) * PS + ar + 16 * x];
#pragma unroll
        for (int x = 0; x < RBN; ++x)
            bf[x] = BKM ? Bop[(bc + 4 * x) * BS + bk0 + ks + kq] : Bop[(bk0 + ks + kq) * PS + bc + 4 * x];
#pragma unroll
        for (int n = 0; n < RBN; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                acc[m][n] = mfma4(af[m], bf[n], acc[m][n]);
    }
}
template <bool BKM>
static __device__ __forceinline__ void mm64(const double* __restrict__ Aop, const double* __restrict__ Bop, int wm, int wn,
                                            int lane, double (&acc)[2][4])
{
    mmk<BKM, NB, 4>(Aop, 0, Bop, 0, wm, wn, lane, acc);
}

// The same solve with ALL of X = L11^-1 (lower triangular; diag_flow.h leaves the off-diagonal quarter too): ONE product,
// Y[i][c] = sum_{k <= c} T[i][k] X[c][k], two barriers instead of six.  A wave's 16 columns need k < wn + 16 only.
static __device__ __forceinline__ void trsm_tile_full(double* __restrict__ T, const double* __restrict__ Bx, int lane, int wave)
{
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 16;
    const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4), dcol = lane & 3;
    double y[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            y[m][n] = 0.0;
    switch (wave >> 1) {
    case 0: mmk<true, 16, 4>(T, 0, Bx, 0, wm, wn, lane, y); break;
    case 1: mmk<true, 32, 4>(T, 0, Bx, 0, wm, wn, lane, y); break;
    case 2: mmk<true, 48, 4>(T, 0, Bx, 0, wm, wn, lane, y); break;
    default: mmk<true, 64, 4>(T, 0, Bx, 0, wm, wn, lane, y); break;
    }
    __syncthreads(); // every wave has read what it needs of T
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            T[(wn + 4 * n + dcol) * PS + wm + 16 * m + drow] = y[m][n];
    __syncthreads();
}

// 64 x 64 tile of column-major G (rows clamped to nrows) <-> registers <-> T[kk * PS + i]; 512 threads
struct TileRegs {
    double v[8];
    __device__ __forceinline__ void load(const double* __restrict__ G, int64_t ld, int nrows)
    {
        const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
        const int ic = i < nrows ? i : nrows - 1;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            v[q] = G[ic + (int64_t)(kk0 + 8 * q) * ld];
    }
    // the same tile (ld = 64) straight from device-coherent memory: written by another workgroup of THIS launch with
    // write-through stores (panel_step_body, head-tile hand-over), possibly on another XCD behind another L2
    __device__ __forceinline__ void load_coherent(const double* G)
    {
        const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            v[q] = __hip_atomic_load(G + i + (int64_t)(kk0 + 8 * q) * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void store(double* __restrict__ T) const
    {
        const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            T[(kk0 + 8 * q) * PS + i] = v[q];
    }
};

// The 32 x 16 accumulator tile of one wave of k_panel_step (acc[m][n]: v_mfma_f64_4x4x4 layout, lane l on
// row 16 m + 4*((l>>2)&3) + (l>>4), column 4 n + (l&3)) re-arranged with cross-lane moves into the
// lane = row layout used for all C traffic: out[it] = element (row = lane & 31, column 2 it + (lane >> 5)).
// A global load/store in the MFMA layout touches 4 columns x 16 rows with neighbouring lanes in
// different columns and costs ~150 (load) / ~450 (store) cycles just to issue (gemm.hip, WaveTileC);
// in the row layout an instruction covers two whole 256-byte column pieces.
static __device__ __forceinline__ void wave_tile_to_rows(const double (&acc)[2][4], double (&out)[8], int lane)
{
    const int row = lane & 31;
    const int src_base = 16 * (row & 3) + 4 * ((row >> 2) & 3);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int src = src_base + ((2 * it + (lane >> 5)) & 3);
        const double v0 = __shfl(acc[0][it >> 1], src);
        const double v1 = __shfl(acc[1][it >> 1], src);
        out[it] = (row >> 4) ? v1 : v0;
    }
}

#define PANEL_PRE 3 // head tiles held in registers (nbo = 256 needs 3)
#ifndef PANEL_HANDOVER
#define PANEL_HANDOVER 1 // 1: head tiles are handed over through Hs + flags; 0: every workgroup re-derives them (round 1)
#endif

// dnext >= 0: the workgroup that owns rows dnext .. dnext+63 (the next OUTER panel's first diagonal block) also
// adds its L L^T to the 64 x 64 scratch Dacc (lane = row order of a workgroup's C tile; dinit: starts the sum) —
// and, when dfirst >= 0, the same product of the column block at dfirst (the panel's first, whose own step has
// no workgroup to spare; done in the step where that workgroup has the most slack).  k_upd_fused subtracts the
// sum from the block and only has to factor it.  (Not subtracted from A directly: the second stream's GEMMs
// may still be updating that block.)
static __device__ __forceinline__ void panel_step_body(double* __restrict__ A, int64_t lda, int64_t j0, int64_t M, int nt,
                                                       const double* __restrict__ Xt_cur, double* __restrict__ Xt_next,
                                                       int do_next, int* __restrict__ info, double* __restrict__ Hs,
                                                       int64_t dnext, int64_t dfirst, int dinit, double* __restrict__ Dacc,
                                                       gpe_epoch_t* hflag, gpe_epoch_t epoch, int spin_limit, const int bx)
{
    // one LDS array, carved: [Bx | T0 | T1]; workgroup 0 re-carves it as [Ls | Ltb | invd | sync | Xw]
    __shared__ __attribute__((aligned(16))) double lds[NB * XS + 2 * NB * PS];
    static_assert(NB * XS + DIAG_LTB + NB + 8 + DIAG_XW_DOUBLES <= NB * XS + 2 * NB * PS, "workgroup 0's carve fits");
    double* Bx = lds;
    double* T0 = lds + NB * XS;
    double* T1 = T0 + NB * PS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 16;
    const int b = bx;
    const int64_t r0 = j0 + NB, R0 = r0 + (int64_t)NB * b;
    const int nrows = (int)((M - R0 < NB) ? M - R0 : NB);
    const int tmax = (b < nt - 1) ? b : nt - 1;

    // every global load this workgroup needs before its first product goes out now: X, its own
    // tile, and the head tiles it will re-derive (one exposed memory latency instead of one per tile)
    PTS(0);
    TileRegs own, head[PANEL_PRE];
    own.load(A + R0 + j0 * lda, lda, nrows);
    double xv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
        xv[q] = Xt_cur[threadIdx.x + 512 * q];
#if !PANEL_HANDOVER
#pragma unroll
    for (int t = 0; t < PANEL_PRE; ++t)
        if (t <= tmax && t != b)
            head[t].load(A + r0 + (int64_t)NB * t + j0 * lda, lda, NB);
#endif
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = threadIdx.x + 512 * q;
        Bx[(e >> 6) * XS + (e & 63)] = xv[q]; // Bx[c][k] = X[c][k]
    }
    own.store(T0);
    // the C tile of the first update (for workgroup 0: the next diagonal block) is fetched now, under
    // the triangular solve, instead of at the top of the update loop
    // C tiles travel in the lane = row layout (wave_tile_to_rows): element it of a thread is
    // row wm + (lane & 31), column wn + 2 it + (lane >> 5) of the 64 x 64 tile
    const int crow = wm + (lane & 31), ccol = wn + (lane >> 5);
    const int crc = crow < nrows ? crow : nrows - 1;
    double c0v[8];
    if (tmax >= 0) {
#pragma unroll
        for (int it = 0; it < 8; ++it)
            c0v[it] = A[R0 + crc + (r0 + ccol + 2 * it) * lda];
    }
    __syncthreads();
    PTS(1);

    // 1. L_b = A_b L11^-T  (half-block form of the inverse, in place in T0)
    trsm_tile_full(T0, Bx, lane, wave);
    PTS(2);
    {
        // Row blocks b < nt are the "head" tiles other workgroups re-derive from A while this one
        // runs: they must not be overwritten in place here.  Their L goes to the scratch tile Hs[b]
        // and k_head_copy moves it into A after the panel's fused steps.
        const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int col = kk0 + 8 * q;
            const double v = T0[col * PS + i];
            if (b < nt) { // write-through: other XCDs read this tile during this launch (PANEL_HANDOVER)
                __hip_atomic_store(Hs + (int64_t)b * (NB * NB) + i + NB * col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // with the hand-over nobody re-derives anything from A's head rows: L goes into place as well and the
                // panel needs no k_head_copy behind it (a launch in front of every look-ahead update)
                if (hflag)
                    A[R0 + i + (j0 + col) * lda] = v;
            }
            else if (i < nrows)
                A[R0 + i + (j0 + col) * lda] = v;
        }
    }

    PTS(3);
#if PANEL_HANDOVER
    // Head tiles change hands instead of being re-derived by every workgroup (round 2; the stamps of tools/kbench_t showed
    // the last workgroup of a first step at 66 k cycles, 47 k of them three re-derived head tiles + updates, against 40 k for
    // workgroup 0 INCLUDING the diagonal block).  A head workgroup publishes: its tile is in Hs, device-wide, then
    // hflag[b] = this launch's epoch (a value no earlier launch used: the words are never reset).  Consumers need only
    // lower-numbered head tiles and the heads wait for nobody but lower-numbered heads, so with workgroups dispatched in
    // index order nobody can wait for a workgroup that is not running; the wait is bounded all the same (below).
    const bool mute = spin_limit < 0; // test hook (GPE_HANDOVER_FAULT): nobody publishes, every consumer gives up at once
    if (spin_limit < 0)
        spin_limit = -spin_limit;
    if (b < nt && hflag && !mute) {
        // The tile went out with device-scope (write-through) stores, the consumers read it and the flag with device-scope
        // loads: no release/acquire fence anywhere.  (An agent-scope release writes back the whole L2 of this XCD — every
        // dirty C tile of every workgroup on it: 7.5 k cycles when each wave issued one, 3 k for a single one, growing
        // with the number of updates in flight; tools/kbench_t.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's part of the tile is acknowledged
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(hflag + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // Every wave waits for a tile's word itself (all lanes read the same word: a wave-uniform spin) and then fetches its
    // eighth of the tile: no broadcast of "it is there" through LDS, no barrier, and the three tiles' loads overlap.  (One
    // thread polling the three words in turn + two barriers + the loads took 11 k cycles from "own L written" to "tiles in
    // registers", tools/kbench_t.)  The poll is bounded; a wave that runs out of patience reports it (info[2] = 1) and the
    // host runs the evaluation again without the hand-over (engine.hip, compute_finish) — its tile may be garbage by then.
    const bool handed = hflag != nullptr; // nullptr: no hand-over in this launch (GPE_PANEL_HANDOVER=0): re-derive
    gpe_epoch_t seen[PANEL_PRE];
#pragma unroll
    for (int t = 0; t < PANEL_PRE; ++t) // all words at once: a poll is a round trip to memory even when the word is set
        seen[t] = (handed && t <= tmax && t != b) ? __hip_atomic_load(hflag + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
#pragma unroll
    for (int t = 0; t < PANEL_PRE; ++t)
        if (t <= tmax && t != b) {
            if (handed) {
                int spins = 0;
                while (seen[t] != epoch) {
                    if (++spins > spin_limit) {
                        if (lane == 0)
                            info[2] = 1;
                        break;
                    }
                    seen[t] = __hip_atomic_load(hflag + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // the tile's loads may not be moved above the poll by the compiler (the hardware returns a wave's
                // loads in order; relaxed atomics alone do not order them in the language): a zero-cost fence
                asm volatile("" ::: "memory");
                head[t].load_coherent(Hs + (int64_t)t * (NB * NB));
            }
            else
                head[t].load(A + r0 + (int64_t)NB * t + j0 * lda, lda, NB);
        }
#else
    const bool handed = false;
#endif
    // 2. in-panel updates of this row block
    double cres[8]; // workgroup 0: the updated next diagonal block (lane = row layout)
#pragma unroll 1
    for (int t = 0; t <= tmax; ++t) {
        double* Cg = A + R0 + (r0 + (int64_t)NB * t) * lda;
        double cv[8];
#pragma unroll
        for (int it = 0; it < 8; ++it)
            cv[it] = (t == 0) ? c0v[it] : Cg[crc + (int64_t)(ccol + 2 * it) * lda];
        const double* Bop = T0;
        if (t == 0)
            PTS(10);
        if (t != b) { // head tile of another row block: recompute L_t = A_t X^T
            if (t == 0)
                head[0].store(T1);
            else if (t == 1)
                head[1].store(T1);
            else if (t == 2)
                head[2].store(T1);
            else {
                TileRegs late;
                if (handed) {
                    int spins = 0;
                    while (__hip_atomic_load(hflag + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch)
                        if (++spins > spin_limit) {
                            if (lane == 0)
                                info[2] = 1;
                            break;
                        }
                    asm volatile("" ::: "memory"); // as above: the tile's loads stay behind the poll
                    late.load_coherent(Hs + (int64_t)t * (NB * NB));
                }
                else
                    late.load(A + r0 + (int64_t)NB * t + j0 * lda, lda, NB);
                late.store(T1);
            }
            __syncthreads();
            if (!handed) {
                trsm_tile_full(T1, Bx, lane, wave);
            }
            Bop = T1;
        }
        double a2[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                a2[m][n] = 0.0;
        if (t == 0)
            PTS(11);
        mm64<false>(T0, Bop, wm, wn, lane, a2);
        if (t == 0)
            PTS(12);
        double a2r[8];
        wave_tile_to_rows(a2, a2r, lane);
        if (t == 0)
            PTS(13);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const double v = cv[it] - a2r[it];
            if (t == 0)
                cres[it] = v;
            // workgroup 0 factors this tile next and writes L over it: no need to store the update
            if (crow < nrows && !(b == 0 && do_next))
                Cg[crow + (int64_t)(ccol + 2 * it) * lda] = v;
        }
        if (t == 0)
            PTS(14);
        __syncthreads(); // T1 is free again
        if (t == 0)
            PTS(15);
    }

    // 2b. the piece(s) of the next outer panel's first diagonal block that this workgroup can provide
    if (dnext >= 0 && R0 == dnext) {
        double pr[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                pr[m][n] = 0.0;
        double cd[8];
        if (!dinit) { // a later step of the panel: add to the running sum
#pragma unroll
            for (int it = 0; it < 8; ++it)
                cd[it] = Dacc[threadIdx.x + 512 * it];
        }
        mm64<false>(T0, T0, wm, wn, lane, pr); // own L (this step's column block) times its transpose
        if (dfirst >= 0) {                      // and the same rows of the column block at dfirst
            TileRegs lf;
            lf.load(A + dnext + dfirst * lda, lda, NB);
            __syncthreads(); // T1's last readers (head-tile products) are done
            lf.store(T1);
            __syncthreads();
            mm64<false>(T1, T1, wm, wn, lane, pr);
        }
        double prr[8];
        wave_tile_to_rows(pr, prr, lane);
#pragma unroll
        for (int it = 0; it < 8; ++it)
            Dacc[threadIdx.x + 512 * it] = (dinit ? 0.0 : cd[it]) + prr[it];
    }

    PTS(4);
    // 3. workgroup 0: factor the next diagonal block (block t = 0 of its own rows) and invert its halves
    if (b != 0 || !do_next)
        return;
    double* Ls = lds;
    double* Ltb = Ls + NB * XS;
    double* invd = Ltb + DIAG_LTB;
#pragma unroll
    for (int it = 0; it < 8; ++it)
        Ls[crow * XS + ccol + 2 * it] = cres[it];
    {
        DiagSync* sy = reinterpret_cast<DiagSync*>(invd + NB);
        diag_flow_init(sy);
        __syncthreads();
        PTS(5);
        diag_flow(Ls, Ltb, invd, sy, A + r0 + r0 * lda, lda, Xt_next, info, r0, wave, lane, invd + NB + 8);
        PTS(6);
        return;
    }
}
// entry points: single GP (the round-1 kernel, unchanged) / batched.  Batched: blockIdx.x = b * G + gp, so that workgroup 0
// of every GP (the one that goes on to factor the next diagonal block, twice as long as the others) is dispatched first
// instead of trailing each GP's rows.
__global__ __launch_bounds__(512) void k_panel_step(double* __restrict__ A, int64_t lda, int64_t j0, int64_t M, int nt,
                                                    const double* __restrict__ Xt_cur, double* __restrict__ Xt_next,
                                                    int do_next, int* __restrict__ info, double* __restrict__ Hs,
                                                    int64_t dnext, int64_t dfirst, int dinit, double* __restrict__ Dacc,
                                                    gpe_epoch_t* hflag, gpe_epoch_t epoch, int spin_limit)
{
    panel_step_body(A, lda, j0, M, nt, Xt_cur, Xt_next, do_next, info, Hs, dnext, dfirst, dinit, Dacc, hflag, epoch, spin_limit,
                    (int)blockIdx.x);
}
__global__ __launch_bounds__(512) void k_panel_step_b(double* __restrict__ A, int64_t lda, int64_t j0, int64_t M, int nt,
                                                      const double* __restrict__ Xt_cur, double* __restrict__ Xt_next,
                                                      int do_next, int* __restrict__ info, double* __restrict__ Hs,
                                                      int64_t dnext, int64_t dfirst, int dinit, double* __restrict__ Dacc,
                                                      gpe_epoch_t* hflag, gpe_epoch_t epoch, int spin_limit,
                                                      const BatchTab* __restrict__ bt)
{
    const int G = bt->G, gp = (int)blockIdx.x % G;
    A = bt_rebase(bt, gp, A);
    Xt_cur = bt_rebase(bt, gp, Xt_cur);
    Xt_next = bt_rebase(bt, gp, Xt_next);
    info = bt_rebase(bt, gp, info);
    Hs = bt_rebase(bt, gp, Hs);
    Dacc = bt_rebase(bt, gp, Dacc);
    hflag = bt_rebase(bt, gp, hflag);
    panel_step_body(A, lda, j0, M, nt, Xt_cur, Xt_next, do_next, info, Hs, dnext, dfirst, dinit, Dacc, hflag, epoch, spin_limit,
                    (int)blockIdx.x / G);
}

// ---------------------------------------------------------------------------------------------
// k_panel256 (round 3) — ALL 64-column steps of a 256-column outer panel in ONE launch, as data flow between the
// workgroups.  The step-by-step form pays, per step, a launch boundary plus the serial sequence
//   [diagonal block | head tiles | everybody's solve | everybody's updates]
// (21 + 20 + 19 us for the three steps of a panel at N = 4096 and ~3 us between launches), although the only true chain is
//   X_s -> L(s, s) = A(s, s) X_s^T -> A(s, s+1) -= L L^T -> factor -> X_{s+1}                (~12 us per step).
// Here workgroup b owns the 64-row strip R_b = rows p0 + 64 (b + 1) .. +63 of the panel for the whole launch and keeps its
// (up to four) 64 x 64 tiles in REGISTERS between the steps: every tile is read once and written once, as L.
//   step s (column block s of the panel), strips b >= s:
//     X_s (s = 0: the diagonal block at p0 was factored by the launch before; s > 0: polled, below)  ->  L_bs = A_bs X_s^T
//     strips b <= 2 publish L_bs (a "head tile": the rows of column block b + 1)
//     A_bc -= L_bs L_{c-1,s}^T for the strip's remaining column blocks c = s+1 .. min(3, b+1)
//     strip b = s now holds the finished diagonal block of column block s + 1: it factors it (diag_flow), X_{s+1} goes out
//       quarter by quarter while it is being computed — and the strip is done.
// Nothing inside the launch is handed over with a flag: block inverses and head tiles are stored, with device-scope stores,
// into buffers that hold an all-ones pattern when the launch starts, and their consumers poll the values (P256::S22 / HP,
// PolledTile, poll_one; diag_flow.h: DiagEarly).  The launch arms the other buffer of the pair for the launch after it.
// A strip only ever waits for lower-numbered strips (X_s comes from strip s - 1 <= b - 1, head tiles from strips < b), so
// with workgroups dispatched in index order nobody waits for a workgroup that is not running (dev.h, requirement (1));
// the polls are bounded all the same and a timeout is reported exactly like k_panel_step's (info[2], the host re-runs).
// From step 1 on every strip solves in the half-block form of the inverse, in two phases (p256_half_solve): three quarters of
// the solve, and the first k-half of the factoring strip's update, run while the previous block is still being factored.
// dnext >= 0: the strip of rows dnext (the next panel's first diagonal block) also leaves sum_s L_bs L_bs^T in Dacc for
// k_upd_fused.  Full 64-column blocks, nbo = 256 only; everything else goes the step-by-step way.
// ---------------------------------------------------------------------------------------------
// head tile h = P256_H(s, t): the tile of strip t at step s (t = s..2), six per panel
#define P256_H(s, t) ((s) == 0 ? (t) : ((s) == 1 ? 2 + (t) : 5))

#define P256_POLLED_S (9 * 1024)                          // X11 | L21 | X22 of three diagonal blocks
#define P256_POLLED_DOUBLES (P256_POLLED_S + 6 * NB * NB) // ... and six head tiles: 33,792 doubles per buffer
struct P256 {
    double* A;
    int64_t lda, p0, R0;
    double* Xt;
    int* info;
    int spin_limit, nrows;
    bool mute, want_d;
    double *Bx, *T0, *T1, *T2;
    double* S22; // the polled copies of X11 | L21 | X22 of diagonal blocks 1..3 (3 x 3 x 1024 doubles), armed by the launch before
    double* HP;  // ... and of the six head tiles (P256_H), 4096 doubles each: they travel the same way, no flag, no acknowledgement
};

// Waiting for a polled block costs memory traffic: 60 strips x 512 threads each re-reading their 2..8 words every microsecond
// is 10^5 uncached transactions per look — it slows everybody's loads (measured: 1.452 -> 1.436 ms per evaluation without it).
// So a wave first watches ONE word of the block, the same for all its lanes (one transaction per look, with a pause), chosen
// among the last to be written, and only then fetches and checks its own.
static __device__ __forceinline__ void poll_one(const double* p, int spin_limit, int* __restrict__ info)
{
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(p);
    int spins = 0;
    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ~0ull) {
        if (++spins > spin_limit) {
            if ((threadIdx.x & 63) == 0)
                info[2] = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
}

// a 64 x 64 tile (ld 64) that another workgroup of this launch is writing, or has written, over an all-ones pattern:
// thread t holds elements (t & 63, (t >> 6) + 8 q) as in TileRegs
// Round 4: the FIRST look at a polled block is an ordinary (cacheable) load, only the re-reads of words that still showed the
// pattern are device-scope.  A device-scope load is served by the memory side, whatever the XCD's L2 holds: every one of the
// nb - s workgroups that use an L tile fetched its 32 KB over the fabric — 2.8 GB per batch of eight N = 2048 factorisations,
// 0.85 GB in the tall launch of N = 4096, both at the ~2 TB/s such loads reach (round-4 measurement: eight interleaved
// factorisations took 3.3x one).  The protocol makes the cached look safe: inside a launch a slot only ever changes from the
// pattern to its final value, word by word (the launch before armed it, and kernel boundaries write back / invalidate the
// L2s), so whatever a cache line holds, a word that is not the pattern is final; a word that is goes the device-scope way.
#ifndef POLL_CACHED
#define POLL_CACHED 1
#endif
#if POLL_CACHED
#define POLL_FIRST_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#else
#define POLL_FIRST_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
struct PolledTile {
    unsigned long long b[8];
    __device__ __forceinline__ void issue(const double* G)
    {
        const unsigned long long* g = reinterpret_cast<const unsigned long long*>(G) + (threadIdx.x & 63) + (threadIdx.x >> 6) * NB;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            b[q] = __hip_atomic_load(g + 8 * q * NB, __ATOMIC_RELAXED, POLL_FIRST_SCOPE);
    }
    __device__ __forceinline__ void finish(const double* G, int spin_limit, int* __restrict__ info)
    {
        const unsigned long long SENT = ~0ull;
        const unsigned long long* g = reinterpret_cast<const unsigned long long*>(G) + (threadIdx.x & 63) + (threadIdx.x >> 6) * NB;
        int spins = 0;
        while (b[0] == SENT || b[1] == SENT || b[2] == SENT || b[3] == SENT || b[4] == SENT || b[5] == SENT || b[6] == SENT
               || b[7] == SENT) {
            if (++spins > spin_limit) {
                info[2] = 1;
                break;
            }
            poll_one(G + NB * NB - 1, spin_limit, info);
            if (b[0] == SENT) b[0] = __hip_atomic_load(g + 0 * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b[1] == SENT) b[1] = __hip_atomic_load(g + 8 * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b[2] == SENT) b[2] = __hip_atomic_load(g + 16 * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b[3] == SENT) b[3] = __hip_atomic_load(g + 24 * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b[4] == SENT) b[4] = __hip_atomic_load(g + 32 * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b[5] == SENT) b[5] = __hip_atomic_load(g + 40 * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b[6] == SENT) b[6] = __hip_atomic_load(g + 48 * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b[7] == SENT) b[7] = __hip_atomic_load(g + 56 * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __device__ __forceinline__ void store(double* __restrict__ T) const
    {
        const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            T[(kk0 + 8 * q) * PS + i] = __longlong_as_double((long long)b[q]);
    }
};

// The strip that factors next solves against the block inverse in the half-block form, in two phases: three quarters of its
// solve and half of its one update run while the previous strip is still factoring (X11 and L21 of that block leave it half-way
// through, diag_flow.h: DiagEarly); what is left behind the arrival of X22 is one 64 x 32 x 32 product and the other half of the
// update.  T (64 x 64, [kk][i], stride PS) <- own L^-T; acc += (own L^-T)(own L^-T)^T over both halves of k.
// Every strip solves this way from step 1 on (PUBHALF / acc_on: the factoring strip publishes the first half of its tile after
// phase A and accumulates its update; the strip of the next panel's first diagonal block accumulates its piece of that block;
// the others only solve): what a strip still has to do once the last rows of X are out is a quarter of the solve.
// Sq: the block's polled quarters (X11 | L21 | X22, 1024 doubles each); pub (PUBHALF): where the strip's own tile goes, polled
template <int S, bool PUBHALF>
static __device__ __forceinline__ void p256_half_solve(const P256& x, double* __restrict__ T, double* __restrict__ Ld,
                                                       const double (&own)[8], double (&a2)[2][4], const bool acc_on,
                                                       const double* __restrict__ Sq, double* __restrict__ pub)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 16; // the 64 x 64 product's wave tile
    const int hn = (wave >> 1) * 8;                        // the half-block products': column within the 32-column half
    const int crow = wm + (lane & 31), ccol = wn + (lane >> 5);
    const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4), dcol = lane & 3;
    double* Bx = x.Bx;
    P2TS(6 * S + 0);
#pragma unroll
    for (int it = 0; it < 8; ++it)
        T[(ccol + 2 * it) * PS + crow] = own[it];
    // ---- phase A: X11 and L21, polled value by value (diag_flow.h: DiagEarly) ----
    const unsigned long long SENT = ~0ull;
    const unsigned long long* Sp = reinterpret_cast<const unsigned long long*>(Sq) + threadIdx.x;
    {
        unsigned long long b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) // X11: e, e + 512; L21: 1024 + e, 1024 + e + 512   (first look: cacheable, see PolledTile)
            b[q] = __hip_atomic_load(Sp + 512 * q, __ATOMIC_RELAXED, POLL_FIRST_SCOPE);
        int spins = 0;
        while (b[0] == SENT || b[1] == SENT || b[2] == SENT || b[3] == SENT) {
            if (++spins > x.spin_limit) {
                x.info[2] = 1;
                break;
            }
            poll_one(Sq + 1023, x.spin_limit, x.info);            // X11's last row
            poll_one(Sq + 1024 + 32 * 31, x.spin_limit, x.info);  // L21's last column
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (b[q] == SENT)
                    b[q] = __hip_atomic_load(Sp + 512 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        P2TS(6 * S + 1);
        TTS2(x, acc_on && PUBHALF, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = threadIdx.x + 512 * q; // X11: e = k + 32 c ; L21: e = c + 32 k
            Bx[(e >> 5) * XS + (e & 31)] = __longlong_as_double((long long)b[q]);
            Ld[(e & 31) * XS + (e >> 5)] = __longlong_as_double((long long)b[2 + q]); // Ld[c][k] = L21[c][k]
        }
    }
    __syncthreads();
    double y1[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    mmk<true, 32, 2>(T, 0, Bx, 0, wm, hn, lane, y1); // Y1 = T1 X11^T
    __syncthreads();                                 // all reads of T[:, 0:32] done
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            T[(hn + 4 * n + dcol) * PS + wm + 16 * m + drow] = y1[m][n];
    __syncthreads();
    double u[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    mmk<true, 32, 2>(T, 0, Ld, 0, wm, hn, lane, u); // Y1 L21^T
    double t2[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            t2[m][n] = T[(32 + hn + 4 * n + dcol) * PS + wm + 16 * m + drow] - u[m][n];
    __syncthreads(); // every wave has read its part of T[:, 32:64]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            T[(32 + hn + 4 * n + dcol) * PS + wm + 16 * m + drow] = t2[m][n];
    if constexpr (PUBHALF) { // columns 0..31 of the strip's L tile are final: its head-tile copy starts its way now (the rest follows behind phase B)
        const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = kk0 + 8 * q;
            __hip_atomic_store(pub + i + NB * col, T[col * PS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc_on)
        mmk<false, 32, 4>(T, 0, T, 0, wm, wn, lane, a2); // the product's first half: Y1 Y1^T (columns 0..31 of T are final)
    P2TS(6 * S + 2);
    TTS2(x, acc_on && PUBHALF, 1);
    // ---- phase B: X22 ----
    {
        unsigned long long b0 = __hip_atomic_load(Sp + 2048, __ATOMIC_RELAXED, POLL_FIRST_SCOPE);
        unsigned long long b1 = __hip_atomic_load(Sp + 2048 + 512, __ATOMIC_RELAXED, POLL_FIRST_SCOPE);
        int spins = 0;
        while (b0 == SENT || b1 == SENT) {
            if (++spins > x.spin_limit) {
                x.info[2] = 1;
                break;
            }
            // (Round 4 let the one workgroup the chain waits for watch its own words instead of poll_one's: no difference,
            // 798.9 against 797.9 evaluations/s — the second look is not what a hop costs.)
            poll_one(Sq + 2048 + 1023, x.spin_limit, x.info); // X22's last row
            if (b0 == SENT)
                b0 = __hip_atomic_load(Sp + 2048, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b1 == SENT)
                b1 = __hip_atomic_load(Sp + 2048 + 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        TTS2(x, acc_on && PUBHALF, 2);
        const int e0 = threadIdx.x, e1 = threadIdx.x + 512; // e = k + 32 c
        Bx[(32 + (e0 >> 5)) * XS + 32 + (e0 & 31)] = __longlong_as_double((long long)b0);
        Bx[(32 + (e1 >> 5)) * XS + 32 + (e1 & 31)] = __longlong_as_double((long long)b1);
    }
    __syncthreads(); // X22 is in LDS (and T[:, 32:64] complete)
    P2TS(6 * S + 3);
    TTS2(x, acc_on && PUBHALF, 3);
    double y2[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    mmk<true, 32, 2>(T, 32, Bx + 32 * XS + 32, 0, wm, hn, lane, y2); // Y2 = T2 X22^T
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            T[(32 + hn + 4 * n + dcol) * PS + wm + 16 * m + drow] = y2[m][n];
    __syncthreads();
    TTS2(x, acc_on && PUBHALF, 4);
    if constexpr (PUBHALF) { // ... and the other 32 columns: the product below covers most of their way
        const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
#pragma unroll
        for (int q = 4; q < 8; ++q) {
            const int col = kk0 + 8 * q;
            __hip_atomic_store(pub + i + NB * col, T[col * PS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc_on)
        mmk<false, 32, 4>(T, 32, T, 32, wm, wn, lane, a2); // the product's second half
    TTS2(x, acc_on && PUBHALF, 5);
}

// the update of a strip's tile of column block T + 1 with the step's tile of strip T (its own: in TT; another strip's: polled)
template <int S, int ROLE, int T>
static __device__ __forceinline__ void p256_update(const P256& x, const double* __restrict__ TT, PolledTile (&hd)[3],
                                                   double (&cv)[4][8], int& nb)
{
    constexpr int CMAX = ROLE < 3 ? ROLE + 1 : 3;
    if constexpr (T >= S && T <= 2 && T + 1 <= CMAX) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wm = (wave & 1) * 32, wn = (wave >> 1) * 16;
        const double* Bop = TT;
        if constexpr (T != ROLE) {
            double* buf = (nb & 1) ? x.T2 : x.T1;
            ++nb;
            hd[T].finish(x.HP + (int64_t)P256_H(S, T) * (NB * NB), x.spin_limit, x.info);
            hd[T].store(buf);
            __syncthreads();
            Bop = buf;
        }
        double a2[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                a2[m][n] = 0.0;
        mm64<false>(TT, Bop, wm, wn, lane, a2);
        double a2r[8];
        wave_tile_to_rows(a2, a2r, lane);
#pragma unroll
        for (int it = 0; it < 8; ++it)
            cv[T + 1][it] -= a2r[it];
    }
}

// One step of one strip.  ROLE = 0..2: the strip with that index (it factors the diagonal block of column block ROLE + 1 at the
// end of step ROLE and is done); ROLE = 3: any strip below the panel's own 256 rows.  Everything about the role is a compile-
// time constant, so that each role's code holds exactly the tiles it needs (the factorisation alone wants 192 VGPRs).
template <int S, int ROLE>
static __device__ __forceinline__ void p256_step(const P256& x, double (&cv)[4][8], double (&pr)[2][4])
{
    constexpr int CMAX = ROLE < 3 ? ROLE + 1 : 3; // last column block of the panel the strip has a tile in
    constexpr bool HEAD = ROLE <= 2;              // other strips need this strip's tile of every step
    constexpr bool CHAIN = ROLE == S && ROLE < 3; // the strip that factors next: everything it does is on the panel's critical path
    if constexpr (ROLE >= S) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wm = (wave & 1) * 32, wn = (wave >> 1) * 16;
        const int crow = wm + (lane & 31), ccol = wn + (lane >> 5);
        // the strip's L tile of this step: the factoring strip keeps it in T1, which its factorisation (re-carving [Bx | T0])
        // leaves alone — the tile's copy into the matrix waits until the factorisation is over
        double* const TT = CHAIN ? x.T1 : x.T0;
        if constexpr (CHAIN && S > 0) {
            double a2c[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
            p256_half_solve<S, true>(x, TT, x.T0, cv[S], a2c, true, x.S22 + (S - 1) * 3072, x.HP + (int64_t)P256_H(S, S) * (NB * NB));
            double a2r[8];
            wave_tile_to_rows(a2c, a2r, lane);
#pragma unroll
            for (int it = 0; it < 8; ++it)
                cv[S + 1][it] -= a2r[it];
            P2TS(6 * S + 5);
        }
        else {
            if constexpr (S > 0)
                p256_half_solve<S, false>(x, TT, x.T2, cv[S], pr, !HEAD && x.want_d, x.S22 + (S - 1) * 3072, nullptr);
            else {
                // ---- X_0 (from the launch before) and this strip's tile of column block 0 into LDS ----
                P2TS(6 * S + 0);
                double xv[8];
                const double* Xs = x.Xt;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    xv[q] = Xs[threadIdx.x + 512 * q];
#pragma unroll
                for (int it = 0; it < 8; ++it)
                    TT[(ccol + 2 * it) * PS + crow] = cv[S][it];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = threadIdx.x + 512 * q;
                    x.Bx[(e >> 6) * XS + (e & 63)] = xv[q]; // Bx[c][k] = X[c][k]
                }
                __syncthreads();
                P2TS(6 * S + 2);
                trsm_tile_full(TT, x.Bx, lane, wave); // L_b0, ends with a barrier
                P2TS(6 * S + 3);
            }
            {
                const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
                double* Ag = x.A + x.R0 + (x.p0 + (int64_t)NB * S) * x.lda;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int col = kk0 + 8 * q;
                    const double v = TT[col * PS + i];
                    if constexpr (HEAD && !(CHAIN && S > 0)) // (the factoring strip's went out in two halves inside its solve)
                        __hip_atomic_store(x.HP + (int64_t)P256_H(S, ROLE) * (NB * NB) + i + NB * col, v, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT); // polled by the other strips: no flag, no acknowledgement
                    if (i < x.nrows)
                        Ag[i + (int64_t)col * x.lda] = v;
                }
            }
            if constexpr (HEAD) {
            }
            else if (S == 0 && !HEAD && x.want_d) // this strip's piece of the next panel's first diagonal block (later steps: inside the solve)
                mm64<false>(TT, TT, wm, wn, lane, pr);
            P2TS(6 * S + 4);
            if constexpr (S < 3) {
                // ---- updates of the strip's remaining column blocks c = S+1 .. CMAX with the tile of strip t = c - 1 ----
                // All tiles are asked for at once; each is completed (polled) where it is used.  Order: the strip's own tile first
                // (nothing to wait for), then highest t first — the tile of strip t = S, the one that factors next, is the last
                // to be complete.  The tiles alternate between two LDS buffers: one barrier per update.
                PolledTile hd[3];
#pragma unroll
                for (int t = 2; t >= S; --t)
                    if (t != ROLE && t + 1 <= CMAX)
                        hd[t].issue(x.HP + (int64_t)P256_H(S, t) * (NB * NB));
                int nb = 0;
                if constexpr (HEAD && !CHAIN)
                    p256_update<S, ROLE, ROLE>(x, TT, hd, cv, nb); // with its own tile first: the polled ones are on their way meanwhile
                if constexpr (ROLE != 2)
                    p256_update<S, ROLE, 2>(x, TT, hd, cv, nb);
                if constexpr (S <= 1 && ROLE != 1)
                    p256_update<S, ROLE, 1>(x, TT, hd, cv, nb);
                if constexpr (S == 0 && ROLE != 0)
                    p256_update<S, ROLE, 0>(x, TT, hd, cv, nb);
                if constexpr (CHAIN)
                    p256_update<S, ROLE, ROLE>(x, TT, hd, cv, nb); // (S = 0 only: later steps update inside the solve)
                if constexpr (!CHAIN)
                    __syncthreads(); // T0 and the buffers are free again
            }
            P2TS(6 * S + 5);
        }
        if constexpr (CHAIN) {
            // ---- tile S + 1 is the finished diagonal block of column block S + 1 ----
            __syncthreads();   // [Bx | T0] have no readers left
            double* Ls = x.Bx; // [Ls | Ltb | invd | sync | Xw] re-carved over [Bx | T0], as in k_panel_step; T1 = this strip's L tile
            double* Ltb = Ls + NB * XS;
            double* invd = Ltb + DIAG_LTB;
#pragma unroll
            for (int it = 0; it < 8; ++it)
                Ls[crow * XS + ccol + 2 * it] = cv[S + 1][it];
            DiagSync* sy = reinterpret_cast<DiagSync*>(invd + NB);
            diag_flow_init(sy);
            __syncthreads();
            P2TS(26);
            DiagEarly ea;
            ea.mute = x.mute;
            ea.S = x.S22 + S * 3072;
            diag_flow(Ls, Ltb, invd, sy, x.A + x.R0 + x.R0 * x.lda, x.lda, x.Xt + (S + 1) * (NB * NB), x.info, x.R0, wave, lane,
                      invd + NB + 8, &ea);
            P2TS(27);
            __syncthreads(); // (the tile in T1 is still to be copied into the matrix: after every wave's part of the factorisation)
            P2TS(28);
            P2TS(29);
            if constexpr (S > 0) { // this strip's L tile of the step into the matrix: nobody reads it there before the launch ends
                const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
                double* Ag = x.A + x.R0 + (x.p0 + (int64_t)NB * S) * x.lda;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int col = kk0 + 8 * q;
                    Ag[i + (int64_t)col * x.lda] = TT[col * PS + i];
                }
            }
        }
    }
}

template <int ROLE>
static __device__ __forceinline__ void p256_strip(const P256& x, double* __restrict__ Dacc)
{
    constexpr int CMAX = ROLE < 3 ? ROLE + 1 : 3;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 16;
    // the strip's tiles, lane = row layout (wave_tile_to_rows): element it of a thread is row wm + (lane & 31), column
    // wn + 2 it + (lane >> 5) of the 64 x 64 tile
    const int crow = wm + (lane & 31), ccol = wn + (lane >> 5);
    const int crc = crow < x.nrows ? crow : x.nrows - 1;
    double cv[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int it = 0; it < 8; ++it)
            cv[c][it] = (c <= CMAX) ? x.A[x.R0 + crc + (x.p0 + (int64_t)NB * c + ccol + 2 * it) * x.lda] : 0.0;
    double pr[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            pr[m][n] = 0.0;
    p256_step<0, ROLE>(x, cv, pr);
    p256_step<1, ROLE>(x, cv, pr);
    p256_step<2, ROLE>(x, cv, pr);
    p256_step<3, ROLE>(x, cv, pr);
    if constexpr (ROLE == 3) {
        if (x.want_d) {
            double prr[8];
            wave_tile_to_rows(pr, prr, lane);
#pragma unroll
            for (int it = 0; it < 8; ++it)
                Dacc[threadIdx.x + 512 * it] = prr[it];
        }
    }
}

__global__ __launch_bounds__(512) void k_panel256(double* __restrict__ A, int64_t lda, int64_t p0, int64_t M,
                                                  double* __restrict__ Xt, int* __restrict__ info, int64_t dnext,
                                                  double* __restrict__ Dacc, int spin_limit, double* __restrict__ S22,
                                                  double* __restrict__ S22_next)
{
    __shared__ __attribute__((aligned(16))) double lds[NB * XS + 3 * NB * PS]; // [Bx | T0 | T1 | T2]: 156,672 B
    static_assert(NB * XS + DIAG_LTB + NB + 8 + DIAG_XW_DOUBLES <= NB * XS + NB * PS, "the factoring strips' carve fits into [Bx | T0]");
    const int b = (int)blockIdx.x;
    P256 x;
    x.A = A;
    x.lda = lda;
    x.p0 = p0;
    x.R0 = p0 + (int64_t)NB * (b + 1);
    x.Xt = Xt;
    x.info = info;
    x.mute = spin_limit < 0; // test hook (GPE_HANDOVER_FAULT): nobody publishes, every consumer gives up at once
    x.spin_limit = spin_limit < 0 ? -spin_limit : spin_limit;
    x.nrows = (int)((M - x.R0 < NB) ? M - x.R0 : NB);
    x.want_d = dnext >= 0 && x.R0 == dnext;
    x.Bx = lds;
    x.T0 = lds + NB * XS;
    x.T1 = x.T0 + NB * PS;
    x.T2 = x.T1 + NB * PS;
    x.S22 = S22;
    x.HP = S22 + P256_POLLED_S;
    { // the polled copies of the NEXT launch start from the all-ones pattern (this launch's were armed by the one before: same
      // stream, complete before this one began); a quarter each for the last four strips
        unsigned long long* nx = reinterpret_cast<unsigned long long*>(S22_next);
        constexpr int QUARTER = P256_POLLED_DOUBLES / 4;
#pragma unroll
        for (int part = 0; part < 4; ++part) {
            const int owner = (int)gridDim.x - 1 - part > 0 ? (int)gridDim.x - 1 - part : 0;
            if (b == owner)
                for (int idx = threadIdx.x; idx < QUARTER; idx += 512)
                    nx[part * QUARTER + idx] = ~0ull;
        }
    }
    P2TS(30);
    switch (b) {
    case 0: p256_strip<0>(x, Dacc); break;
    case 1: p256_strip<1>(x, Dacc); break;
    case 2: p256_strip<2>(x, Dacc); break;
    default: p256_strip<3>(x, Dacc); break;
    }
    P2TS(31);
}

static std::atomic<gpe_epoch_t> g_handover_epoch{0}; // a value no earlier launch of this process has used; 64 bits: never wraps

void launch_panel256(hipStream_t s, double* A, int64_t lda, int64_t p0, int64_t M, double* Xt, int* info, int64_t dnext,
                     double* Dacc, double* S22, double* S22_next, hipEvent_t stop)
{
    static const bool fault = getenv("GPE_HANDOVER_FAULT") && atoi(getenv("GPE_HANDOVER_FAULT")) != 0;
    const int spin_limit = fault ? -16 : GPE_FLOW_SPIN_LIMIT;
    const int64_t rows = M - (p0 + NB);
    if (rows <= 0)
        return;
    const dim3 grid((unsigned)((rows + NB - 1) / NB)), block(512);
    FlowGate gate(s); // (its strips poll each other inside the launch: dev.h)
    if (stop)
        GPE_LAUNCH_STOP("k_panel256", k_panel256, grid, block, 0, s, stop, A, lda, p0, M, Xt, info, dnext, Dacc, spin_limit, S22, S22_next);
    else
        GPE_LAUNCH(k_panel256, grid, block, 0, s, A, lda, p0, M, Xt, info, dnext, Dacc, spin_limit, S22, S22_next);
}

// ---------------------------------------------------------------------------------------------
// k_tail (round 3) — the LAST T <= 1024 columns of the factorisation (or all of it when N <= 1024) as ONE launch: a tiled
// data-flow Cholesky.  The last four outer panels of N = 4096 hold 1.5 % of the flops and took 20 % of the time: each is a
// k_panel256 of 4-16 strips (~48 us: three diagonal blocks one after the other), a fused update for a handful of tiles
// (~16 us) and two launch boundaries, although the whole remaining matrix — 136 tiles of 64 x 64 — fits on the chip with one
// workgroup per tile.  Here workgroup (b, c) owns tile (b, c) of the lower triangle (b >= c; row strip nt = the right-hand-side
// rows) and keeps it in registers for the whole launch:
//   steps s = 0 .. c-1:  tile -= L(b, s) L(c, s)^T, both operands polled from the owners of those tiles (PolledTile; the next
//                        step's operands are asked for before this step's product)
//   step c, b == c:      the diagonal block is complete: factor it (diag_flow), its inverse leaves in polled quarters
//   step c, b >  c:      L(b, c) = tile X_c^T in the half-block form, in two phases (tail_tile_solve; the chain workgroup:
//                        tail_chain_updates_and_crossing), published in two halves
// Workgroups are numbered column by column, the diagonal tile first: every wait is for a lower-numbered workgroup.  The chain
// diag(c) -> X_c -> L(c+1, c) -> last update of tile (c+1, c+1) -> diag(c+1) is what k_panel256's is, without the fused
// updates and launch boundaries in between.  Polled buffers: LP (a 4096-double slot per tile, blockIdx order) and SP (3072 doubles
// per diagonal block), all-ones when the launch starts; every workgroup arms its own slot of the OTHER pair for the next launch.
// ---------------------------------------------------------------------------------------------
// Round 4: the launch is no longer tied to the END of the matrix.  A "tall" launch factors the nt tile columns t0 .. t1 of a
// panel that has nfull >= nt full row strips under its first row (rows t0 .. N64) plus the right-hand-side strip: the whole
// 1536-column head of an N = 4096 factorisation is one such launch (24 tile columns x 64 row strips), one k = 1536 update and
// the closing launch (nfull = nt) follow — no 256-column panels, no look-ahead stream.  Batched (k_tail_b): the tiles of G
// members interleave in the 1-D grid (id = tile * G + member), so that the G chains advance side by side and every wait is
// still for a lower-numbered workgroup.
struct TailArgs {
    double* A;
    int64_t lda, t0; // the launch starts at row / column t0
    int nt, nb;      // tile columns; row strips (nfull, + 1 for the right-hand-side rows)
    int nfull;       // full 64-row strips (>= nt; == nt for the closing launch)
    int rhs_rows;
    double* Xt;      // inverse of the diagonal block at t0 (the others follow at + 4096 each)
    int* info;
    double *LP, *SP, *LPn, *SPn;
    int spin_limit;
    const int* order; // dispatch order: workgroup w works on tile (b, c) = (order[2 w], order[2 w + 1]); null: column by column
    // gen (Xg != null): the launch GENERATES its tiles of K from the samples instead of reading them from A — the kernel matrix
    // is never written for the columns this launch factors (kernel/kernel.hpp:81-84 with the functors of kfun_fast.h, the
    // pair formula and summation order of kbuild.hip); rows >= Ns of the last strip are obs_mean's rows, read from Om
    const double* Xg; // SoA samples, Xg[d * ldx + i]
    int64_t ldx, Ns;  // Ns: samples (rows below Ns in the last strip: right-hand sides)
    const double* Om; // obs_mean, Om[i + p * ldom]
    int64_t ldom;
    double* Al;       // optional: the backward sweep's output, pre-filled with its sentinel here (what the build launch does)
    int64_t ldal;
    int P;
};
// LDS of a k_tail workgroup: two pairs of operand tiles [A0 | B0 | A1 | B1] (40 KB each: all of the CU's 160 KB) for the pipelined
// products of the update loop; behind it the carve of the solve and the factorisation (CH_*, further down)
#define TAIL_LDS_DOUBLES (4 * NB * PS)
static_assert(TAIL_LDS_DOUBLES >= NB * XS + 3 * NB * PS && TAIL_LDS_DOUBLES * 8 <= 160 * 1024, "k_tail LDS carve");
static __device__ __forceinline__ int tail_tile_id(int nb, int b, int c) { return c * nb - (c * (c - 1)) / 2 + (b - c); }

// element it of a thread's tile slice: global row I (clamped into the strip by the caller), global columns J0 + 2 it
static __device__ __forceinline__ void tail_gen_tile(const TailArgs& a, const KParams* __restrict__ kp, int64_t I, int64_t J0,
                                                     double (&out)[8])
{
    if (I >= a.Ns) { // a right-hand-side row: obs_mean^T
        const double* om = a.Om + (I - a.Ns) * a.ldom;
#pragma unroll
        for (int it = 0; it < 8; ++it)
            out[it] = om[J0 + 2 * it];
        return;
    }
    double z[8];
#pragma unroll
    for (int it = 0; it < 8; ++it)
        z[it] = 0.0;
    const int D = kp->D;
    for (int d = 0; d < D; ++d) {
        const double* xr = a.Xg + (int64_t)d * a.ldx;
        const double xi = xr[I], ie = kp->inv_ell[d];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const double q = (xi - xr[J0 + 2 * it]) * ie;
            z[it] = fma(q, q, z[it]);
        }
    }
    const int kind = kp->kind;
    const double sf2 = kp->sf2, da = kp->diag_add;
#pragma unroll
    for (int it = 0; it < 8; ++it)
        out[it] = kfun_fast_rt(kind, z[it], sf2) + (I == J0 + 2 * it ? da : 0.0);
}

// ---- k_tail's chain workgroup (round 5) ----------------------------------------------------------------------------------------
// Stamps (tools/kbench_t, profiles/r05_chain_stamps.log) showed that a hop of the chain is NOT "panel wave, then the crossing":
// three loops of about the same length go round at once —
//   (1) panel wave of block c-1 ends -> X22 visible -> phase B of block c's crossing -> first pivot -> panel wave of block c ends
//   (2) second half of L(c-1, c-2) published -> the LAST update step of workgroup c (it needs that tile) -> phase A -> phase B up to
//       the publication of L(c, c-1)'s second half
//   (3) X11 / L21 of block c-1 visible -> phase A -> phase B -> first pivot -> ... -> X11 / L21 of block c
// and every stage in them is tens of matrix-core instructions between barriers: two waves share a SIMD's matrix pipe, a
// 64 x 64 x 32 product is 0.85 us of it, the five products of a crossing 3 us.  So, here:
//  * products only over what is not zero and not thrown away: Y Y^T and L L^T feed the LOWER triangle of the diagonal block —
//    40 of its 64 units of 16 x 4, five per wave (syrk40) —, X11 and X22 are triangular (tri_solve32 stops at the diagonal and
//    pairs the waves of a SIMD so that their k ranges add up to the same);
//  * no layout conversions and no staging on the path: the diagonal block's lower triangle lives in the matrix-core accumulators
//    from the moment the workgroup starts (a2v = -tile) to the factorisation — every update of the loop and both halves of the
//    crossing add to that ONE chain, and what the factorisation reads is its negative, stored once (syrk40_store_neg); the tile
//    (c, c-1) sits in LDS in the layout the solve multiplies ([kk][i]) and the loop's sum is subtracted from it in place;
//  * the last update step half by half: BOTH of its tiles, L(c-1, c-2) and L(c, c-2), are published in two halves 3 us apart —
//    the k = 0..31 halves of both products run before the second halves arrive, 1.4 us of matrix-core time is left behind them;
//  * results go to a scratch block nobody is reading (no write-after-read barriers); the chain workgroup watches ITS OWN words of
//    the polled quarters and tiles without a pause (chain_watch; the last step's two tiles in ONE round trip: chain_watch2) —
//    one workgroup at a time is there, and the launch is waiting for it.
//    (Measured and dropped: diag_flow reading the block from one buffer and publishing L into another, which saves the barrier
//    behind its waves' first loads — the panel wave then runs 0.5 us longer per block, 6.97 against 6.44 us, 1.160 against 1.154 ms.)
// LDS (doubles), two halves of 10240 = the two operand pairs of the update loop; the LAST step uses pair 0:
//   pair 1: T [kk][i] (5120) | X11 (stride 34, 1088) | L21 (stride 34, 1088)
//   pair 0: opA | opB of the last step; behind barrier A1: D (64 x XS) | H | invd | sy | Xw (Ys = Y1 / Y2 lies inside Xw) | X22
#define CH_T 10240
#define CH_X11 (CH_T + NB * PS)
#define CH_LD (CH_X11 + 32 * 34)
#define CH_D 0
#define CH_AUX (NB * XS)
#define CH_YS (CH_AUX + 1096)
#define CH_X22 (CH_AUX + 1096 + DIAG_XW_DOUBLES)
static_assert(CH_X22 + 32 * 34 <= CH_T && CH_YS + 32 * PS <= CH_X22 && CH_LD + 32 * 34 <= 4 * NB * PS, "chain workgroup LDS carve");

// acc[u] += the wave's five 16 x 4 units of the lower triangle of Aop Aop^T over k in [ak0, ak0 + KLEN)  (Aop: [kk][i], stride PS)
//   waves 0..3: column block j1 = w (columns 4w ..), row blocks 0..3 -> u = 0..3;  column block 15 - w, row block 3 -> u = 4
//   waves 4..7: column block j1 = w, row blocks 1..3 -> u = 0..2;  column block 15 - w, row blocks 2, 3 -> u = 3, 4
template <int KLEN>
static __device__ __forceinline__ void syrk40(const double* __restrict__ Aop, int ak0, int wave, int lane, double (&acc)[5])
{
    const int r16 = lane & 15, kq = lane >> 4, c4 = lane & 3;
    const int j1 = wave, j2 = 15 - wave;
    if (wave < 4) {
#pragma unroll
        for (int ks = 0; ks < KLEN; ks += 4) {
            const double* row = Aop + (ak0 + ks + kq) * PS;
            const double a0 = row[r16], a1 = row[16 + r16], a2 = row[32 + r16], a3 = row[48 + r16];
            const double b1 = row[4 * j1 + c4], b2 = row[4 * j2 + c4];
            acc[0] = mfma4(a0, b1, acc[0]);
            acc[1] = mfma4(a1, b1, acc[1]);
            acc[2] = mfma4(a2, b1, acc[2]);
            acc[3] = mfma4(a3, b1, acc[3]);
            acc[4] = mfma4(a3, b2, acc[4]);
        }
    }
    else {
#pragma unroll
        for (int ks = 0; ks < KLEN; ks += 4) {
            const double* row = Aop + (ak0 + ks + kq) * PS;
            const double a1 = row[16 + r16], a2 = row[32 + r16], a3 = row[48 + r16];
            const double b1 = row[4 * j1 + c4], b2 = row[4 * j2 + c4];
            acc[0] = mfma4(a1, b1, acc[0]);
            acc[1] = mfma4(a2, b1, acc[1]);
            acc[2] = mfma4(a3, b1, acc[2]);
            acc[3] = mfma4(a2, b2, acc[3]);
            acc[4] = mfma4(a3, b2, acc[4]);
        }
    }
}
// The eight columns (of a 32-column half) a wave's triangular products compute: waves w and w + 4 share a SIMD (dev.h) and get
// 0 | 24 and 8 | 16 — their k ranges, hn + 8 each, add up to 40 on every SIMD
static __host__ __device__ __forceinline__ int tri_solve_cols(int wave)
{
    const int hq = wave >> 1;
    return hq == 0 ? 0 : (hq == 1 ? 8 : (hq == 2 ? 24 : 16));
}
// y[m][n] = sum_{k <= column} Aop[ak0 + k][wm + 16 m + ..] X[column][k] for the wave's columns hn + 4 n + ..: X (32 x 32, row-major,
// stride 34) is lower triangular, the k loop stops at the wave's last column
static __device__ __forceinline__ void tri_solve32(const double* __restrict__ Aop, int ak0, const double* __restrict__ X, int wm,
                                                   int hn, int lane, double (&y)[2][2])
{
    const int ar = wm + (lane & 15), bc = hn + (lane & 3), kq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 32; ks += 4) {
        if (ks >= hn + 8) // (wave-uniform)
            break;
        double af[2], bf[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
            af[m] = Aop[(ak0 + ks + kq) * PS + ar + 16 * m];
#pragma unroll
        for (int n = 0; n < 2; ++n)
            bf[n] = X[(bc + 4 * n) * 34 + ks + kq];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                y[m][n] = mfma4(af[m], bf[n], y[m][n]);
    }
}

// The chain workgroup watches ITS OWN words of a polled block (device scope, no pause) until none shows the pattern: one look at
// poll_one's word, a pause and a second fetch of the own words are 0.5 us between "visible" and "seen" — on the chain.
template <int NW, int STRIDE = 512> // word q of a thread lies STRIDE words behind word q - 1
static __device__ __forceinline__ void chain_watch(const unsigned long long* __restrict__ p, unsigned long long (&b)[NW], int spin_limit,
                                                   int* __restrict__ info)
{
    static_assert(NW == 2 || NW == 4, "written out: the words stay in registers");
    const unsigned long long SENT = ~0ull;
    int spins = 0;
    if constexpr (NW == 4) {
        while (b[0] == SENT || b[1] == SENT || b[2] == SENT || b[3] == SENT) {
            if (++spins > spin_limit) {
                info[2] = 1;
                break;
            }
            if (b[0] == SENT) b[0] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b[1] == SENT) b[1] = __hip_atomic_load(p + STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b[2] == SENT) b[2] = __hip_atomic_load(p + 2 * STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b[3] == SENT) b[3] = __hip_atomic_load(p + 3 * STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    else {
        while (b[0] == SENT || b[1] == SENT) {
            if (++spins > spin_limit) {
                info[2] = 1;
                break;
            }
            if (b[0] == SENT) b[0] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b[1] == SENT) b[1] = __hip_atomic_load(p + STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("" ::: "memory");
}
// ... two blocks at once (four words of each): one round trip covers both
static __device__ __forceinline__ void chain_watch2(const unsigned long long* __restrict__ pa, unsigned long long (&a)[4],
                                                    const unsigned long long* __restrict__ pb, unsigned long long (&b)[4], int spin_limit,
                                                    int* __restrict__ info)
{
    const unsigned long long SENT = ~0ull;
    constexpr int STRIDE = 8 * NB;
    int spins = 0;
    while (a[0] == SENT || a[1] == SENT || a[2] == SENT || a[3] == SENT || b[0] == SENT || b[1] == SENT || b[2] == SENT || b[3] == SENT) {
        if (++spins > spin_limit) {
            info[2] = 1;
            break;
        }
        if (a[0] == SENT) a[0] = __hip_atomic_load(pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a[1] == SENT) a[1] = __hip_atomic_load(pa + STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a[2] == SENT) a[2] = __hip_atomic_load(pa + 2 * STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a[3] == SENT) a[3] = __hip_atomic_load(pa + 3 * STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b[0] == SENT) b[0] = __hip_atomic_load(pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b[1] == SENT) b[1] = __hip_atomic_load(pb + STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b[2] == SENT) b[2] = __hip_atomic_load(pb + 2 * STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b[3] == SENT) b[3] = __hip_atomic_load(pb + 3 * STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("" ::: "memory");
}
// acc = -(the wave's units of a 64 x 64 block held row-major with stride XS) / the block's units = -acc  (syrk40's layout)
static __host__ __device__ __forceinline__ void syrk40_units(int wave, int q, int& i, int& j)
{
    if (wave < 4) {
        i = q < 4 ? q : 3;
        j = q < 4 ? wave : 15 - wave;
    }
    else {
        i = q < 3 ? q + 1 : q - 1;
        j = q < 3 ? wave : 15 - wave;
    }
}
static __device__ __forceinline__ void syrk40_load_neg(const double* __restrict__ Dl, int wave, int lane, double (&acc)[5])
{
    const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4), dcol = lane & 3;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        int i, j;
        syrk40_units(wave, q, i, j);
        acc[q] = -Dl[(16 * i + drow) * XS + 4 * j + dcol];
    }
}
static __device__ __forceinline__ void syrk40_store_neg(double* __restrict__ Dl, int wave, int lane, const double (&acc)[5])
{
    const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4), dcol = lane & 3;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        int i, j;
        syrk40_units(wave, q, i, j);
        Dl[(16 * i + drow) * XS + 4 * j + dcol] = -acc[q];
    }
}

// The chain workgroup of column c >= 1 from "tiles (c, c-1) and (c, c) loaded" (cl, cv: lane = row layout) to "the diagonal block is
// complete in lds + CH_D" (a barrier away from the factorisation); L(c, c-1) is left in lds + CH_T for the matrix.
static __device__ __forceinline__ void tail_chain_updates_and_crossing(const TailArgs& a, const P256& x, double* __restrict__ lds,
                                                                       const int c, const double (&cl)[8], const double (&cv)[8])
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 16; // a 64 x 64 product's wave tile
    const int hn = tri_solve_cols(wave);
    const int crow = wm + (lane & 31), ccol = wn + (lane >> 5);
    const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4), dcol = lane & 3;
    double* const T = lds + CH_T;
    double* const X11 = lds + CH_X11;
    double* const Ld = lds + CH_LD;
    double* const Dl = lds + CH_D;
    double* const Ys = lds + CH_YS;
    double* const X22 = lds + CH_X22;
    const double* Sq = a.SP + (int64_t)(c - 1) * 3072;
    double* pub = a.LP + (int64_t)tail_tile_id(a.nb, c, c - 1) * (NB * NB);
    const unsigned long long* Sp = reinterpret_cast<const unsigned long long*>(Sq) + threadIdx.x;
    // The diagonal block's lower triangle lives in the matrix-core accumulators from here to the factorisation: a2v = -(tile) now,
    // + every product of the loop and of the crossing, and -a2v is what the factorisation reads.  (Through LDS once, here, where
    // nothing is waiting: the tile was loaded / generated in the lane = row layout.)
    double a2v[5];
#pragma unroll
    for (int it = 0; it < 8; ++it)
        lds[crow * XS + ccol + 2 * it] = cv[it];
    __syncthreads();
    syrk40_load_neg(lds, wave, lane, a2v);
    __syncthreads(); // (the loop's first operands land in the same place)
    unsigned long long xb[4]; // X11: e, e + 512; L21: 1024 + e, 1024 + e + 512
    if (c > 1) {
        // Steps s < c-1: tile (c, c-1) -= L(c, s) L(c-1, s)^T, tile (c, c) -= L(c, s) L(c, s)^T.  Software-pipelined over two pairs of
        // operand buffers (the operands of step s+1 are on their way under the products of step s: one barrier a step); the pair
        // alternates so that the LAST step uses pair 0.
        PolledTile pa, pb;
        pa.issue(a.LP + (int64_t)tail_tile_id(a.nb, c, 0) * (NB * NB));
        pb.issue(a.LP + (int64_t)tail_tile_id(a.nb, c - 1, 0) * (NB * NB));
        double a2l[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll 1
        for (int s = 0; s < c - 2; ++s) {
            double* const opA = lds + ((c - 2 - s) & 1) * (2 * NB * PS);
            double* const opB = opA + NB * PS;
            pa.finish(a.LP + (int64_t)tail_tile_id(a.nb, c, s) * (NB * NB), x.spin_limit, x.info);
            pa.store(opA);
            pb.finish(a.LP + (int64_t)tail_tile_id(a.nb, c - 1, s) * (NB * NB), x.spin_limit, x.info);
            pb.store(opB);
            pa.issue(a.LP + (int64_t)tail_tile_id(a.nb, c, s + 1) * (NB * NB));
            pb.issue(a.LP + (int64_t)tail_tile_id(a.nb, c - 1, s + 1) * (NB * NB));
            __syncthreads(); // this step's operands are in LDS (and every wave is through with the pair of step s-1)
            mm64<false>(opA, opB, wm, wn, lane, a2l);
            syrk40<NB>(opA, 0, wave, lane, a2v);
        }
        // The last step, s = c-2.  BOTH of its tiles are late: L(c-1, c-2) is what the chain workgroup before this one has only just
        // solved, L(c, c-2) what the tile below it has — each published in two halves, columns 0..31 behind phase A of its solve,
        // 32..63 behind phase B some 3 us later.  Each half's products as it comes (k = 0..31, then 32..63: the order of the whole
        // product), this workgroup's own words watched without a pause: 1.4 us of matrix-core time behind the second halves
        // instead of 2.8.
        double* const opA = lds;
        double* const opB = lds + NB * PS;
        {
            const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
            const unsigned long long* ga = reinterpret_cast<const unsigned long long*>(a.LP + (int64_t)tail_tile_id(a.nb, c, c - 2) * (NB * NB)) + i + kk0 * NB;
            const unsigned long long* gb = reinterpret_cast<const unsigned long long*>(a.LP + (int64_t)tail_tile_id(a.nb, c - 1, c - 2) * (NB * NB)) + i + kk0 * NB;
            unsigned long long ha[4] = {pa.b[0], pa.b[1], pa.b[2], pa.b[3]}, hb[4] = {pb.b[0], pb.b[1], pb.b[2], pb.b[3]};
            chain_watch2(ga, ha, gb, hb, x.spin_limit, x.info);
            TTS2(x, true, 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                opA[(kk0 + 8 * q) * PS + i] = __longlong_as_double((long long)ha[q]);
                opB[(kk0 + 8 * q) * PS + i] = __longlong_as_double((long long)hb[q]);
            }
            __syncthreads(); // columns 0..31 of both tiles in LDS; every wave is through with pair 1
#pragma unroll
            for (int it = 0; it < 8; ++it)
                T[(ccol + 2 * it) * PS + crow] = cl[it];
            mmk<false, 32, 4>(opA, 0, opB, 0, wm, wn, lane, a2l);
            syrk40<32>(opA, 0, wave, lane, a2v);
            unsigned long long ka[4] = {pa.b[4], pa.b[5], pa.b[6], pa.b[7]}, kb[4] = {pb.b[4], pb.b[5], pb.b[6], pb.b[7]};
            chain_watch2(ga + 32 * NB, ka, gb + 32 * NB, kb, x.spin_limit, x.info);
            TTS2(x, true, 9);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                opA[(32 + kk0 + 8 * q) * PS + i] = __longlong_as_double((long long)ka[q]);
                opB[(32 + kk0 + 8 * q) * PS + i] = __longlong_as_double((long long)kb[q]);
            }
            __syncthreads(); // columns 32..63 (and T)
            TTS2(x, true, 10);
            mmk<false, 32, 4>(opA, 32, opB, 32, wm, wn, lane, a2l);
            syrk40<32>(opA, 32, wave, lane, a2v);
#pragma unroll
            for (int q = 0; q < 4; ++q) // first look at X11 / L21 of block c-1, device scope: in steady state they have just become
                                        // visible, and the load's way passes under the products' tail and the update of T below
                xb[q] = __hip_atomic_load(Sp + 512 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // T -= all the loop's products, in place: a lane's own elements
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                double* t = T + (wn + 4 * n + dcol) * PS + wm + 16 * m + drow;
                *t = *t - a2l[m][n];
            }
    }
    else {
#pragma unroll
        for (int it = 0; it < 8; ++it)
            T[(ccol + 2 * it) * PS + crow] = cl[it];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            xb[q] = __hip_atomic_load(Sp + 512 * q, __ATOMIC_RELAXED, POLL_FIRST_SCOPE);
    }
    TTS(c, 0);
    // ---- phase A: X11 and L21 of block c-1, polled value by value (diag_flow.h: DiagEarly) ----
    {
        chain_watch<4>(Sp, xb, x.spin_limit, x.info);
        TTS2(x, true, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = threadIdx.x + 512 * q; // X11: e = k + 32 c ; L21: e = c + 32 k
            X11[(e >> 5) * 34 + (e & 31)] = __longlong_as_double((long long)xb[q]);
            Ld[(e & 31) * 34 + (e >> 5)] = __longlong_as_double((long long)xb[2 + q]); // Ld[c][k] = L21[c][k]
        }
    }
    __syncthreads(); // (A1) T, X11, L21 in LDS; pair 0 is free
    diag_flow_init(reinterpret_cast<DiagSync*>(lds + CH_AUX + DIAG_LTB + NB));
    double y1[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    tri_solve32(T, 0, X11, wm, hn, lane, y1); // Y1 = T1 X11^T
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            Ys[(hn + 4 * n + dcol) * PS + wm + 16 * m + drow] = y1[m][n];
    __syncthreads(); // (A2) Y1 in Ys; every wave is through with T[:, 0:32]
    TTS2(x, true, 6);
    double u[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    mmk<true, 32, 2, 34>(Ys, 0, Ld, 0, wm, hn, lane, u); // Y1 L21^T
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            double* t = T + (32 + hn + 4 * n + dcol) * PS + wm + 16 * m + drow; // (a lane's own elements)
            *t = *t - u[m][n];
            T[(hn + 4 * n + dcol) * PS + wm + 16 * m + drow] = y1[m][n]; // the first half of L(c, c-1), for the matrix
        }
    { // columns 0..31 of L(c, c-1) are final: the polled copy starts its way now
        const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = kk0 + 8 * q;
            __hip_atomic_store(pub + i + NB * col, Ys[col * PS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    TTS2(x, true, 7);
    // first look at X22 (in steady state it arrives about now: the load's way passes under the product)
    unsigned long long xc[2];
    xc[0] = __hip_atomic_load(Sp + 2048, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    xc[1] = __hip_atomic_load(Sp + 2048 + 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    syrk40<32>(Ys, 0, wave, lane, a2v); // + Y1 Y1^T
    TTS2(x, true, 1);
    // ---- phase B: X22 ----
    {
        chain_watch<2>(Sp + 2048, xc, x.spin_limit, x.info);
        TTS2(x, true, 2);
        const int e0 = threadIdx.x, e1 = threadIdx.x + 512; // e = k + 32 c
        X22[(e0 >> 5) * 34 + (e0 & 31)] = __longlong_as_double((long long)xc[0]);
        X22[(e1 >> 5) * 34 + (e1 & 31)] = __longlong_as_double((long long)xc[1]);
    }
    __syncthreads(); // (B1) X22 in LDS, T[:, 32:64] complete, Ys free
    TTS2(x, true, 3);
    double y2[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    tri_solve32(T, 32, X22, wm, hn, lane, y2); // Y2 = T2 X22^T
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            Ys[(hn + 4 * n + dcol) * PS + wm + 16 * m + drow] = y2[m][n];
    __syncthreads(); // (B2) Y2 in Ys; every wave is through with T[:, 32:64]
    TTS2(x, true, 4);
    syrk40<32>(Ys, 0, wave, lane, a2v);    // + Y2 Y2^T
    syrk40_store_neg(Dl, wave, lane, a2v); // the diagonal block's lower triangle, where the factorisation reads it
    // (off the chain: the other 32 columns of L(c, c-1) for the tiles below — the next chain workgroup's last update step has
    // a few microseconds of slack — and for the matrix)
    {
        const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = kk0 + 8 * q;
            __hip_atomic_store(pub + i + NB * (32 + col), Ys[col * PS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            T[(32 + hn + 4 * n + dcol) * PS + wm + 16 * m + drow] = y2[m][n];
    TTS2(x, true, 5);
}

// The solve of any other tile (b, c), b > c + 1: L(b, c) = tile X_c^T in the half-block form, two phases, with the chain
// workgroup's means — triangular products (tri_solve32), results through a scratch block (no write-after-read barriers: four
// barriers instead of seven), the tile's second half updated by each lane in place.  `near` (b - c <= 3: the tiles whose L the
// chain's next workgroups wait for, loop (2) above) watch their own words of the block's quarters without a pause; the others —
// hundreds in a tall launch — keep the one-word look with a pause (poll_one: their polling is memory traffic for everybody).
// own: the tile with every earlier step applied (lane = row layout).  L(b, c) is left in lds + CH_T ([kk][i]) behind a barrier.
static __device__ __forceinline__ void tail_tile_solve(const P256& x, double* __restrict__ lds, const double (&own)[8],
                                                       const double* __restrict__ Sq, double* __restrict__ pub, const bool near)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 16;
    const int hn = tri_solve_cols(wave);
    const int crow = wm + (lane & 31), ccol = wn + (lane >> 5);
    const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4), dcol = lane & 3;
    double* const T = lds + CH_T;
    double* const X11 = lds + CH_X11;
    double* const Ld = lds + CH_LD;
    double* const Ys = lds + CH_YS;
    double* const X22 = lds + CH_X22;
    const unsigned long long SENT = ~0ull;
    const unsigned long long* Sp = reinterpret_cast<const unsigned long long*>(Sq) + threadIdx.x;
#pragma unroll
    for (int it = 0; it < 8; ++it)
        T[(ccol + 2 * it) * PS + crow] = own[it];
    // ---- phase A: X11 and L21 ----
    {
        unsigned long long xb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) // (first look: cacheable, see PolledTile)
            xb[q] = __hip_atomic_load(Sp + 512 * q, __ATOMIC_RELAXED, POLL_FIRST_SCOPE);
        if (near)
            chain_watch<4>(Sp, xb, x.spin_limit, x.info);
        else {
            int spins = 0;
            while (xb[0] == SENT || xb[1] == SENT || xb[2] == SENT || xb[3] == SENT) {
                if (++spins > x.spin_limit) {
                    x.info[2] = 1;
                    break;
                }
                poll_one(Sq + 1023, x.spin_limit, x.info);            // X11's last row
                poll_one(Sq + 1024 + 32 * 31, x.spin_limit, x.info);  // L21's last column
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (xb[q] == SENT)
                        xb[q] = __hip_atomic_load(Sp + 512 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = threadIdx.x + 512 * q; // X11: e = k + 32 c ; L21: e = c + 32 k
            X11[(e >> 5) * 34 + (e & 31)] = __longlong_as_double((long long)xb[q]);
            Ld[(e & 31) * 34 + (e >> 5)] = __longlong_as_double((long long)xb[2 + q]);
        }
    }
    __syncthreads(); // (A1)
    double y1[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    tri_solve32(T, 0, X11, wm, hn, lane, y1); // Y1 = T1 X11^T
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            Ys[(hn + 4 * n + dcol) * PS + wm + 16 * m + drow] = y1[m][n];
    __syncthreads(); // (A2) Y1 in Ys; every wave is through with T[:, 0:32]
    double u[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    mmk<true, 32, 2, 34>(Ys, 0, Ld, 0, wm, hn, lane, u); // Y1 L21^T
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            double* t = T + (32 + hn + 4 * n + dcol) * PS + wm + 16 * m + drow; // (a lane's own elements)
            *t = *t - u[m][n];
            T[(hn + 4 * n + dcol) * PS + wm + 16 * m + drow] = y1[m][n];
        }
    { // columns 0..31 of L(b, c) are final: the polled copy starts its way
        const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = kk0 + 8 * q;
            __hip_atomic_store(pub + i + NB * col, Ys[col * PS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- phase B: X22 ----
    {
        unsigned long long xc[2];
        xc[0] = __hip_atomic_load(Sp + 2048, __ATOMIC_RELAXED, POLL_FIRST_SCOPE);
        xc[1] = __hip_atomic_load(Sp + 2048 + 512, __ATOMIC_RELAXED, POLL_FIRST_SCOPE);
        if (near)
            chain_watch<2>(Sp + 2048, xc, x.spin_limit, x.info);
        else {
            int spins = 0;
            while (xc[0] == SENT || xc[1] == SENT) {
                if (++spins > x.spin_limit) {
                    x.info[2] = 1;
                    break;
                }
                poll_one(Sq + 2048 + 1023, x.spin_limit, x.info); // X22's last row
                if (xc[0] == SENT)
                    xc[0] = __hip_atomic_load(Sp + 2048, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (xc[1] == SENT)
                    xc[1] = __hip_atomic_load(Sp + 2048 + 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const int e0 = threadIdx.x, e1 = threadIdx.x + 512; // e = k + 32 c
        X22[(e0 >> 5) * 34 + (e0 & 31)] = __longlong_as_double((long long)xc[0]);
        X22[(e1 >> 5) * 34 + (e1 & 31)] = __longlong_as_double((long long)xc[1]);
    }
    __syncthreads(); // (B1) X22 in LDS, T[:, 32:64] complete, Ys free
    double y2[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    tri_solve32(T, 32, X22, wm, hn, lane, y2); // Y2 = T2 X22^T
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            Ys[(hn + 4 * n + dcol) * PS + wm + 16 * m + drow] = y2[m][n];
    __syncthreads(); // (B2) Y2 in Ys; every wave is through with T[:, 32:64]
    {
        const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = kk0 + 8 * q;
            __hip_atomic_store(pub + i + NB * (32 + col), Ys[col * PS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            T[(32 + hn + 4 * n + dcol) * PS + wm + 16 * m + drow] = y2[m][n];
    __syncthreads(); // L(b, c) complete in T
}

static __device__ __forceinline__ void tail_body(const TailArgs& a, const int wgid, double* __restrict__ lds,
                                                 const KParams* __restrict__ kp = nullptr)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 16;
    const int crow = wm + (lane & 31), ccol = wn + (lane >> 5);
    int c = 0, b;
    if (a.order) {
        b = __builtin_amdgcn_readfirstlane(a.order[2 * wgid]);
        c = __builtin_amdgcn_readfirstlane(a.order[2 * wgid + 1]);
    }
    else {
        int id = wgid, colh = a.nb;
        while (id >= colh) {
            id -= colh;
            ++c;
            --colh;
        }
        b = c + id;
    }
    const int slot = tail_tile_id(a.nb, b, c); // the tile's slot in the polled buffers (independent of the dispatch order)
    const bool mute = a.spin_limit < 0;
    P256 x;
    x.A = a.A;
    x.lda = a.lda;
    x.p0 = a.t0;
    x.R0 = a.t0 + (int64_t)NB * b;
    x.Xt = a.Xt;
    x.info = a.info;
    x.mute = mute;
    x.spin_limit = mute ? -a.spin_limit : a.spin_limit;
    x.nrows = b < a.nfull ? NB : a.rhs_rows;
    x.want_d = false;
    x.Bx = lds;
    x.T0 = lds + NB * XS;
    x.T1 = x.T0 + NB * PS;
    x.T2 = x.T1 + NB * PS;
    x.S22 = a.SP;
    x.HP = a.LP;
    double* const myslot = a.LP + (int64_t)slot * (NB * NB);
    { // the other pair of buffers, for the next launch: this tile's slot (and its diagonal block's quarters)
        unsigned long long* nx = reinterpret_cast<unsigned long long*>(a.LPn + (int64_t)slot * (NB * NB));
#pragma unroll
        for (int q = 0; q < 8; ++q)
            nx[threadIdx.x + 512 * q] = ~0ull;
        if (b == c) {
            unsigned long long* ns = reinterpret_cast<unsigned long long*>(a.SPn + (int64_t)c * 3072);
#pragma unroll
            for (int q = 0; q < 6; ++q)
                ns[threadIdx.x + 512 * q] = ~0ull;
        }
    }
    if (b == c + 1 && b < a.nt)
        return; // the sub-diagonal tile (c + 1, c) belongs to the workgroup of the diagonal tile of its row (below)
    // the tile, lane = row layout
    const int crc = crow < x.nrows ? crow : x.nrows - 1;
    double cv[8];
    if (kp) {
        tail_gen_tile(a, kp, x.R0 + crc, a.t0 + (int64_t)NB * c + ccol, cv);
        if (a.Al && b >= a.nfull && threadIdx.x < NB) // the right-hand-side strip: the sweep's sentinel for these columns
            for (int p = 0; p < a.P; ++p)
                reinterpret_cast<unsigned long long*>(a.Al)[a.t0 + (int64_t)NB * c + threadIdx.x + (int64_t)p * a.ldal] = ~0ull;
    }
    else {
#pragma unroll
        for (int it = 0; it < 8; ++it)
            cv[it] = a.A[x.R0 + crc + (a.t0 + (int64_t)NB * c + ccol + 2 * it) * a.lda];
    }
    if (b == c) {
        // ---- a diagonal tile's workgroup: the chain.  It also owns the tile to the left, (c, c-1): L(c, c-1) never has to
        // travel to reach the block it completes (k_panel256's factoring strip).  Steps s < c-1 update both tiles with the same
        // polled L(c, s); step c-1 is the crossing (tail_chain_updates_and_crossing).
        double* const Dl = lds + CH_D;
        double* const Ltb = lds + CH_AUX;
        double* const invd = Ltb + DIAG_LTB;
        DiagSync* const sy = reinterpret_cast<DiagSync*>(invd + NB);
        if (c > 0) {
            double cl[8]; // tile (c, c-1)
            if (kp)
                tail_gen_tile(a, kp, x.R0 + crow, a.t0 + (int64_t)NB * (c - 1) + ccol, cl);
            else {
#pragma unroll
                for (int it = 0; it < 8; ++it)
                    cl[it] = a.A[x.R0 + crow + (a.t0 + (int64_t)NB * (c - 1) + ccol + 2 * it) * a.lda];
            }
            tail_chain_updates_and_crossing(a, x, lds, c, cl, cv);
            TTS(c, 1);
        }
        else {
#pragma unroll
            for (int it = 0; it < 8; ++it)
                Dl[crow * XS + ccol + 2 * it] = cv[it];
            diag_flow_init(sy);
        }
        __syncthreads(); // the diagonal block is complete in Dl (and nobody reads the crossing's scratch any more)
        DiagEarly ea;
        ea.mute = mute;
        ea.S = a.SP + (int64_t)c * 3072;
        TTS(c, 2);
        diag_flow(Dl, Ltb, invd, sy, a.A + x.R0 + x.R0 * a.lda, a.lda, a.Xt + (int64_t)c * (NB * NB), a.info, x.R0, wave, lane,
                  invd + NB + 8, &ea);
        TTS(c, 3);
        if (c > 0) { // L(c, c-1) into the matrix: nobody reads it there before the launch ends
            __syncthreads();
            const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
            double* Ag = a.A + x.R0 + (a.t0 + (int64_t)NB * (c - 1)) * a.lda;
            const double* T = lds + CH_T;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int col = kk0 + 8 * q;
                Ag[i + (int64_t)col * a.lda] = T[col * PS + i];
            }
        }
        return;
    }
    // ---- any other tile: steps 0 .. c-1, then its solve ----
    PolledTile pa, pb;
    if (c > 0) {
        pa.issue(a.LP + (int64_t)tail_tile_id(a.nb, b, 0) * (NB * NB));
        pb.issue(a.LP + (int64_t)tail_tile_id(a.nb, c, 0) * (NB * NB));
    }
    double a2[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll 1
    for (int s = 0; s < c; ++s) { // (pipelined over two pairs of operand buffers, one barrier a step: see the diagonal workgroup's loop)
        double* const opA = lds + (s & 1) * (2 * NB * PS);
        double* const opB = opA + NB * PS;
        pa.finish(a.LP + (int64_t)tail_tile_id(a.nb, b, s) * (NB * NB), x.spin_limit, x.info);
        pa.store(opA);
        pb.finish(a.LP + (int64_t)tail_tile_id(a.nb, c, s) * (NB * NB), x.spin_limit, x.info);
        pb.store(opB);
        if (s + 1 < c) { // the next step's operands: on their way under this step's product
            pa.issue(a.LP + (int64_t)tail_tile_id(a.nb, b, s + 1) * (NB * NB));
            pb.issue(a.LP + (int64_t)tail_tile_id(a.nb, c, s + 1) * (NB * NB));
        }
        __syncthreads();
        mm64<false>(opA, opB, wm, wn, lane, a2);
    }
    if (c > 0) {
        double a2r[8];
        wave_tile_to_rows(a2, a2r, lane);
#pragma unroll
        for (int it = 0; it < 8; ++it)
            cv[it] -= a2r[it];
        __syncthreads(); // the operand buffers are free again
    }
    tail_tile_solve(x, lds, cv, a.SP + (int64_t)c * 3072, myslot, b - c <= 3);
    { // L(b, c) into the matrix
        const int i = threadIdx.x & 63, kk0 = threadIdx.x >> 6;
        double* Ag = a.A + x.R0 + (a.t0 + (int64_t)NB * c) * a.lda;
        const double* T = lds + CH_T;
        if (i < x.nrows) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int col = kk0 + 8 * q;
                Ag[i + (int64_t)col * a.lda] = T[col * PS + i];
            }
        }
    }
}

__global__ __launch_bounds__(512) void k_tail(TailArgs a)
{
    __shared__ __attribute__((aligned(16))) double lds[TAIL_LDS_DOUBLES]; // [Bx | T0 | T1 | T2] / [A0 | B0 | A1 | B1]: all 160 KB
    tail_body(a, (int)blockIdx.x, lds);
}
// the same generating its own tiles of K (a.Xg; the kernel parameters ride in the kernel arguments)
__global__ __launch_bounds__(512) void k_tail_g(TailArgs a, KParams kp)
{
    __shared__ __attribute__((aligned(16))) double lds[TAIL_LDS_DOUBLES];
    tail_body(a, (int)blockIdx.x, lds, &kp);
}
// G members at once: blockIdx.x = tile * G + member (the members' chains advance side by side; a wait is for a lower tile of
// the same member, i.e. a lower-numbered workgroup)
__global__ __launch_bounds__(512) void k_tail_b(TailArgs a, const BatchTab* __restrict__ bt)
{
    __shared__ __attribute__((aligned(16))) double lds[TAIL_LDS_DOUBLES];
    const int G = bt->G, gp = (int)blockIdx.x % G;
    a.A = bt_rebase(bt, gp, a.A);
    a.Xt = bt_rebase(bt, gp, a.Xt);
    a.info = bt_rebase(bt, gp, a.info);
    a.LP = bt_rebase(bt, gp, a.LP);
    a.SP = bt_rebase(bt, gp, a.SP);
    a.LPn = bt_rebase(bt, gp, a.LPn);
    a.SPn = bt_rebase(bt, gp, a.SPn);
    if (a.Xg) { // (every member's own samples, obs_mean and kernel parameters)
        a.Xg = bt_rebase(bt, gp, a.Xg);
        a.Om = bt_rebase(bt, gp, a.Om);
        a.Al = bt_rebase(bt, gp, a.Al);
    }
    tail_body(a, (int)blockIdx.x / G, lds, a.Xg ? &bt->kp[gp] : nullptr);
}

// ---- the update of a ragged order's last block, behind the data-flow launch ------------------------------------------------------
// C[0:m, 0:n] -= A[0:m, 0:k] A[0:n, 0:k]^T (lower part: i >= j) with m <= 64 rows (the ragged rows of the order + the
// right-hand-side rows), n < 64 columns and k = everything the data-flow launch factored, up to 2816: ONE tile.  Its k loop on one
// compute unit is bound by that unit's load rate — 46 us at k = 1088, 69 us at k = 1664 (profiles/r05_tail_sizes.log: N = 1700 took
// longer than N = 2048).  Here the k range is dealt to up to 32 workgroups, each leaves its partial product in a scratch slot, and
// a second launch adds the slots IN ORDER (bitwise reproducible) and subtracts the sum.
// The scratch is the pair of polled buffers the data-flow launch has just used: dead until the NEXT launch arms them again, all of
// them (tail_body: every workgroup arms its own slots of the other pair).
__global__ __launch_bounds__(256) void k_ragged_partial(const double* __restrict__ A, int64_t ld, int m, int n, int64_t k, int kc,
                                                        double* __restrict__ part)
{
    __shared__ __attribute__((aligned(16))) double As[32][NB];
    const int li = threadIdx.x & 63, lk = threadIdx.x >> 6; // loading: row li, k rows lk, lk + 4, ..
    const int ti = threadIdx.x & 15, tj = threadIdx.x >> 4; // computing: a 4 x 4 block, rows 4 ti .., columns 4 tj ..
    const int64_t k0 = (int64_t)blockIdx.x * kc, k1 = k0 + kc < k ? k0 + kc : k;
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            acc[r][c] = 0.0;
    double nx[8]; // the next 32 k rows, on their way under this block's products
    auto fetch = [&](int64_t kb) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { // (rows >= m read as zero; the B operand is the first n rows of the same strip)
            const int kk = lk + 4 * q;
            nx[q] = (kb + kk < k1 && li < m) ? A[li + (kb + kk) * ld] : 0.0;
        }
    };
    fetch(k0);
    for (int64_t kb = k0; kb < k1; kb += 32) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q)
            As[lk + 4 * q][li] = nx[q];
        if (kb + 32 < k1)
            fetch(kb + 32);
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = As[kk][4 * ti + r];
                b[r] = As[kk][4 * tj + r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[r][c] = fma(a[r], b[c], acc[r][c]);
        }
    }
    double* out = part + (int64_t)blockIdx.x * (NB * NB);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            out[4 * ti + r + NB * (4 * tj + c)] = acc[r][c];
}
#define RAGGED_MAX_G 32
__global__ __launch_bounds__(256) void k_ragged_fold(double* __restrict__ C, int64_t ldc, int m, int n, int G, const double* __restrict__ part)
{
    const int i = threadIdx.x & 63, j = 4 * (int)blockIdx.x + (threadIdx.x >> 6); // one element a thread, 16 workgroups
    if (i >= m || j >= n || i < j)
        return;
    // every slot's value requested before the first is added (written out to RAGGED_MAX_G: one after the other the loads cost a
    // trip to memory each — 62 us for 13 slots in one workgroup), then the sum in slot order
    double v[RAGGED_MAX_G];
#pragma unroll
    for (int g = 0; g < RAGGED_MAX_G; ++g)
        v[g] = g < G ? part[(int64_t)g * (NB * NB) + i + NB * j] : 0.0;
    double sum = 0.0;
#pragma unroll
    for (int g = 0; g < RAGGED_MAX_G; ++g)
        sum += v[g]; // (slots >= G add +0.0)
    C[i + (int64_t)j * ldc] -= sum;
}
// how the k range is dealt: G workgroups (0: not worth it / no room) of kc rows each, kc a multiple of the kernel's 32-row blocks,
// the last one not empty; also the test hook gpe_debug_ragged_split (host only)
int ragged_split(int64_t k, int64_t scratch_doubles, int* kc_out)
{
    if (k < 256 || scratch_doubles < 2 * NB * NB)
        return 0;
    int64_t G = k / 64;
    G = G > RAGGED_MAX_G ? RAGGED_MAX_G : G;
    G = G > scratch_doubles / (NB * NB) ? scratch_doubles / (NB * NB) : G;
    const int64_t kc = ((k + G - 1) / G + 31) / 32 * 32;
    G = (k + kc - 1) / kc;
    *kc_out = (int)kc;
    return (int)G;
}
// ... and the rest of a ragged order's last block in ONE more launch (round 6, later): the slots are added in order and subtracted,
// the block (jb < 64 columns, padded with the identity) is factored and inverted by diag_flow, the right-hand-side rows under it are
// solved with its inverse — what k_ragged_fold, k_diag_full and a k_gemm4 launch did one after the other with a launch gap each
// (4 + 21 + 5 us and three gaps at N = 1100: profiles/r06_ragged_orders.log).  One workgroup of 512 threads.
//   C = A[N64 .., N64 ..]: rows 0 .. jb-1 the block (lower triangle), rows jb .. jb+P-1 the right-hand-side rows; part: G slots of
//   64 x 64 (i + 64 j); Lscr: 64 x 64 doubles of scratch (diag_flow stores whole columns: into the matrix they would run over the
//   right-hand-side rows); Xt: the block's inverse, transposed, identity-padded (what the sweeps read).
__global__ __launch_bounds__(DIAG_THREADS) void k_ragged_finish(double* __restrict__ C, int64_t ldc, int jb, int P, int G,
                                                                const double* __restrict__ part, double* __restrict__ Lscr,
                                                                double* __restrict__ Xt, int* __restrict__ info, int64_t goff)
{
    __shared__ __attribute__((aligned(16))) double Ls[NB * XS];
    __shared__ __attribute__((aligned(16))) double Ltb[DIAG_LTB];
    __shared__ __attribute__((aligned(16))) double invd[NB];
    __shared__ DiagSync sy;
    __shared__ __attribute__((aligned(16))) double Xw[DIAG_XW_DOUBLES];
    __shared__ double Rr[NB * 65]; // the right-hand-side rows: Rr[p * 65 + k]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = jb + P;
    for (int e = threadIdx.x; e < NB * 65; e += DIAG_THREADS)
        Rr[e] = 0.0;
    for (int e = threadIdx.x; e < NB * NB; e += DIAG_THREADS) // the identity the short block is padded with; zeros above the diagonal
        Ls[(e & 63) * XS + (e >> 6)] = (e & 63) == (e >> 6) ? 1.0 : 0.0;
    __syncthreads();
    // the jb live columns only (64 jb elements, i + 64 j): every slot's value of TWO elements requested before the first is added,
    // then the sums in slot order (k_ragged_fold's)
#pragma unroll 1
    for (int e0 = 0; e0 < NB * jb; e0 += 2 * DIAG_THREADS) {
        double v[2][RAGGED_MAX_G], c0[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int e = e0 + u * DIAG_THREADS + (int)threadIdx.x;
            e = e < NB * jb ? e : NB * jb - 1;
            const int i = e & 63, j = e >> 6;
#pragma unroll
            for (int g = 0; g < RAGGED_MAX_G; ++g)
                v[u][g] = part[(int64_t)(g < G ? g : G - 1) * (NB * NB) + e];
            c0[u] = C[(i < m ? i : m - 1) + (int64_t)j * ldc];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = e0 + u * DIAG_THREADS + (int)threadIdx.x;
            const int i = e & 63, j = e >> 6;
            double sum = 0.0;
#pragma unroll
            for (int g = 0; g < RAGGED_MAX_G; ++g)
                sum += g < G ? v[u][g] : 0.0;
            const double val = c0[u] - sum;
            if (e < NB * jb && i < m && i >= j) {
                if (i < jb)
                    Ls[i * XS + j] = val;
                else
                    Rr[(i - jb) * 65 + j] = val;
            }
        }
    }
    diag_flow_init(&sy);
    __syncthreads();
    diag_flow(Ls, Ltb, invd, &sy, Lscr, NB, Xt, info, goff, w, lane, Xw);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's stores of L and X^T are acknowledged
    __syncthreads();
    for (int e = threadIdx.x; e < NB * NB; e += DIAG_THREADS) {
        const int i = e & 63, j = e >> 6;
        if (i < jb && j <= i)
            C[i + (int64_t)j * ldc] = __hip_atomic_load(Lscr + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // z[p][c] = sum_k r[p][k] X[c][k]   (X^T[k + 64 c], staged through LDS: Ls is free; zero above the diagonal)
    if (P > 0) {
        for (int e = threadIdx.x; e < NB * NB; e += DIAG_THREADS)
            Ls[(e >> 6) * XS + (e & 63)] = __hip_atomic_load(Xt + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // Ls[c][k]
        __syncthreads();
        for (int e = threadIdx.x; e < P * NB; e += DIAG_THREADS) {
            const int c = e & 63, p = e >> 6;
            if (c < jb) {
                double z = 0.0;
                for (int k = 0; k <= c; ++k)
                    z = fma(Rr[p * 65 + k], Ls[c * XS + k], z);
                C[jb + p + (int64_t)c * ldc] = z;
            }
        }
    }
}
bool launch_ragged_finish(hipStream_t s, double* C, int64_t ldc, const double* A, int64_t ld, int64_t jb, int64_t P, int64_t k,
                          double* scratch, int64_t scratch_doubles, double* Xt, int* info, int64_t goff)
{
    const int64_t m = jb + P;
    if (g_batch.bt || g_batch.G != 1 || jb < 1 || jb >= NB || P < 0 || m > NB || !scratch || scratch_doubles < 3 * NB * NB)
        return false;
    int kc = 0;
    const int G = ragged_split(k, scratch_doubles - NB * NB, &kc); // (one slot is diag_flow's scratch)
    if (G < 2)
        return false;
    GPE_LAUNCH(k_ragged_partial, dim3((unsigned)G), dim3(256), 0, s, A, ld, (int)m, (int)jb, k, kc, scratch);
    GPE_LAUNCH(k_ragged_finish, dim3(1), dim3(DIAG_THREADS), 0, s, C, ldc, (int)jb, (int)P, G, (const double*)scratch,
               scratch + (int64_t)G * (NB * NB), Xt, info, goff);
    return true;
}
// false: not this shape (the caller takes the general product)
bool launch_ragged_update(hipStream_t s, double* C, int64_t ldc, const double* A, int64_t ld, int64_t m, int64_t n, int64_t k,
                          double* scratch, int64_t scratch_doubles)
{
    if (g_batch.bt || g_batch.G != 1 || m < 1 || m > NB || n < 1 || n > NB || !scratch)
        return false;
    int kc = 0;
    const int G = ragged_split(k, scratch_doubles, &kc);
    if (G < 2)
        return false;
    GPE_LAUNCH(k_ragged_partial, dim3((unsigned)G), dim3(256), 0, s, A, ld, (int)m, (int)n, k, kc, scratch);
    GPE_LAUNCH(k_ragged_fold, dim3(NB / 4), dim3(256), 0, s, C, ldc, (int)m, (int)n, G, (const double*)scratch);
    return true;
}

// ---- dispatch order of a data-flow launch ----------------------------------------------------------------------------
// Workgroups are handed out in index order and each holds a CU from its dispatch to its last store, so WHEN a tile's
// workgroup becomes resident decides whether it spends its residency working or waiting — and 256 resident workgroups are all
// there is.  Column by column (rounds 3's order) a tall launch fills the chip with the 65 - c tiles of the next four columns,
// all waiting for their column's block inverse, while the diagonal workgroup of column c + 4 — 2 (c + 3) catch-up products
// of its own — is not even dispatched: from c ~ 12 on the chain waits for catch-up work (round-4 measurement: 18.9 us per
// column in the tall launch of N = 4096 against 13 in the closing launch); in a batch of G members every member has 256 / G
// resident workgroups, i.e. no look-ahead at all.  Any order is legal in which every wait is for a lower-numbered workgroup.
// Here: time slot tau per tile (in units of columns), sorted by (tau, column, row):
//   diagonal workgroup of column c (it also owns tile (c, c-1))          as early as its operands allow: behind slot c - 2
//   tile (b, c) of the triangle, b >= c + 2                              slot max(c, b - W): just in time for row b's
//                                                                        diagonal workgroup — by then its operands are there
//   tile (b, c) below the triangle (tall launch; the right-hand-side strip)   slot c + lag: behind the chain, operands ready
// W = 0 and lag = 0: column by column.  The table is checked against the dependencies before it is used.
#include <array>
#include <map>
#include <mutex>
#include <vector>
// the table itself, on the host: flat[2 w] = row strip, flat[2 w + 1] = tile column of workgroup w; false: a wait for a
// higher-numbered workgroup somewhere (the caller then falls back to the column-by-column order)
// Round 6: the chain workgroups come TAIL_DLEAD columns earlier than their operands allow.  Stamps of a closing launch
// (profiles/r06_closing_launch_stamps.log) show hops of 14-23 us at columns 19-28 where the chain workgroup's own catch-up
// products ("earlier updates done") end late: it was dispatched behind the tiles of column c - 2, when the 256 resident
// workgroups in front of it had retired, with 2 (c - 2) products still to do.  Dispatched four columns earlier it does most of
// them while the chain is still four columns away: N = 4096 1.127 -> 1.108 ms, 3072 0.724 -> 0.707, 2560 0.542 -> 0.532, <= 2048
// unchanged (profiles/r06_diag_lead.log; 2 ... 8 the same, 12 and 16 lose it again).  Such a workgroup may wait for a tile that
// is dispatched AFTER it — the one exception to "every wait is for a lower-numbered workgroup".  It is harmless as long as
// few of them can be in that state at once: at any point q of the dispatch order, the chain workgroups in front of q that
// wait for something at or behind q hold a CU each while everything else in front of q waits only for lower-numbered
// workgroups, i.e. makes progress on the other CUs; the table is accepted only if that number never exceeds TAIL_DLEAD_MAX_BLOCKED
// (build_tail_order checks it).  Batched launches (every member has 256 / G resident workgroups) keep the strict order.
#define TAIL_DLEAD 4
#define TAIL_DLEAD_MAX_BLOCKED 16
static bool build_tail_order(int nt, int nb, int W, int lag, std::vector<int>& flat, int dlead = 0)
{
    struct T {
        int key, c, b;
    };
    std::vector<T> ts;
    for (int c = 0; c < nt; ++c)
        for (int b = c; b < nb; ++b) {
            int k;
            if (b == c)
                k = 2 * c - 3 - 2 * dlead; // behind the tiles of slot c - 2 (its last operands: (c, c-2) and the diagonal workgroup c - 1)
            else if (b >= nt) // (a lag that shrinks along the launch — late columns' tiles started early for their catch-up
                              // products — measured slower at every slope: profiles/r04_lag_slope_negative.log)
                k = 2 * (c + std::max(lag, 0));
            else if (b == c + 1)
                k = 2 * c; // (owned by the diagonal workgroup of its row: this workgroup only arms its slot)
            else
                k = 2 * (W > 0 ? std::max(c, b - W) : c);
            ts.push_back(T{k, c, b});
        }
    std::stable_sort(ts.begin(), ts.end(), [](const T& x, const T& y) {
        if (x.key != y.key)
            return x.key < y.key;
        if (x.c != y.c)
            return x.c < y.c;
        return x.b < y.b;
    });
    auto tid = [&](int b, int c) { return c * nb - (c * (c - 1)) / 2 + (b - c); };
    std::vector<int> pos(ts.size());
    for (size_t i = 0; i < ts.size(); ++i)
        pos[tid(ts[i].b, ts[i].c)] = (int)i;
    // which workgroup factors diagonal block c / publishes the slot of tile (b, s)?
    auto dpos = [&](int c) { return pos[tid(c, c)]; };
    auto owner = [&](int b, int s) { return (b == s + 1 && b < nt) ? dpos(b) : pos[tid(b, s)]; };
    bool legal = true;
    std::vector<int> blocked(ts.size() + 1, 0); // difference array: chain workgroups in front of q waiting for something at / behind q
    for (const T& t : ts) {
        const int me = pos[tid(t.b, t.c)];
        if (t.b == t.c) {
            int last = -1; // the latest-dispatched workgroup this one waits for
            for (int s2 = 0; s2 < t.c - 1; ++s2)
                last = std::max(last, std::max(owner(t.c, s2), owner(t.c - 1, s2)));
            if (last > me) {
                if (dlead <= 0)
                    legal = false;
                ++blocked[(size_t)me + 1]; // counts at q = me + 1 .. last
                --blocked[(size_t)last + 1];
            }
            if (t.c > 0)
                legal = legal && dpos(t.c - 1) < me;
        }
        else if (!(t.b == t.c + 1 && t.b < nt)) {
            for (int s2 = 0; s2 < t.c && legal; ++s2)
                legal = owner(t.b, s2) < me && owner(t.c, s2) < me;
            legal = legal && dpos(t.c) < me;
        }
        if (!legal)
            break;
    }
    for (size_t q = 1, run = 0; q < blocked.size() && legal; ++q) {
        run += blocked[q];
        legal = (int)run <= TAIL_DLEAD_MAX_BLOCKED;
    }
    flat.assign(2 * ts.size(), 0);
    for (size_t i = 0; i < ts.size(); ++i) {
        flat[2 * i] = ts[i].b;
        flat[2 * i + 1] = ts[i].c;
    }
    return legal;
}
static const int* tail_order(int nt, int nb, int dlead, int lag)
{
    const int W = 0; // (the just-in-time window of the table: measured as a loss — kept in build_tail_order for the record)
    if (dlead <= 0 && lag <= 0)
        return nullptr;
    static std::mutex mu;
    static std::map<std::array<int, 6>, int*> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const std::array<int, 6> key{dev, nt, nb, dlead, lag, 0};
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end())
        return it->second;
    std::vector<int> flat;
    int* d = nullptr;
    if (build_tail_order(nt, nb, W, lag, flat, dlead) || (dlead > 0 && build_tail_order(nt, nb, W, lag, flat, 0))) {
        if (hipMalloc(&d, sizeof(int) * flat.size()) != hipSuccess || hipMemcpy(d, flat.data(), sizeof(int) * flat.size(), hipMemcpyHostToDevice) != hipSuccess)
            d = nullptr;
    }
    else
        fprintf(stderr, "gpe: tail_order(%d, %d, lead %d, lag %d) violates a dependency — column-by-column order used\n", nt, nb, dlead, lag);
    cache[key] = d;
    return d;
}
// test hook: the work split of the chain workgroup's products (syrk40 / tri_solve32) as the device code has it
void debug_chain_split(int wave, int* units10, int* cols)
{
    for (int q = 0; q < 5; ++q)
        syrk40_units(wave, q, units10[2 * q], units10[2 * q + 1]);
    *cols = tri_solve_cols(wave);
}
// test hook (include/gpe.h: gpe_debug_tail_order): 1 if the dispatch table of a data-flow launch of nt tile columns x nb row
// strips is a permutation of its tiles in which every wait is for a lower-numbered workgroup, 0 if not; host only
int debug_tail_order(int nt, int nb, int lag, int pair)
{
    if (pair)
        return -1; // (the two-blocks-per-chain-workgroup form of round 4 was removed in round 6)
    if (nt < 1 || nb < nt || nb > 4096)
        return -1;
    std::vector<int> flat;
    if (!build_tail_order(nt, nb, 0, lag, flat, 0)) // the strict table (batched launches)
        return 0;
    if (!build_tail_order(nt, nb, 0, lag, flat, TAIL_DLEAD)) // ... and the one single launches use (checked below)
        return 0;
    std::vector<char> seen((size_t)nt * nb, 0);
    size_t n = 0;
    for (size_t i = 0; i + 1 < flat.size(); i += 2) {
        const int b = flat[i], c = flat[i + 1];
        if (c < 0 || c >= nt || b < c || b >= nb || seen[(size_t)c * nb + b])
            return 0;
        seen[(size_t)c * nb + b] = 1;
        ++n;
    }
    return n == (size_t)(nt * nb - nt * (nt - 1) / 2) ? 1 : 0;
}

// Tile columns t0 .. t1-1 (whole 64-blocks) of the rows t0 .. M-1: N64 - t0 full row strips (N64 = the matrix order rounded down
// to 64; t1 == N64: the closing launch) and, as one more row strip, the M - N64 <= 64 rows below them — right-hand-side rows and
// the rows of a ragged last block the caller finishes.  Fully updated by everything in front of t0.  buf_cur / buf_next:
// tail_buf_doubles(nt, nb) each, all-ones (this launch arms buf_next)
void launch_tail(hipStream_t s, double* A, int64_t lda, int64_t t0, int64_t t1, int64_t N64, int64_t M, double* Xt_all, int* info,
                 double* buf_cur, double* buf_next, const TailGen* gen)
{
    TailArgs a{};
    a.A = A;
    a.lda = lda;
    a.t0 = t0;
    a.nt = (int)((t1 - t0) / NB);
    a.nfull = (int)((N64 - t0) / NB);
    a.rhs_rows = (int)(M - N64);
    a.nb = a.nfull + (a.rhs_rows > 0 ? 1 : 0);
    a.Xt = Xt_all + (t0 / NB) * (NB * NB);
    a.info = info;
    a.SP = buf_cur;
    a.LP = buf_cur + (int64_t)a.nt * 3072;
    a.SPn = buf_next;
    a.LPn = buf_next + (int64_t)a.nt * 3072;
    static const bool fault = getenv("GPE_HANDOVER_FAULT") && atoi(getenv("GPE_HANDOVER_FAULT")) != 0;
    a.spin_limit = fault ? -16 : GPE_FLOW_SPIN_LIMIT;
    // measured (profiles/r04_dispatch_order.log, N = 4096): lag 2..4 -> 794-800 evaluations/s against 735 column by column; a
    // just-in-time window W > 0 LOSES (6: 675, 8: 694, 12: 738, 16: 770): a tile dispatched late has its catch-up products still
    // to do when its row's diagonal workgroup asks for it; waiting workgroups are not what limits the closing launch
    static const int ord_lag = getenv("GPE_TAIL_LAG") ? atoi(getenv("GPE_TAIL_LAG")) : 3;
    a.order = tail_order(a.nt, a.nb, g_batch.bt ? 0 : TAIL_DLEAD, ord_lag);
    const int64_t tiles = tail_tiles(a.nt, a.nb);
    if (gen) {
        a.Xg = gen->Xg;
        a.ldx = gen->ldx;
        a.Ns = gen->Ns;
        a.Om = gen->Om;
        a.ldom = gen->ldom;
        a.Al = gen->Al;
        a.ldal = gen->ldal;
        a.P = gen->P;
    }
    FlowGate gate(s); // (one data-flow launch at a time on the device: dev.h)
    if (g_batch.bt)
        GPE_LAUNCH(k_tail_b, dim3((unsigned)(tiles * g_batch.G)), dim3(512), 0, s, a, g_batch.bt);
    else if (gen)
        GPE_LAUNCH(k_tail_g, dim3((unsigned)tiles), dim3(512), 0, s, a, *gen->kp);
    else
        GPE_LAUNCH(k_tail, dim3((unsigned)tiles), dim3(512), 0, s, a);
}

// ---------------------------------------------------------------------------------------------
// k_upd_fused — the next-panel update (rows >= pe of columns [pe, pe2), k = pe - p0) and, in the SAME
// launch, the factorisation of the next diagonal block.  The update is the 64 x 64 direct-to-LDS GEMM
// (gemm_glds64.h) on gridDim.x - 1 workgroups, which leave tile (0, 0) alone; the last workgroup forms
// that tile itself — A[pe:pe+64, pe:pe+64] - L_d L_d^T with L_d = A[pe:pe+64, p0:pe], (pe - p0) / 64
// products of 64^3 — and then factors and half-inverts it exactly like workgroup 0 of k_panel_step.
// k_diag used to follow the update as a launch of its own (13.6 us on the critical path of every
// outer panel, with 255 CUs idle); here it runs underneath the update (~18 us).
// (Round 3 also folded the panel's last 64-column step into this launch — UpdFold — for the step-by-step panels; the
// one-launch panels made it unreachable and round 4 removed it.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_upd_fused(GemmArgs g, double* __restrict__ A, int64_t lda, int64_t p0, int64_t pe,
                                                   double* __restrict__ Xt_next, int* __restrict__ info,
                                                   const double* __restrict__ Dacc)
{
    constexpr int GEMM_LDS = 4 * Glds64Shape<16>::STAGE, DIAG_LDS = 2 * NB * PS;
    constexpr int LDS_DOUBLES = GEMM_LDS > DIAG_LDS ? GEMM_LDS : DIAG_LDS;
    __shared__ __attribute__((aligned(16))) double lds[LDS_DOUBLES]; // the update's 4 operand stages / the diagonal workgroup's tiles
    if (blockIdx.x + 1 < gridDim.x) {
        gemm_glds64_body<16, 4, 8>(g, lds, (int)blockIdx.x, (int)gridDim.x - 1, true);
        return;
    }
    // ---- the diagonal workgroup ----
    static_assert(NB * XS + DIAG_LTB + NB + 8 + DIAG_XW_DOUBLES <= 2 * NB * PS,
                  "[Ls | Ltb | invd | sync | Xw] is carved out of the two operand tiles");
    double* T0 = lds;
    double* T1 = lds + NB * PS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 16;
    const int crow = wm + (lane & 31), ccol = wn + (lane >> 5);
    double c0v[8]; // the tile before the update, lane = row layout
#pragma unroll
    for (int it = 0; it < 8; ++it)
        c0v[it] = A[pe + crow + (pe + ccol + 2 * it) * lda];
    const int nkb = (int)((pe - p0) / NB); // 0: the panel steps already applied every piece (k_panel_step, dnext)
    TileRegs tl;
    if (nkb > 0)
        tl.load(A + pe + p0 * lda, lda, NB);
    double acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            acc[m][n] = 0.0;
#pragma unroll 1
    for (int c = 0; c < nkb; ++c) { // two tiles alternate: a wave that refills one has passed the barrier behind its last readers
        double* T = (c & 1) ? T1 : T0;
        tl.store(T);
        if (c + 1 < nkb)
            tl.load(A + pe + (p0 + (int64_t)NB * (c + 1)) * lda, lda, NB);
        __syncthreads();
        mm64<false>(T, T, wm, wn, lane, acc);
    }
    __syncthreads(); // the operand tiles are dead: re-carve
    double* Ls = lds;
    double* Ltb = Ls + NB * XS;
    double* invd = Ltb + DIAG_LTB;
    {
        double a2r[8];
        wave_tile_to_rows(acc, a2r, lane);
#pragma unroll
        for (int it = 0; it < 8; ++it) // Dacc: what the panel steps summed up (same thread <-> element mapping)
            Ls[crow * XS + ccol + 2 * it] = c0v[it] - a2r[it] - (Dacc ? Dacc[threadIdx.x + 512 * it] : 0.0);
    }
    {
        DiagSync* sy = reinterpret_cast<DiagSync*>(invd + NB);
        diag_flow_init(sy);
        __syncthreads();
        diag_flow(Ls, Ltb, invd, sy, A + pe + pe * lda, lda, Xt_next, info, pe, wave, lane, invd + NB + 8);
        return;
    }
}

// g: the next-panel update as for launch_gemm_sub (tri, 64-multiple shapes checked by the caller)
void launch_upd_fused(hipStream_t s, const GemmArgs& g0, double* A, int64_t lda, int64_t p0, int64_t pe, double* Xt_next,
                      int* info, const double* Dacc)
{
    constexpr int TM = 64, TN = 64;
    GemmArgs g = g0;
    const int tiles_m = (int)((g.m + TM - 1) / TM), tiles_n = (int)((g.n + TN - 1) / TN);
    int fold = 1;
    const int nsup = (tiles_n + 1) / 2;
    for (int sc = 0; sc < nsup; ++sc) { // live-tile enumeration of gemm.hip:launch_glds64
        const int t2 = tiles_n - 1 - sc;
        int len = tiles_m - first_live_tile<TM, TN>(g, sc);
        if (t2 != sc)
            len += tiles_m - first_live_tile<TM, TN>(g, t2);
        fold = len > fold ? len : fold;
    }
    g.fold_len = fold;
    g.total = nsup * fold;
    const dim3 grid((unsigned)g.total + 1), block(512);
    if (g.stop_event)
        GPE_LAUNCH_STOP("k_upd_fused", k_upd_fused, grid, block, 0, s, (hipEvent_t)g.stop_event, g, A, lda, p0, pe, Xt_next, info, Dacc);
    else
        GPE_LAUNCH(k_upd_fused, grid, block, 0, s, g, A, lda, p0, pe, Xt_next, info, Dacc);
}

#ifdef DIAG_TIMING
void dump_tail_timing(int nt)
{
    long long h[64][4];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tail_ts), sizeof(h));
    const long long t0 = h[0][2];
    printf("k_tail, the diagonal workgroups (us after the first one starts factoring): column | earlier updates done | its left tile solved, block complete | factoring | panel wave done\n");
    for (int c = 0; c < nt && c < 64; ++c)
        printf("  %2d | %7.2f | %7.2f | %7.2f | %7.2f   (step %5.2f)\n", c, c ? (h[c][0] - t0) * 0.01 : 0.0, c ? (h[c][1] - t0) * 0.01 : 0.0,
               (h[c][2] - t0) * 0.01, (h[c][3] - t0) * 0.01, c ? (h[c][2] - h[c - 1][2]) * 0.01 : 0.0);
    long long cy[64][4];
    hipMemcpyFromSymbol(cy, HIP_SYMBOL(g_tail_cyc), sizeof(cy));
    printf("  factoring -> panel wave done, per column: us | shader-clock cycles | MHz\n   ");
    for (int c = 0; c < nt && c < 64; ++c)
        printf(" %d: %.2f %lld %.0f |", c, (h[c][3] - h[c][2]) * 0.01, cy[c][3] - cy[c][2], (cy[c][3] - cy[c][2]) / ((h[c][3] - h[c][2]) * 0.01));
    printf("\n");
    long long g[64][12];
    hipMemcpyFromSymbol(g, HIP_SYMBOL(g_tail_ts2), sizeof(g));
    printf("  between two blocks, us after the panel wave of column c-1 is through: X11/L21 seen | phase A done | X22 seen | X22 in LDS | Y2 written | second half product done | block complete | factoring\n");
    for (int c = 1; c < nt && c < 64; ++c) {
        const long long p = h[c - 1][3];
        printf("  %2d | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f   (updates done %6.2f; inside phase A: Y1 in LDS %6.2f, T2 updated + first half out %6.2f)\n", c, (g[c][0] - p) * 0.01, (g[c][1] - p) * 0.01, (g[c][2] - p) * 0.01,
               (g[c][3] - p) * 0.01, (g[c][4] - p) * 0.01, (g[c][5] - p) * 0.01, (h[c][1] - p) * 0.01, (h[c][2] - p) * 0.01, (h[c][0] - p) * 0.01,
               (g[c][6] - p) * 0.01, (g[c][7] - p) * 0.01);
        if (c >= 2)
            printf("       the last update step: first halves of L(c, c-2), L(c-1, c-2) seen %6.2f | second halves seen %6.2f | in LDS %6.2f\n",
                   (g[c][8] - p) * 0.01, (g[c][9] - p) * 0.01, (g[c][10] - p) * 0.01);
    }
}
void dump_p256_timing()
{
    long long h[5][32];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_p256_ts), sizeof(h));
    const char* names[5] = {"strip 0", "strip 1", "strip 2", "strip 3", "last strip"};
    long long t0 = h[0][30];
    for (int r = 0; r < 5; ++r)
        t0 = h[r][30] < t0 ? h[r][30] : t0;
    // wall_clock64 (s_memrealtime): the 100 MHz constant clock, the same on every CU -> 10 ns units
    printf("k_panel256 stamps (us after the first strip's start; per step: enter | X flag seen | X+tile in LDS | solved | L out/published | updates done)\n");
    for (int r = 0; r < 5; ++r) {
        printf("  %-10s start %6.2f :", names[r], (h[r][30] - t0) * 0.01);
        const int smax = r < 3 ? r : 3;
        for (int S = 0; S <= smax; ++S) {
            printf(" [S%d", S);
            for (int i = 0; i < 6; ++i)
                if (!(S == 0 && i == 1))
                    printf(" %6.2f", (h[r][6 * S + i] - t0) * 0.01);
            printf("]");
        }
        if (r < 3)
            printf(" diag start %6.2f wave 0 done %6.2f all acked %6.2f X out %6.2f", (h[r][26] - t0) * 0.01, (h[r][27] - t0) * 0.01,
                   (h[r][28] - t0) * 0.01, (h[r][29] - t0) * 0.01);
        printf(" end %6.2f\n", (h[r][31] - t0) * 0.01);
    }
}
void dump_panel_timing()
{
    long long h[64];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_panel_ts), sizeof(h));
    printf("k_panel_step WG0 cycles: loads %lld | trsm %lld | writeL %lld | updates %lld | to-diag %lld | rounds %lld | tail %lld | total %lld\n",
           h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[7] - h[0]);
    printf("k_panel_step last WG cycles: loads %lld | trsm %lld | writeL %lld | updates %lld | total %lld\n", h[33] - h[32],
           h[34] - h[33], h[35] - h[34], h[36] - h[35], h[36] - h[32]);
    printf("  its first update (wave 0): wait + fetch head tiles %lld | tile -> LDS + barrier %lld | 64^3 product %lld | to row layout %lld | C -= , store %lld | barrier %lld\n",
           h[42] - h[35], h[43] - h[42], h[44] - h[43], h[45] - h[44], h[46] - h[45], h[47] - h[46]);
}
#endif
void launch_panel_step(hipStream_t s, double* A, int64_t lda, int64_t j0, int64_t M, int nt, const double* Xt_cur,
                       double* Xt_next, int do_next, int* info, double* Hs, int64_t dnext, int64_t dfirst, int dinit,
                       double* Dacc, gpe_epoch_t* hflag)
{
    // a value no earlier launch of this process has used (0 is what fresh flag words hold)
    const gpe_epoch_t epoch = ++g_handover_epoch;
    // test hook: the consumers wait (briefly) for a value nobody writes, i.e. every hand-over of the launch "times out"
    static const bool fault = getenv("GPE_HANDOVER_FAULT") && atoi(getenv("GPE_HANDOVER_FAULT")) != 0;
    const int spin_limit = fault ? -16 : GPE_FLOW_SPIN_LIMIT;
    const int64_t rows = M - (j0 + NB);
    if (rows <= 0)
        return;
    // (No FlowGate here: the consumers of this launch wait for its FIRST workgroups only, a batch of 64 members runs two
    // sub-batches of these steps on two streams on purpose — one's panel steps under the other's updates, 8.9 k against 7.9 k
    // evaluations/s with the steps ordered — and two rounds of that have not seen a lost hand-over; the polls are bounded.)
    if (g_batch.bt)
        GPE_LAUNCH(k_panel_step_b, dim3((unsigned)((rows + NB - 1) / NB) * g_batch.G), dim3(512), 0, s, A, lda, j0, M, nt,
                           Xt_cur, Xt_next, do_next, info, Hs, dnext, dfirst, dinit, Dacc, hflag, epoch, spin_limit, g_batch.bt);
    else
        GPE_LAUNCH(k_panel_step, dim3((unsigned)((rows + NB - 1) / NB)), dim3(512), 0, s, A, lda, j0, M, nt, Xt_cur,
                           Xt_next, do_next, info, Hs, dnext, dfirst, dinit, Dacc, hflag, epoch, spin_limit);
}

// head tiles of the fused steps of one outer panel -> their place in A.  Step f (f = 0..nf-1) of the
// panel starting at column p0 left nt0 - f tiles: tile t = rows p0 + 64 (f + 1 + t), columns p0 + 64 f.
__global__ __launch_bounds__(256) void k_head_copy(double* __restrict__ A, int64_t lda, int64_t p0, int nt0,
                                                   const double* __restrict__ H, const BatchTab* __restrict__ bt)
{
    BT_REBASE(bt, A);
    BT_REBASE(bt, H);
    int f = 0, t = blockIdx.x;
    while (t >= nt0 - f) {
        t -= nt0 - f;
        ++f;
    }
    const double* src = H + (int64_t)blockIdx.x * (NB * NB);
    double* dst = A + (p0 + NB * (int64_t)(f + 1 + t)) + (p0 + NB * (int64_t)f) * lda;
    for (int e = threadIdx.x; e < NB * NB; e += 256)
        dst[(e & 63) + (int64_t)(e >> 6) * lda] = src[e];
}
void launch_head_copy(hipStream_t s, double* A, int64_t lda, int64_t p0, int nt0, int nf, const double* H)
{
    int tiles = 0;
    for (int f = 0; f < nf; ++f)
        tiles += nt0 - f;
    if (tiles > 0)
        GPE_LAUNCH(k_head_copy, dim3((unsigned)tiles, 1, g_batch.G), dim3(256), 0, s, A, lda, p0, nt0, H, g_batch.bt);
}

// inverses of the diagonal blocks of an existing factor: block b at L[64 b, 64 b]
__global__ __launch_bounds__(256) void k_diag_inv(const double* __restrict__ L, int64_t ldl, int64_t N, int64_t b0,
                                                  double* __restrict__ Xt_all)
{
    __shared__ __attribute__((aligned(16))) double Ls[NB * XS];
    __shared__ __attribute__((aligned(16))) double Xs[NB * XS];
    __shared__ __attribute__((aligned(16))) double Ts[NB * XS];
    __shared__ double invd[NB];
    const int64_t b = b0 + blockIdx.x;
    const int64_t j0 = b * NB;
    const int jb = (int)((N - j0 < NB) ? N - j0 : NB);
    const double* L11 = L + j0 + j0 * ldl;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e & 63, c = e >> 6;
        double v = 0.0;
        if (r < jb && c < jb)
            v = (c <= r) ? L11[r + (int64_t)c * ldl] : 0.0;
        else if (r == c)
            v = 1.0;
        Ls[r * XS + c] = v;
        if (r == c)
            invd[r] = 1.0 / v;
    }
    __syncthreads();
    invert_L64(Ls, invd, Xs, Ts);
    store_Xt(Xs, Xt_all + b * (NB * NB));
}

void launch_diag_inv(hipStream_t s, const double* L, int64_t ldl, int64_t N, int64_t b0, int64_t nblocks,
                     double* Xt_all)
{
    if (nblocks <= 0)
        return;
    GPE_LAUNCH(k_diag_inv, dim3((unsigned)nblocks), dim3(256), 0, s, L, ldl, N, b0, Xt_all);
}
