// sweep2.hip — the backward sweep  a = L^-T y  (GP::_compute_alpha, src/limbo/model/gp.hpp:605-611, second solve) for ONE
// right-hand side with the hop's arithmetic taken off the chain (round 6).
//
// k_trsv_bwd_flow (solve.hip, rounds 1-5): workgroup j folds in L[t, j]^T a_t as the a_t appear and then forms a_j = X_j^T w_j.
// Of its 1.7 us a hop at N = 4096 (profiles/r04_sweep_stamps.log) 1.1 are "a_{j+1} stored -> seen -> staged through LDS ->
// folded" and 0.6 the product with X_j: fold and solve both wait for a_{j+1}.  They need not:
//     a_j = g_j - M2_j a_{j+2} - M_j a_{j+1},     M_j = X_j^T L[j+1, j]^T,  M2_j = X_j^T L[j+2, j]^T  (64 x 64 each),
//     g_j = X_j^T ( y_j - sum_{t >= j+3} L[t, j]^T a_t )
// M_j and M2_j are two 64^3 products on the matrix cores when the launch starts (every workgroup is resident from the start and
// has nothing to do until the chain comes near), g_j is complete two hops before a_{j+1} exists, h_j = g_j - M2_j a_{j+2} one hop
// before, and what is left behind the arrival of a_{j+1} is one 64 x 64 matrix-vector product from LDS: read, multiply, sum over
// the eight waves, store.  (With M_j alone — a_j = g_j - M_j a_{j+1}, g_j over t >= j+2 — the hop went 1.7 -> 1.5 us: g_j itself
// then waits for a_{j+2} one hop back, fold and X_j^T and all, and is ready only 0.3 us before a_{j+1}; profiles/r06_sweep_m_stamps.log.)
// Same hand-off as k_trsv_bwd_flow (the output vector pre-filled with an all-ones pattern, every value one 8-byte agent-scope
// store, consumers poll the value), same block -> workgroup mapping (flow_block_of: consecutive blocks of the chain on one XCD,
// every wait is for a workgroup dispatched earlier), same bounded polls and the same answer to a timeout (*err, the host re-runs
// block by block).  Single-GP launches with one right-hand side; everything else keeps k_trsv_bwd_flow.
// (A first design — one chain workgroup walking all blocks, fed f_j / M_j / M2_j by helper workgroups — was built and measured no
// faster, 113 us against 111: two hand-offs per three hops instead of one per hop; profiles/r06_sweep_chain_negative.log.)
#include "gemm_glds64.h"

#define NB 64
#define S2_PS 80 // stride of the [kk][i] operand tiles in LDS (== 16 mod 32, as potrf.hip's PS)
#define S2_MS 65 // stride of M in LDS: Mt[k][c] at k * 65 + c

#ifdef S2_TIMING
__device__ long long g_s2_ts[256][6]; // per block: start | M done | folds done (g_j ready) | a_{j+1} seen | published
#define S2TS(i) do { if (threadIdx.x == 0) g_s2_ts[j][i] = wall_clock64(); } while (0)
#else
#define S2TS(i) do { } while (0)
#endif

__global__ __launch_bounds__(512) void k_trsv_bwd_m(const double* __restrict__ L, int64_t ld, int64_t N, const double* __restrict__ Xt_all,
                                                    const double* __restrict__ y, int64_t ysi, double* a, int* __restrict__ err,
                                                    const double* __restrict__ om, double* __restrict__ part)
{
    __shared__ __attribute__((aligned(16))) double lds[2 * NB * S2_PS]; // operand tiles of the product, then the fold's staging
    __shared__ double Mt[NB * S2_MS], M2t[NB * S2_MS];
    __shared__ double xs[NB], wj[NB], gj[NB], part_s[8 * NB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int64_t nblk = (N + NB - 1) / NB;
    const int64_t j = flow_block_of(nblk, true); // dependencies only on lower blockIdx.x, consecutive blocks on one XCD (dev.h)
    if (j < 0)
        return;
    const int64_t j0 = j * NB;
    const int jb = (int)((N - j0 < NB) ? N - j0 : NB);
    const double* Xt = Xt_all + j * (NB * NB); // Xt[c + 64 r] = X[r][c],  X = L_jj^-1
    const unsigned long long SENT = ~0ull;
    S2TS(0);
    // ---- M_j[c][k] = sum_r X[r][c] L[t0 + k][j0 + r] for t = j + 1, M2_j the same for t = j + 2 ----
    if (j + 1 < nblk) {
        double* As = lds;              // [r][c]
        double* Bs = lds + NB * S2_PS; // [r][k]
        const int wm = (wvu & 1) * 32, wn = (wvu >> 1) * 16;
        const int ar = wm + (lane & 15), bc = wn + (lane & 3), kq = lane >> 4;
        const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4), dcol = lane & 3;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = wvu + 8 * q;
            As[r * S2_PS + lane] = Xt[lane + NB * r];
        }
        for (int q2 = 1; q2 <= 2 && j + q2 < nblk; ++q2) {
            const int64_t t0 = (j + q2) * NB;
            const int tb = (int)((N - t0 < NB) ? N - t0 : NB);
            const int kc = lane < tb ? lane : tb - 1;
            const double km = lane < tb ? 1.0 : 0.0;
            if (q2 == 2)
                __syncthreads(); // (the first product's readers of Bs are through)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = wvu + 8 * q;
                const int rc = r < jb ? r : jb - 1;
                Bs[r * S2_PS + lane] = L[t0 + kc + (j0 + rc) * ld] * (r < jb ? km : 0.0);
            }
            __syncthreads();
            double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
            for (int ks = 0; ks < NB; ks += 4) {
                double af[2], bf[4];
#pragma unroll
                for (int x = 0; x < 2; ++x)
                    af[x] = As[(ks + kq) * S2_PS + ar + 16 * x];
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    bf[x] = Bs[(ks + kq) * S2_PS + bc + 4 * x];
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        acc[m][n] = mfma4(af[m], bf[n], acc[m][n]);
            }
            double* Md = q2 == 1 ? Mt : M2t;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    Md[(wn + 4 * n + dcol) * S2_MS + wm + 16 * m + drow] = acc[m][n]; // [k][c]
        }
    }
    S2TS(1);
    // ---- w_j = y_j - sum_{t = nblk-1 .. j+2} L[t, j]^T a_t  (k_trsv_bwd_flow's fold, four tiles in flight) ----
    double* Stg = lds; // [c][k], stride 65
    __syncthreads();   // (the product's operand tiles are consumed, Mt is written)
    if (threadIdx.x < NB)
        wj[lane] = (lane < jb) ? y[(j0 + lane) * ysi] : 0.0;
    double tl[4][8];
    unsigned long long pb[4] = {SENT, SENT, SENT, SENT};
    auto fetch = [&](double (&dst)[8], unsigned long long& peek, int64_t tt) {
        const int64_t t0 = tt * NB;
        const int tb = (int)((N - t0 < NB) ? N - t0 : NB);
        const int kc = lane < tb ? lane : tb - 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) { // (unconditional loads from clamped addresses; the mask is applied where the tile is used — fold —: with
            // the multiplication here the compiler issued load, wait, multiply, load, .. one at a time in front of the loop, and the loop's
            // own fetches came out worse too: 98.6 -> 90.5 us, profiles/r06_sweep_m_stamps.log)
            const int c = wvu + 8 * q;
            const int cc = c < jb ? c : jb - 1;
            dst[q] = L[t0 + kc + (j0 + cc) * ld];
        }
        peek = __hip_atomic_load((const unsigned long long*)(a + t0 + kc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // the values of block t: from the first look, or polled — three looks that this XCD's L2 may serve for every one that goes to
    // memory (the producer stores with agent scope: always correct, quicker when producer and consumer share an XCD)
    auto await = [&](unsigned long long peek, int64_t t) {
        double v = 0.0;
        const int64_t t0 = t * NB;
        const int tb = (int)((N - t0 < NB) ? N - t0 : NB);
        if (lane < tb) {
            unsigned long long bits = peek;
            int spins = 0;
            while (bits == SENT) {
                if ((spins & 3) != 3)
                    bits = __hip_atomic_load((const unsigned long long*)(a + t0 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else
                    bits = __hip_atomic_load((const unsigned long long*)(a + t0 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (bits != SENT)
                    break;
                if (++spins > GPE_FLOW_SPIN_LIMIT) {
                    *err = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            v = __longlong_as_double((long long)bits);
        }
        return v;
    };
    auto fold = [&](const double (&src)[8], unsigned long long peek, int64_t t) {
        __syncthreads();
        const int tbf = (int)((N - t * NB < NB) ? N - t * NB : NB);
        const double rowmask = lane < tbf ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            Stg[(wv + 8 * q) * 65 + lane] = src[q] * (wvu + 8 * q < jb ? rowmask : 0.0);
        if (threadIdx.x < NB)
            xs[lane] = await(peek, t);
        __syncthreads();
        double acc = 0.0;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = 8 * wv + kk;
            acc = fma(Stg[lane * 65 + k], xs[k], acc);
        }
        part_s[wv * NB + lane] = acc;
        __syncthreads();
        if (threadIdx.x < NB) {
            double s = part_s[lane];
#pragma unroll
            for (int w = 1; w < 8; ++w)
                s += part_s[w * NB + lane];
            wj[lane] -= s;
        }
    };
    const int64_t lo = j + 2; // contributors t > lo: blocks j + 1 and j + 2 arrive through M_j and M2_j
    int64_t t = nblk - 1;
    auto clampt = [&](int64_t tt) { return tt > lo ? tt : (lo + 1 < nblk ? lo + 1 : nblk - 1); };
    if (t > lo) {
        fetch(tl[0], pb[0], clampt(t));
        fetch(tl[1], pb[1], clampt(t - 1));
        fetch(tl[2], pb[2], clampt(t - 2));
        fetch(tl[3], pb[3], clampt(t - 3));
    }
    for (; t - 3 > lo; t -= 4) {
        fold(tl[0], pb[0], t);
        fetch(tl[0], pb[0], clampt(t - 4));
        fold(tl[1], pb[1], t - 1);
        fetch(tl[1], pb[1], clampt(t - 5));
        fold(tl[2], pb[2], t - 2);
        fetch(tl[2], pb[2], clampt(t - 6));
        fold(tl[3], pb[3], t - 3);
        fetch(tl[3], pb[3], clampt(t - 7));
    }
    if (t > lo)
        fold(tl[0], pb[0], t);
    if (t - 1 > lo)
        fold(tl[1], pb[1], t - 1);
    if (t - 2 > lo)
        fold(tl[2], pb[2], t - 2);
    // the first looks at a_{j+2} and a_{j+1}, on their way under the product below
    unsigned long long p2 = SENT, p1 = SENT;
    if (threadIdx.x < NB) {
        if (j + 2 < nblk) {
            const int64_t t0 = (j + 2) * NB;
            const int tb = (int)((N - t0 < NB) ? N - t0 : NB);
            p2 = __hip_atomic_load((const unsigned long long*)(a + t0 + (lane < tb ? lane : tb - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (j + 1 < nblk) {
            const int64_t t0 = (j + 1) * NB;
            const int tb = (int)((N - t0 < NB) ? N - t0 : NB);
            p1 = __hip_atomic_load((const unsigned long long*)(a + t0 + (lane < tb ? lane : tb - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    // ---- g_j = X_j^T w_j ----
    {
        double acc = 0.0;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int r = 8 * wv + kk;
            acc = fma(Xt[lane + NB * r], wj[r], acc);
        }
        part_s[wv * NB + lane] = acc;
        __syncthreads();
        if (threadIdx.x < NB) {
            double s = part_s[lane];
#pragma unroll
            for (int w = 1; w < 8; ++w)
                s += part_s[w * NB + lane];
            gj[lane] = s;
        }
    }
    S2TS(2);
    // one step of the recurrence: gj <- gj - Mx[.][.] a_t, a_t awaited here; returns the new value in wave 0's lanes
    auto step = [&](const double* Mx, unsigned long long peek, int64_t tt) {
        if (threadIdx.x < NB)
            xs[lane] = await(peek, tt);
        __syncthreads(); // a_t in LDS (and gj written, and the previous readers of part_s are through)
        double acc = 0.0;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = 8 * wv + kk;
            acc = fma(Mx[k * S2_MS + lane], xs[k], acc);
        }
        part_s[wv * NB + lane] = acc;
        __syncthreads();
        double r = 0.0;
        if (threadIdx.x < NB) {
            double s = part_s[lane];
#pragma unroll
            for (int w = 1; w < 8; ++w)
                s += part_s[w * NB + lane];
            r = gj[lane] - s;
            gj[lane] = r;
        }
        return r;
    };
    // ---- h_j = g_j - M2_j a_{j+2}  (a hop early), then the hop itself: a_j = h_j - M_j a_{j+1} ----
    double aj = 0.0;
    if (j + 2 < nblk)
        (void)step(M2t, p2, j + 2);
    S2TS(3);
    if (j + 1 < nblk)
        aj = step(Mt, p1, j + 1);
    else {
        __syncthreads();
        if (threadIdx.x < NB)
            aj = gj[lane];
    }
    if (threadIdx.x < NB) {
        if (lane < jb)
            __hip_atomic_store((unsigned long long*)(a + j0 + lane), (unsigned long long)__double_as_longlong(aj), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        S2TS(4);
        if (part) { // gp.hpp:274-277: sum_i log L_ii and sum_i om_i a_i over the block — wave 0, fixed-order butterflies
            double ldp = lane < jb ? log(L[(j0 + lane) + (j0 + lane) * ld]) : 0.0;
            double oa = (om && lane < jb) ? om[j0 + lane] * aj : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                ldp += __shfl_xor(ldp, o);
                oa += __shfl_xor(oa, o);
            }
            if (lane == 0) {
                part[j] = ldp;
                part[nblk + j] = oa;
            }
        }
    }
}

#ifdef S2_TIMING
#include <cstdio>
void dump_s2_timing(int nblk)
{
    static long long h[256][6];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_s2_ts), sizeof(h));
    const long long t0 = h[nblk - 1][0];
    printf("backward sweep with M_j (us after the first workgroup started): block | start | M_j done | g_j ready | a_{j+1} seen | a_j published (hop)\n");
    for (int j = nblk - 1; j >= 0; --j)
        printf("  %3d | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f (%5.2f)\n", j, (h[j][0] - t0) * 0.01, (h[j][1] - t0) * 0.01, (h[j][2] - t0) * 0.01,
               (h[j][3] - t0) * 0.01, (h[j][4] - t0) * 0.01, j + 1 < nblk ? (h[j][4] - h[j + 1][4]) * 0.01 : 0.0);
}
#endif

// a <- L^-T y, one right-hand side (y[i * ysi]); `a` holds the all-ones pattern (prefilled) or is filled here; part (optional,
// 2 nblk doubles) as launch_trsv_bwd_flow's.  nblk <= 256: every workgroup of the launch must be resident (the caller checks).
void launch_trsv_bwd_m(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all, const double* y, int64_t ysi, double* a,
                       int* err, int prefilled, const double* om, double* part)
{
    const int64_t nblk = (N + NB - 1) / NB;
    if (!prefilled)
        hipMemsetAsync(a, 0xFF, sizeof(double) * (size_t)N, s);
    FlowGate gate(s); // (one data-flow launch at a time on the device: dev.h)
    GPE_LAUNCH(k_trsv_bwd_m, dim3(GPE_FLOW_GRID(nblk)), dim3(512), 0, s, L, ld, N, Xt_all, y, ysi, a, err, om, part);
}
