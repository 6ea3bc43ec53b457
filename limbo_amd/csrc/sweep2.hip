// sweep2.hip — the backward sweep  a = L^-T y  (GP::_compute_alpha, src/limbo/model/gp.hpp:605-611, second solve) as a CHAIN
// workgroup fed by helper workgroups (round 6).
//
// k_trsv_bwd_flow (solve.hip, rounds 1-5) gives every 64-row block j a workgroup that folds in L[t, j]^T a_t as the a_t appear and
// then forms a_j = X_j^T w_j.  Its chain is 64 hops of 1.7 us at N = 4096 (profiles/r04_sweep_stamps.log) and most of a hop is not
// arithmetic: a_{j+1} stored -> visible to the next workgroup -> polled -> staged through LDS -> folded (1.1 us), then the
// product with X_j (0.6 us).  Every hop crosses from one workgroup to another.
//
// Here ONE workgroup walks the whole chain and never waits for memory on it:
//     a_j = f_j - M_j a_{j+1} - M2_j a_{j+2},        M_j = X_j^T L[j+1, j]^T,   M2_j = X_j^T L[j+2, j]^T   (64 x 64 each)
//     f_j = X_j^T ( y_j - sum_{t >= j+3} L[t, j]^T a_t )
// f_j, M_j and M2_j come from HELPER workgroups, one per block: helper j multiplies its two tiles by X_j^T on the matrix cores
// when the launch starts (all helpers at once: ~6 us, once), then folds the far contributions t = nblk-1 .. j+3 exactly as
// k_trsv_bwd_flow does and publishes f_j — three hops before the chain needs it.  A hop of the chain is two 64 x 64
// matrix-vector products from registers (a thread holds 8 + 8 entries of M_j and M2_{j-1}, prefetched three hops ahead, and takes
// a_{j+1} from its own wave by v_readlane), partial sums through LDS, ONE barrier: ~0.2 us of arithmetic; what bounds it is the
// helpers' hand-off (a_{j+3} visible -> fold -> X_j^T -> f_j visible: ~2 us over three hops) and the 64 KB of M the chain's one CU
// reads per hop.
// Hand-offs: a (the result, sentinel-prefilled, one 8-byte agent-scope store per value, value-polled: as k_trsv_bwd_flow), f (the
// same; the chain puts the sentinel back behind its read, so the slots are armed again when the launch ends), M (written with
// agent-scope stores, then a flag word = the launch's epoch behind s_waitcnt vmcnt(0); the chain waits for all flags once).
// Chain and helpers wait for EACH OTHER: all 1 + nblk <= 257 workgroups must become resident — they do whenever nothing else
// that waits inside a launch holds the CUs (dev.h: FlowGate; kernels that merely run finish and make room).  Polls are bounded; a
// timeout raises *err and the host re-runs the sweep block by block (engine.hip: flow_failed), as for every data-flow launch.
// One right-hand side, single-GP launches; everything else keeps k_trsv_bwd_flow.
#include "gemm_glds64.h"
#include <cstdio>
#include <cstdlib>

#define NB 64
#define S2_PS 80 // stride of the [kk][i] operand tiles in LDS (== 16 mod 32, as potrf.hip's PS)

struct Sweep2Args {
    const double* L;
    int64_t ld, N;
    const double* Xt_all;
    const double* y;
    int64_t ysi; // y[i * ysi]
    double* a;   // out, all-ones when the launch starts
    int* err;
    const double* om; // obs_mean's column (for the om . a partial sums), or null
    double* part;     // 2 nblk doubles: sum log L_ii per block | sum om a per block; or null
    double* M;        // 2 nblk tiles of 4096 doubles: M_j | M2_j, element (c, k) at (k >> 1) * 128 + 2 c + (k & 1)
    double* f;        // nblk x 64 doubles, all-ones between launches
    unsigned long long* flag; // nblk words
    unsigned long long epoch;
    long long* dbg; // TEMP: stamps
};

static __device__ __forceinline__ unsigned long long s2_poll(const double* p, int* err)
{
    const unsigned long long SENT = ~0ull;
    unsigned long long b = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (b == SENT) {
        if (++spins > GPE_FLOW_SPIN_LIMIT) {
            *err = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        b = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return b;
}

// ---- helper j -------------------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void s2_helper(const Sweep2Args& g, const int64_t j, double* __restrict__ lds)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int64_t nblk = (g.N + NB - 1) / NB, j0 = j * NB;
    const int jb = (int)((g.N - j0 < NB) ? g.N - j0 : NB);
    const double* Xt = g.Xt_all + j * (NB * NB); // Xt[c + 64 r] = X[r][c]
    double* As = lds;               // [r][c], stride S2_PS
    double* Bs = lds + NB * S2_PS;  // [r][k]
    // ---- M_j = X_j^T L[j+1, j]^T and M2_j = X_j^T L[j+2, j]^T: M[c][k] = sum_r X[r][c] L[t0 + k][j0 + r] ----
    {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = wvu + 8 * q;
            As[r * S2_PS + lane] = Xt[lane + NB * r];
        }
        const int wm = (wvu & 1) * 32, wn = (wvu >> 1) * 16;
        const int ar = wm + (lane & 15), bc = wn + (lane & 3), kq = lane >> 4;
        const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4), dcol = lane & 3;
        for (int q2 = 1; q2 <= 2; ++q2) {
            const int64_t t = j + q2;
            if (t >= nblk)
                break;
            const int64_t t0 = t * NB;
            const int tb = (int)((g.N - t0 < NB) ? g.N - t0 : NB);
            const int kc = lane < tb ? lane : tb - 1;
            const double km = lane < tb ? 1.0 : 0.0;
            __syncthreads(); // Bs is free (and As is written)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = wvu + 8 * q;
                const int rc = r < jb ? r : jb - 1;
                Bs[r * S2_PS + lane] = g.L[t0 + kc + (j0 + rc) * g.ld] * (r < jb ? km : 0.0);
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
            double* Ms = g.M + (2 * j + (q2 - 1)) * (NB * NB);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int c = wm + 16 * m + drow, k = wn + 4 * n + dcol;
                    __hip_atomic_store(Ms + (k >> 1) * 128 + 2 * c + (k & 1), acc[m][n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the tiles are out before the flag is
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(g.flag + j, g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- far_j = y_j - sum_{t = nblk-1 .. j+3} L[t, j]^T a_t  (k_trsv_bwd_flow's fold, four tiles in flight) ----
    double* Stg = lds;                       // [c][k], stride 65
    double* xs = lds + NB * 65;              // a_t
    double* wj = xs + NB;                    // the running right-hand side
    double* part_s = wj + NB;                // [8][64]
    const unsigned long long SENT = ~0ull;
    __syncthreads();
    if (threadIdx.x < NB)
        wj[lane] = (lane < jb) ? g.y[(j0 + lane) * g.ysi] : 0.0;
    double tl[4][8];
    unsigned long long pb[4] = {SENT, SENT, SENT, SENT};
    auto fetch = [&](double (&dst)[8], unsigned long long& peek, int64_t tt) {
        const int64_t t0 = tt * NB;
        const int tb = (int)((g.N - t0 < NB) ? g.N - t0 : NB);
        const int kc = lane < tb ? lane : tb - 1;
        const double rowmask = lane < tb ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = wvu + 8 * q;
            const int cc = c < jb ? c : jb - 1;
            dst[q] = g.L[t0 + kc + (j0 + cc) * g.ld] * (c < jb ? rowmask : 0.0);
        }
        peek = __hip_atomic_load((const unsigned long long*)(g.a + t0 + kc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto fold = [&](const double (&src)[8], unsigned long long peek, int64_t t) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q)
            Stg[(wv + 8 * q) * 65 + lane] = src[q];
        if (threadIdx.x < NB) {
            const int64_t t0 = t * NB;
            const int tb = (int)((g.N - t0 < NB) ? g.N - t0 : NB);
            double v = 0.0;
            if (lane < tb) {
                unsigned long long bits = peek;
                if (bits == SENT)
                    bits = s2_poll(g.a + t0 + lane, g.err);
                v = __longlong_as_double((long long)bits);
            }
            xs[lane] = v;
        }
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
    const int64_t lo = j + 2; // contributors t > lo
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
    __syncthreads();
    // ---- f_j = X_j^T far_j ----
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
            __hip_atomic_store((unsigned long long*)(g.f + j * NB + lane), (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // sum_i log L_ii over the block (gp.hpp:274-277): wave 0, fixed-order butterfly
    if (g.part && threadIdx.x < NB) {
        double ldp = lane < jb ? log(g.L[(j0 + lane) + (j0 + lane) * g.ld]) : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            ldp += __shfl_xor(ldp, o);
        if (lane == 0)
            g.part[j] = ldp;
    }
}

// ---- the chain -------------------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ double s2_lane(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// thread (c = lane, ks = wave): entries k = 8 ks .. 8 ks + 7 of row c of a tile stored as (k >> 1) * 128 + 2 c + (k & 1)
static __device__ __forceinline__ void s2_load_rows(const double* __restrict__ tile, int c, int ks, bool live, double (&m)[8])
{
    // Agent-scope loads (the tiles were written by other CUs, possibly behind another XCD's L2, in THIS launch), and UNCONDITIONAL
    // ones: the address is always a valid tile, a tile that does not exist is cancelled bit-wise behind the load.  With the loads
    // under a branch the compiler cannot count how many younger ones are in flight and waits for all of them (vmcnt(0)) at the first
    // use — which made every hop wait for the loads it had just issued for three hops later.
    const unsigned long long keep = live ? ~0ull : 0ull;
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2) {
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(tile + (4 * ks + i2) * 128 + 2 * c);
        m[2 * i2] = __longlong_as_double((long long)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & keep));
        m[2 * i2 + 1] = __longlong_as_double((long long)(__hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & keep));
    }
}
// The chain workgroup is a data-flow of specialised waves (as diag_flow.h's block): waves 0..7 compute, wave 8 polls f, wave 9
// stores.  gfx9 counts loads and stores in ONE counter (vmcnt) and the compiler falls back to "wait for everything" wherever both
// kinds can be pending — with the stores of a_j in the same instruction stream as the three-hops-ahead loads of M every hop waited
// for the loads it had just issued (measured: 1.8 us a hop, the memory latency).  Here the compute waves only ever load, the store
// wave only ever stores, and the poll wave's waits are its own.  One barrier a hop, executed by all ten waves.
static __device__ __forceinline__ void s2_chain(const Sweep2Args& g, double* __restrict__ lds)
{
    const int c = threadIdx.x & 63, ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nblk = (g.N + NB - 1) / NB;
    double* pa_s = lds;              // [2][8][64]: this hop's partial sums of M_j a_{j+1}
    double* pb_s = lds + 2 * 8 * NB; // [2][8][64]: ... of M2_{j-1} a_{j+1}
    double* fs = lds + 4 * 8 * NB;   // [2][64]: f_j
    const unsigned long long SENT = ~0ull;
    // every helper's tiles of M are out (they are computed when the launch starts, all at once)
    for (int64_t q = threadIdx.x; q < nblk; q += blockDim.x) {
        int spins = 0;
        while (__hip_atomic_load(g.flag + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.epoch) {
            if (++spins > GPE_FLOW_SPIN_LIMIT) {
                *g.err = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    if (g.dbg && threadIdx.x == 0) g.dbg[1] = wall_clock64();
    if (ks == 8) { // ---- the poll wave: f_j into LDS in front of hop j's barrier ----
        unsigned long long peek = __hip_atomic_load((const unsigned long long*)(g.f + (nblk - 1) * NB + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int par = 0;
        for (int64_t j = nblk - 1; j >= 0; --j) {
            unsigned long long b = peek;
            if (b == SENT) {
                if (g.dbg && c == 0) g.dbg[300 + j] = 1;
                b = s2_poll(g.f + j * NB + c, g.err);
            }
            fs[par * NB + c] = __longlong_as_double((long long)b);
            if (j >= 1)
                peek = __hip_atomic_load((const unsigned long long*)(g.f + (j - 1) * NB + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            par ^= 1;
        }
        return;
    }
    if (ks == 9) { // ---- the store wave: a_j out, f_j's slot armed again, the om . a partial sum ----
        double carry = 0.0;
        int par = 0;
        for (int64_t j = nblk - 1; j >= 0; --j) {
            const int64_t j0 = j * NB;
            const int jb = (int)((g.N - j0 < NB) ? g.N - j0 : NB);
            __syncthreads();
            double sa = pa_s[(par * 8) * NB + c], sb = pb_s[(par * 8) * NB + c];
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                sa += pa_s[(par * 8 + w) * NB + c];
                sb += pb_s[(par * 8 + w) * NB + c];
            }
            const double aj = (fs[par * NB + c] - carry) - sa; // (the compute waves' own expression, bit for bit)
            carry = sb;
            if (c < jb)
                __hip_atomic_store((unsigned long long*)(g.a + j0 + c), (unsigned long long)__double_as_longlong(aj), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((unsigned long long*)(g.f + j * NB + c), SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g.part) {
                double oa = (g.om && c < jb) ? g.om[j0 + c] * aj : 0.0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
                    oa += __shfl_xor(oa, o);
                if (c == 0)
                    g.part[nblk + j] = oa;
            }
            if (g.dbg && c == 0) g.dbg[2 + j] = wall_clock64();
            par ^= 1;
        }
        return;
    }
    // ---- the compute waves ----
    double mA[3][8], mB[3][8]; // hop j: M_j and M2_{j-1}, both times a_{j+1}; three hops in flight
    auto load_hop = [&](int slot, int64_t j) {
        // M_j exists for 0 <= j <= nblk-2; M2_{j-1} for 1 <= j <= nblk-2
        const bool la = j >= 0 && j + 1 < nblk, lb = j >= 1 && j + 1 < nblk;
        s2_load_rows(g.M + (2 * (la ? j : 0)) * (NB * NB), c, ks, la, mA[slot]);
        s2_load_rows(g.M + (2 * (lb ? j - 1 : 0) + 1) * (NB * NB), c, ks, lb, mB[slot]);
    };
    double aj1 = 0.0;   // a_{j+1}[c] (every compute wave holds the whole block: lane = row)
    double carry = 0.0; // (M2_j a_{j+2})[c]
    load_hop(0, nblk - 1);
    load_hop(1, nblk - 2);
    load_hop(2, nblk - 3);
    int par = 0;
#define S2_HOP(SLOT, J)                                                                                                               \
    do {                                                                                                                              \
        const int64_t j = (J);                                                                                                        \
        {                                                                                                                             \
            double pa0 = 0.0, pa1 = 0.0, pb0 = 0.0, pb1 = 0.0;                                                                        \
            _Pragma("unroll") for (int i = 0; i < 8; i += 2)                                                                          \
            {                                                                                                                         \
                const double x0 = s2_lane(aj1, 8 * ks + i), x1 = s2_lane(aj1, 8 * ks + i + 1);                                        \
                pa0 = fma(mA[SLOT][i], x0, pa0);                                                                                      \
                pa1 = fma(mA[SLOT][i + 1], x1, pa1);                                                                                  \
                pb0 = fma(mB[SLOT][i], x0, pb0);                                                                                      \
                pb1 = fma(mB[SLOT][i + 1], x1, pb1);                                                                                  \
            }                                                                                                                         \
            pa_s[(par * 8 + ks) * NB + c] = pa0 + pa1;                                                                                \
            pb_s[(par * 8 + ks) * NB + c] = pb0 + pb1;                                                                                \
            load_hop(SLOT, j - 3);                                                                                                    \
            __syncthreads();                                                                                                          \
            double sa = pa_s[(par * 8) * NB + c], sb = pb_s[(par * 8) * NB + c];                                                      \
            _Pragma("unroll") for (int w = 1; w < 8; ++w)                                                                             \
            {                                                                                                                         \
                sa += pa_s[(par * 8 + w) * NB + c];                                                                                   \
                sb += pb_s[(par * 8 + w) * NB + c];                                                                                   \
            }                                                                                                                         \
            aj1 = (fs[par * NB + c] - carry) - sa;                                                                                    \
            carry = sb;                                                                                                               \
            par ^= 1;                                                                                                                 \
        }                                                                                                                             \
    } while (0)
    // whole groups of three hops without a branch inside (the loads' counting again); the last one or two hops behind them
    int64_t jj = nblk - 1;
    for (; jj >= 2; jj -= 3) {
        S2_HOP(0, jj);
        S2_HOP(1, jj - 1);
        S2_HOP(2, jj - 2);
    }
    if (jj >= 0)
        S2_HOP(0, jj);
    if (jj >= 1)
        S2_HOP(1, jj - 1);
#undef S2_HOP
}

__global__ __launch_bounds__(640) void k_trsv_bwd_chain(Sweep2Args g)
{
    __shared__ __attribute__((aligned(16))) double lds[2 * NB * S2_PS];
    if (blockIdx.x == 0)
        s2_chain(g, lds);
    else {
        if (threadIdx.x >= 512)
            return; // (a helper is eight waves; a barrier counts the waves still alive)
        const int64_t nblk = (g.N + NB - 1) / NB;
        s2_helper(g, nblk - (int64_t)blockIdx.x, lds); // the last blocks' helpers first
    }
}

// scratch for up to `blocks` blocks: [M: 2 blocks tiles | f: blocks x 64 | flags: blocks words] — laid out by the CAPACITY, not by
// the launch's own block count: launches of different orders on one handle find f where the previous one re-armed it
int64_t sweep2_scratch_doubles(int64_t blocks) { return blocks * (2 * NB * NB + NB + 1); }
// a <- L^-T y; `a` must hold the all-ones pattern (prefilled) or is filled here.  scratch: sweep2_scratch_doubles(cap_blocks) doubles
// whose f part holds all-ones and whose flags hold anything but `epoch` (the caller arms them once; a launch leaves them armed)
void launch_trsv_bwd_chain(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all, const double* y, int64_t ysi,
                           double* a, int* err, int prefilled, const double* om, double* part, double* scratch, int64_t cap_blocks,
                           unsigned long long epoch)
{
    const int64_t nblk = (N + NB - 1) / NB;
    if (nblk > cap_blocks)
        return;
    if (!prefilled)
        hipMemsetAsync(a, 0xFF, sizeof(double) * (size_t)N, s);
    Sweep2Args g{};
    g.L = L;
    g.ld = ld;
    g.N = N;
    g.Xt_all = Xt_all;
    g.y = y;
    g.ysi = ysi;
    g.a = a;
    g.err = err;
    g.om = om;
    g.part = part;
    g.M = scratch;
    g.f = scratch + cap_blocks * (2 * NB * NB);
    g.flag = reinterpret_cast<unsigned long long*>(g.f + cap_blocks * NB);
    g.epoch = epoch;
    static long long* dbg = nullptr;
    if (getenv("GPE_SWEEP_DBG") && !dbg) { hipMalloc(&dbg, 8 * 1024); }
    g.dbg = dbg;
    if (dbg) hipMemsetAsync(dbg, 0, 8 * 1024, s);
    FlowGate gate(s); // (chain and helpers wait for each other inside the launch: dev.h)
    GPE_LAUNCH(k_trsv_bwd_chain, dim3((unsigned)(1 + nblk)), dim3(640), 0, s, g);
    if (dbg) {
        static int calls = 0;
        if (++calls == 10) {
            hipStreamSynchronize(s);
            static long long h[1024];
            hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "sweep2 stamps (us): start 0, flags seen %.2f; hops:", (h[1] - h[0]) * 0.01);
            for (int64_t j = nblk - 1; j >= 0; --j)
                fprintf(stderr, " %lld:%.2f%s", (long long)j, (h[2 + j] - h[0]) * 0.01, h[300 + j] ? "*" : "");
            fprintf(stderr, "\n");
        }
    }
}
