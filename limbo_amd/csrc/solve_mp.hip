// solve_mp.hip — the one-launch triangular sweeps of solve.hip for SEVERAL right-hand sides at once (gfx950).
//
// k_trsv_bwd_flow / k_trsv_fwd_flow (solve.hip) walk their right-hand sides one after the other: a GP with P
// outputs (GP::_compute_alpha, src/limbo/model/gp.hpp:605-611, obs_mean is N x P) or a handful of query points
// (gp.hpp:620) pays the whole 64-hop chain P times.  Here the NP <= 4 right-hand sides of a pass travel together:
// the tile of L a workgroup fetches per contributor is the same for all of them, so one pass costs little more
// than a single right-hand side.  Wave p (p < NP) is the one that looks at right-hand side p — it holds the
// prefetched first look at that value, polls it, and owns w_j[p] — so the register budget of the single-RHS
// kernels is unchanged.  Same hand-off as there: output pre-filled with an all-ones pattern, one 8-byte
// agent-scope store per value, value-polling loads, bounded polling that raises *err.
#include "dev.h"

#define NB 64
#define LSTR 65
#define FW 8
#define FQ (NB / FW)

static __device__ __forceinline__ double mp_poll(const double* addr, unsigned long long peek, int* __restrict__ err)
{
    const unsigned long long SENT = ~0ull;
    unsigned long long bits = peek;
    int spins = 0;
    while (bits == SENT) { // three looks this XCD's L2 may serve for every one that goes to memory (solve.hip)
        if ((spins & 3) != 3)
            bits = __hip_atomic_load((const unsigned long long*)addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            bits = __hip_atomic_load((const unsigned long long*)addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bits != SENT)
            break;
        if (++spins > GPE_FLOW_SPIN_LIMIT) { // ~seconds: a lost producer, never a legal state
            *err = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return __longlong_as_double((long long)bits);
}

// fixed-order sum over the 8 waves of part_s[w][p][lane]
template <int NP>
static __device__ __forceinline__ double mp_sum(const double (&ps)[FW][NP][NB], int p, int lane)
{
    double s = ps[0][p][lane];
#pragma unroll
    for (int w = 1; w < FW; ++w)
        s += ps[w][p][lane];
    return s;
}

// ---- backward: a = L^-T y, P <= NP right-hand sides together ---------------------------------------------------
template <int NP>
__global__ __launch_bounds__(64 * FW) void k_trsv_bwd_flow_mp(const double* __restrict__ L, int64_t ld, int64_t N,
                                                          const double* __restrict__ Xt_all,
                                                          const double* __restrict__ y, int64_t ysi, int64_t ysp,
                                                          double* a, int64_t ldw, int P, int* __restrict__ err,
                                                          const double* __restrict__ om, int64_t ldom,
                                                          double* __restrict__ part, int part_acc,
                                                          const BatchTab* __restrict__ bt)
{
    BT_REBASE(bt, L);
    BT_REBASE(bt, Xt_all);
    BT_REBASE(bt, y);
    BT_REBASE(bt, a);
    BT_REBASE(bt, err);
    BT_REBASE(bt, om);
    BT_REBASE(bt, part);
    __shared__ double Stg[NB * LSTR];
    __shared__ double xs[NP][NB];
    __shared__ double wj[NP][NB];
    __shared__ double part_s[FW][NP][NB];
    __shared__ double red_s[NP + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int64_t nblk = (N + NB - 1) / NB;
    const int64_t j = flow_block_of(nblk, true); // dev.h
    if (j < 0)
        return;
    const int64_t j0 = j * NB;
    const int jb = (int)((N - j0 < NB) ? N - j0 : NB);
    const unsigned long long SENT = ~0ull;
    const bool mine = wvu < NP && wvu < P;                         // this wave looks after right-hand side wvu
    double* ap = a + (int64_t)(mine ? wvu : 0) * ldw;              // (the others read RHS 0's slot and ignore it)
    double ld_part = 0.0;
    if (part && threadIdx.x < NB && lane < jb)
        ld_part = log(L[(j0 + lane) + (j0 + lane) * ld]);
    if (wvu < NP)
        wj[wvu][lane] = (mine && lane < jb) ? y[(j0 + lane) * ysi + wvu * ysp] : 0.0;

    double tl[4][FQ];
    unsigned long long pb[4] = {SENT, SENT, SENT, SENT};
    auto fetch = [&](double (&dst)[FQ], unsigned long long& peek, int64_t tt) {
        const int64_t t0 = tt * NB;
        const int tb = (int)((N - t0 < NB) ? N - t0 : NB);
        const int kc = lane < tb ? lane : tb - 1;
#pragma unroll
        for (int q = 0; q < FQ; ++q) { // (the ragged tile's mask: in fold, where the tile is used — sweep2.hip says why)
            const int c = wvu + FW * q;
            const int cc = c < jb ? c : jb - 1;
            const double* col = L + t0 + (j0 + cc) * ld;
            dst[q] = col[kc];
        }
        peek = __hip_atomic_load((const unsigned long long*)(ap + t0 + kc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto fold = [&](const double (&src)[FQ], unsigned long long peek, int64_t t) {
        __syncthreads(); // Stg / xs / part_s of the previous contributor are consumed
        const double rowmask = lane < ((N - t * NB < NB) ? (int)(N - t * NB) : NB) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < FQ; ++q)
            Stg[(wv + FW * q) * LSTR + lane] = src[q] * (wvu + FW * q < jb ? rowmask : 0.0); // Stg[c][k]
        if (wvu < NP) {
            const int64_t t0 = t * NB;
            const int tb = (int)((N - t0 < NB) ? N - t0 : NB);
            xs[wvu][lane] = (mine && lane < tb) ? mp_poll(ap + t0 + lane, peek, err) : 0.0;
        }
        __syncthreads();
        double acc[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
            acc[p] = 0.0;
#pragma unroll
        for (int kk = 0; kk < FQ; ++kk) {
            const int k = FQ * wv + kk;
            const double s = Stg[lane * LSTR + k];
#pragma unroll
            for (int p = 0; p < NP; ++p)
                acc[p] = fma(s, xs[p][k], acc[p]);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p)
            part_s[wv][p][lane] = acc[p];
        __syncthreads();
        if (wvu < NP)
            wj[wvu][lane] -= mp_sum<NP>(part_s, wvu, lane);
    };
    int64_t t = nblk - 1;
    auto clampt = [&](int64_t tt) { return tt > j ? tt : (j + 1 < nblk ? j + 1 : nblk - 1); };
    if (t > j) {
        fetch(tl[0], pb[0], clampt(t));
        fetch(tl[1], pb[1], clampt(t - 1));
        fetch(tl[2], pb[2], clampt(t - 2));
        fetch(tl[3], pb[3], clampt(t - 3));
    }
    for (; t - 3 > j; t -= 4) {
        fold(tl[0], pb[0], t);
        fetch(tl[0], pb[0], clampt(t - 4));
        fold(tl[1], pb[1], t - 1);
        fetch(tl[1], pb[1], clampt(t - 5));
        fold(tl[2], pb[2], t - 2);
        fetch(tl[2], pb[2], clampt(t - 6));
        fold(tl[3], pb[3], t - 3);
        fetch(tl[3], pb[3], clampt(t - 7));
    }
    if (t > j)
        fold(tl[0], pb[0], t);
    if (t - 1 > j)
        fold(tl[1], pb[1], t - 1);
    if (t - 2 > j)
        fold(tl[2], pb[2], t - 2);
    __syncthreads();
    // a_j = X_j^T w_j :  a[p][c] = sum_r Xt[c + 64 r] w[p][r]   (coalesced along c), the 8 waves split r
    const double* Xt = Xt_all + j * (NB * NB);
    {
        double acc[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
            acc[p] = 0.0;
#pragma unroll
        for (int kk = 0; kk < FQ; ++kk) {
            const int r = FQ * wv + kk;
            const double x = Xt[lane + NB * r];
#pragma unroll
            for (int p = 0; p < NP; ++p)
                acc[p] = fma(x, wj[p][r], acc[p]);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p)
            part_s[wv][p][lane] = acc[p];
    }
    __syncthreads();
    double oa = 0.0;
    if (mine && lane < jb) {
        const double v = mp_sum<NP>(part_s, wvu, lane);
        __hip_atomic_store((unsigned long long*)(ap + j0 + lane), (unsigned long long)__double_as_longlong(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (part)
            oa = om[j0 + lane + (int64_t)wvu * ldom] * v;
    }
    if (part) { // per-block partial sums of the log-likelihood terms (gp.hpp:274-277), fixed order
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ld_part += __shfl_xor(ld_part, o);
            oa += __shfl_xor(oa, o);
        }
        if (wvu < NP && lane == 0)
            red_s[wvu] = oa;
        if (threadIdx.x == 0)
            red_s[NP] = ld_part;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int p = 0; p < NP; ++p)
                s += red_s[p];
            if (!part_acc)
                part[j] = red_s[NP];
            part[nblk + j] = (part_acc ? part[nblk + j] : 0.0) + s;
        }
    }
}

// ---- forward: y = L^-1 b, P <= NP right-hand sides together ----------------------------------------------------
template <int NP>
__global__ __launch_bounds__(64 * FW) void k_trsv_fwd_flow_mp(const double* __restrict__ L, int64_t ld, int64_t N,
                                                          const double* __restrict__ Xt_all,
                                                          const double* __restrict__ b, int64_t ldb, double* yout,
                                                          int64_t ldy, int P, int* __restrict__ err)
{
    __shared__ double Xs[NB * LSTR]; // Xs[c][k] = (L_jj^-1)[c][k]
    __shared__ double xs[NP][NB];
    __shared__ double wj[NP][NB];
    __shared__ double part_s[FW][NP][NB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int64_t nblk = (N + NB - 1) / NB;
    const int64_t j = flow_block_of(nblk, false); // dev.h
    if (j < 0)
        return;
    const int64_t j0 = j * NB;
    const int jb = (int)((N - j0 < NB) ? N - j0 : NB);
    const unsigned long long SENT = ~0ull;
    const bool mine = wvu < NP && wvu < P;
    double* yp = yout + (int64_t)(mine ? wvu : 0) * ldy;
    {
        const double* Xt = Xt_all + j * (NB * NB); // Xt[k + 64 c] = (L_jj^-1)[c][k], identity-padded past jb
        for (int e = threadIdx.x; e < NB * NB; e += 64 * FW)
            Xs[(e >> 6) * LSTR + (e & 63)] = Xt[e];
    }
    if (wvu < NP)
        wj[wvu][lane] = (mine && lane < jb) ? b[j0 + lane + (int64_t)wvu * ldb] : 0.0;
    const int rc = lane < jb ? lane : jb - 1;
    const double rowmask = lane < jb ? 1.0 : 0.0;

    double tl[4][FQ];
    unsigned long long pb[4] = {SENT, SENT, SENT, SENT};
    auto fetch = [&](double (&dst)[FQ], unsigned long long& peek, int64_t tt) {
        const int64_t t0 = tt * NB;
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const double* col = L + j0 + (t0 + wvu + FW * q) * ld;
            dst[q] = col[rc]; // (masked where it is used, below)
        }
        peek = __hip_atomic_load((const unsigned long long*)(yp + t0 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto fold = [&](const double (&src)[FQ], unsigned long long peek, int64_t t) {
        __syncthreads(); // xs / part_s of the previous contributor are consumed
        if (wvu < NP)
            xs[wvu][lane] = mine ? mp_poll(yp + t * NB + lane, peek, err) : 0.0;
        __syncthreads();
        double acc[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
            acc[p] = 0.0;
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
                acc[p] = fma(src[q] * rowmask, xs[p][wvu + FW * q], acc[p]);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p)
            part_s[wv][p][lane] = acc[p];
        __syncthreads();
        if (wvu < NP)
            wj[wvu][lane] -= mp_sum<NP>(part_s, wvu, lane);
    };
    int64_t t = 0;
    auto clampt = [&](int64_t tt) { return tt < j ? tt : (j > 0 ? j - 1 : 0); };
    if (j > 0) {
        fetch(tl[0], pb[0], clampt(0));
        fetch(tl[1], pb[1], clampt(1));
        fetch(tl[2], pb[2], clampt(2));
        fetch(tl[3], pb[3], clampt(3));
    }
    for (; t + 3 < j; t += 4) {
        fold(tl[0], pb[0], t);
        fetch(tl[0], pb[0], clampt(t + 4));
        fold(tl[1], pb[1], t + 1);
        fetch(tl[1], pb[1], clampt(t + 5));
        fold(tl[2], pb[2], t + 2);
        fetch(tl[2], pb[2], clampt(t + 6));
        fold(tl[3], pb[3], t + 3);
        fetch(tl[3], pb[3], clampt(t + 7));
    }
    if (t < j)
        fold(tl[0], pb[0], t);
    if (t + 1 < j)
        fold(tl[1], pb[1], t + 1);
    if (t + 2 < j)
        fold(tl[2], pb[2], t + 2);
    __syncthreads();
    // y_j = L_jj^-1 w_j :  y[p][c] = sum_k Xs[c][k] w[p][k], the 8 waves split k
    {
        double acc[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
            acc[p] = 0.0;
#pragma unroll
        for (int kk = 0; kk < FQ; ++kk) {
            const int k = FQ * wv + kk;
            const double x = Xs[lane * LSTR + k];
#pragma unroll
            for (int p = 0; p < NP; ++p)
                acc[p] = fma(x, wj[p][k], acc[p]);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p)
            part_s[wv][p][lane] = acc[p];
    }
    __syncthreads();
    if (mine && lane < jb) {
        const double v = mp_sum<NP>(part_s, wvu, lane);
        __hip_atomic_store((unsigned long long*)(yp + j0 + lane), (unsigned long long)__double_as_longlong(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// P in 2..4 right-hand sides in one pass (the callers in solve.hip split larger P into passes of 4 and take
// P = 1 themselves).  `a` / `y` already hold the sentinel.
void launch_trsv_bwd_flow_mp(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all,
                             const double* y, int64_t ysi, int64_t ysp, double* a, int64_t ldw, int P, int* err,
                             const double* om, int64_t ldom, double* part, int part_acc)
{
    const unsigned nblk = (unsigned)((N + NB - 1) / NB);
    if (P <= 2)
        GPE_LAUNCH((k_trsv_bwd_flow_mp<2>), dim3(g_batch.bt ? (unsigned)nblk : GPE_FLOW_GRID(nblk), 1, g_batch.G), dim3(64 * FW), 0, s, L, ld, N, Xt_all, y, ysi,
                           ysp, a, ldw, P, err, om, ldom, part, part_acc, g_batch.bt);
    else
        GPE_LAUNCH((k_trsv_bwd_flow_mp<4>), dim3(g_batch.bt ? (unsigned)nblk : GPE_FLOW_GRID(nblk), 1, g_batch.G), dim3(64 * FW), 0, s, L, ld, N, Xt_all, y, ysi,
                           ysp, a, ldw, P, err, om, ldom, part, part_acc, g_batch.bt);
}
void launch_trsv_fwd_flow_mp(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all,
                             const double* b, int64_t ldb, double* y, int64_t ldy, int P, int* err)
{
    const unsigned nblk = (unsigned)((N + NB - 1) / NB);
    if (P <= 2)
        GPE_LAUNCH((k_trsv_fwd_flow_mp<2>), dim3(GPE_FLOW_GRID(nblk)), dim3(64 * FW), 0, s, L, ld, N, Xt_all, b, ldb, y, ldy, P,
                           err);
    else
        GPE_LAUNCH((k_trsv_fwd_flow_mp<4>), dim3(GPE_FLOW_GRID(nblk)), dim3(64 * FW), 0, s, L, ld, N, Xt_all, b, ldb, y, ldy, P,
                           err);
}
