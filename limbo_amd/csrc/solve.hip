// solve.hip — vector triangular sweeps, reductions and small utilities (gfx950).
//
// k_trsv_fwd_step / k_trsv_bwd_step: one 64-unknown block step of L y = b / L^T a = y
// (GP::_compute_alpha, src/limbo/model/gp.hpp:605-611), using the inverses of the 64 x 64
// diagonal blocks that the factorisation leaves behind (potrf.hip).
#include "dev.h"
#include <cstdlib>

#define NB 64
#define LSTR 65

// solve_mp.hip: 2..4 right-hand sides in one pass of the one-launch sweeps (output already holds the sentinel)
void launch_trsv_bwd_flow_mp(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all,
                             const double* y, int64_t ysi, int64_t ysp, double* a, int64_t ldw, int P, int* err,
                             const double* om, int64_t ldom, double* part, int part_acc);
void launch_trsv_fwd_flow_mp(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all,
                             const double* b, int64_t ldb, double* y, int64_t ldy, int P, int* err);

// xs[p][c] = sum_k M[c][k] v_p[k] with M = X (forward, TRANS = 0: M[c][k] = Xt[k + 64 c]) or
// M = X^T (backward, TRANS = 1: M[c][k] = Xt[c + 64 k]).  256 threads; v_p = w[j0 .. j0+jb) of
// right-hand side p (zero beyond jb).  Stg: NB*LSTR doubles of LDS scratch.  Ends with a barrier.
template <int TRANS>
__device__ __forceinline__ void diag_matvec(const double* __restrict__ Xt, const double* __restrict__ w, int64_t ldw,
                                            int64_t j0, int jb, int P, double* __restrict__ Stg,
                                            double (*__restrict__ xs)[NB], double (*__restrict__ part)[4][NB])
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Stg[c][k] = M[c][k]; global reads are coalesced along the fast index of Xt
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int f = e & 63, s = e >> 6; // Xt[f + 64 s]
        if (!TRANS)
            Stg[s * LSTR + f] = Xt[e]; // c = s, k = f
        else
            Stg[f * LSTR + s] = Xt[e]; // c = f, k = s
    }
    __syncthreads();
    for (int p = 0; p < P; ++p) {
        double acc = 0.0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int k = 16 * wv + kk;
            const double v = (k < jb) ? w[j0 + k + (int64_t)p * ldw] : 0.0;
            acc = fma(Stg[lane * LSTR + k], v, acc);
        }
        part[p][wv][lane] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 64)
        for (int p = 0; p < P; ++p)
            xs[p][lane] = (part[p][0][lane] + part[p][1][lane]) + (part[p][2][lane] + part[p][3][lane]);
    __syncthreads();
}

// One 64-unknown block step of L y = b / L^T a = y (GP::_compute_alpha, gp.hpp:605-611).  Every
// workgroup of a step first forms the block's solution redundantly as a 64 x 64 mat-vec with the
// inverse of the diagonal block (computed by k_diag during the factorisation) — redundancy costs
// no time and saves a kernel boundary per step — and then applies the rank-64 update to its own
// slab of the remaining right-hand side.  The solution goes to a separate `out` vector so no
// workgroup reads what another one overwrites.
// L: full matrix (col-major, ld); block at j0 of size jb; N = order; Xt: this block's inverse.
// w: running right-hand side (N x P, ldw); out: solution (N x P, ldw)
__global__ __launch_bounds__(256) void k_trsv_fwd_step(const double* __restrict__ L, int64_t ld, int64_t N,
                                                       int64_t j0, int jb, const double* __restrict__ Xt,
                                                       double* __restrict__ w, double* __restrict__ out, int64_t ldw,
                                                       int P)
{
    __shared__ double Stg[NB * LSTR];
    __shared__ double xs[GPE_MAX_P][NB];
    __shared__ double part[GPE_MAX_P][4][NB];
    diag_matvec<0>(Xt, w, ldw, j0, jb, P, Stg, xs, part);
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x < 64 && lane < jb)
        for (int p = 0; p < P; ++p)
            out[j0 + lane + (int64_t)p * ldw] = xs[p][lane];
    // update rows below: w[r] -= sum_k L[r][j0+k] x[k]
    const int64_t r = j0 + jb + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < N) {
        double acc[GPE_MAX_P];
#pragma unroll
        for (int p = 0; p < GPE_MAX_P; ++p)
            acc[p] = 0.0;
        const double* Lr = L + r + j0 * ld;
        for (int k = 0; k < jb; ++k) {
            const double l = Lr[(int64_t)k * ld];
#pragma unroll
            for (int p = 0; p < GPE_MAX_P; ++p)
                if (p < P)
                    acc[p] = fma(l, xs[p][k], acc[p]);
        }
#pragma unroll
        for (int p = 0; p < GPE_MAX_P; ++p)
            if (p < P)
                w[r + (int64_t)p * ldw] -= acc[p];
    }
}

__global__ __launch_bounds__(256) void k_trsv_bwd_step(const double* __restrict__ L, int64_t ld, int64_t N,
                                                       int64_t j0, int jb, const double* __restrict__ Xt,
                                                       double* __restrict__ w, double* __restrict__ out, int64_t ldw,
                                                       int P)
{
    __shared__ double Stg[NB * LSTR];
    __shared__ double xs[GPE_MAX_P][NB];
    __shared__ double part[GPE_MAX_P][4][NB];
    (void)N;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // this workgroup's 64 earlier unknowns c0..c0+63: prefetch its slab of L into registers first
    // (independent of the mat-vec):  T[k][c] = L[j0+k][c0+c], read coalesced along k
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    double tl[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int64_t col = c0 + wv + 4 * q;
        tl[q] = (c0 < j0 && col < j0 && lane < jb) ? L[j0 + lane + col * ld] : 0.0;
    }
    diag_matvec<1>(Xt, w, ldw, j0, jb, P, Stg, xs, part);
    if (blockIdx.x == 0 && threadIdx.x < 64 && lane < jb)
        for (int p = 0; p < P; ++p)
            out[j0 + lane + (int64_t)p * ldw] = xs[p][lane];
    if (c0 < j0) { // w[c] -= sum_k L[j0+k][c] x[k]
#pragma unroll
        for (int q = 0; q < 16; ++q)
            Stg[(wv + 4 * q) * LSTR + lane] = tl[q]; // Stg[c][k]
        __syncthreads();
        const int c = lane;
        const int64_t col = c0 + c;
        for (int p = 0; p < P; ++p) {
            double acc = 0.0;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int k = 16 * wv + kk;
                acc = fma(Stg[c * LSTR + k], xs[p][k], acc);
            }
            part[p][wv][c] = acc;
        }
        __syncthreads();
        if (threadIdx.x < 64 && col < j0)
            for (int p = 0; p < P; ++p)
                w[col + (int64_t)p * ldw] -= (part[p][0][c] + part[p][1][c]) + (part[p][2][c] + part[p][3][c]);
    }
}

// one full sweep = ceil(N/64) launches.  trans = 0: L y = b (top down); 1: L^T a = y (bottom up)
// Xt_all: inverses of the diagonal blocks (block b at Xt_all + 4096 b)
void launch_trsv_sweep(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all, double* w,
                       double* out, int64_t ldw, int P, int trans)
{
    if (N <= 0)
        return;
    const int64_t nblk = (N + NB - 1) / NB;
    if (!trans) {
        for (int64_t b = 0; b < nblk; ++b) {
            int64_t j0 = b * NB;
            int jb = (int)((N - j0 < NB) ? N - j0 : NB);
            int64_t rest = N - j0 - jb;
            unsigned grid = (unsigned)((rest + 255) / 256);
            if (grid == 0)
                grid = 1;
            GPE_LAUNCH(k_trsv_fwd_step, dim3(grid), dim3(256), 0, s, L, ld, N, j0, jb, Xt_all + b * NB * NB, w,
                               out, ldw, P);
        }
    }
    else {
        for (int64_t b = nblk - 1; b >= 0; --b) {
            int64_t j0 = b * NB;
            int jb = (int)((N - j0 < NB) ? N - j0 : NB);
            unsigned grid = (unsigned)((j0 + 63) / 64);
            if (grid == 0)
                grid = 1;
            GPE_LAUNCH(k_trsv_bwd_step, dim3(grid), dim3(256), 0, s, L, ld, N, j0, jb, Xt_all + b * NB * NB, w,
                               out, ldw, P);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_trsv_bwd_flow — the whole backward sweep L^T a = y in ONE launch (a dependent launch costs
// ~3.4 us here; the 64-launch sweep spent 0.5 ms at N = 4096, almost all of it launch floor).
// Workgroup j owns unknown block j.  It folds in the contributions L[t, j]^T a_t of the later
// blocks t = nblk-1 .. j+1 as those a_t appear in global memory, then forms a_j = X_j^T w_j with the
// stored block inverse and publishes it.  Hand-off without flags or fences: the output vector is
// pre-filled with an all-ones bit pattern (a NaN no arithmetic produces); every value is published
// by ONE naturally aligned 8-byte agent-scope store and consumed by agent-scope loads that poll the
// value itself (MI355X_MICROARCH.md, "granule" hand-off: an 8-byte store is never torn and needs no
// ordering with anything else).  A workgroup only waits for workgroups dispatched before it (flow_block_of,
// dev.h), so the pipeline makes progress whatever else occupies the chip; the poll is bounded anyway and raises
// *err instead of hanging (the host then re-runs the sweep block by block).
// ---------------------------------------------------------------------------------------------
#ifdef FLOW_TIMING
__device__ long long g_flow_ts[256][8]; // per block j: start, after fold #1, #2, #8, last fold done, solved, published, #folds
#define FTS(i) do { if (threadIdx.x == 0) g_flow_ts[j][i] = wall_clock64(); } while (0)
#else
#define FTS(i) do { } while (0)
#endif
// y is addressed as y[i * ysi + p * ysp]: a column per right-hand side (ysi = 1, ysp = ldw) or the rows
// appended under the factor by the fused forward solve (ysi = ld, ysp = 1) — no conversion launch.
// part (optional, 2 nblk doubles): workgroup j also leaves sum_i log L_ii over its block in part[j] and
// sum_{i,p} om[i,p] a[i,p] in part[nblk + j] (added to what is there when part_acc): the log-likelihood
// terms of gp.hpp:274-277 then need no launch of their own, the host adds the nblk partials in order.
// fixed-order sum of the per-wave partial results (bitwise reproducible)
template <int W>
static __device__ __forceinline__ double flow_sum(const double (&p)[W][NB], int lane)
{
    double s = p[0][lane];
#pragma unroll
    for (int w = 1; w < W; ++w)
        s += p[w][lane];
    return s;
}
unsigned flow_grid(int64_t nblk)
{
    static const bool xcd = !(getenv("GPE_FLOW_XCD") && atoi(getenv("GPE_FLOW_XCD")) == 0);
    return (unsigned)(xcd ? 8 * nblk : nblk);
}
#define FW 8            // waves per workgroup of the data-flow sweep (16 measured no better)
#define FQ (NB / FW)    // tile columns (and k-slices) per wave
static __device__ __forceinline__ void trsv_bwd_flow_body(const double* __restrict__ L, int64_t ld, int64_t N,
                                                           const double* __restrict__ Xt_all, const double* __restrict__ y,
                                                           int64_t ysi, int64_t ysp, double* a, int64_t ldw, int P,
                                                           int* __restrict__ err, const double* __restrict__ om,
                                                           int64_t ldom, double* __restrict__ part, int part_acc, int64_t bx, int64_t gx)
{
    __shared__ double Stg[NB * LSTR];
    __shared__ double xs[NB];
    __shared__ double wj[NB];
    __shared__ double part_s[FW][NB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv); // the same value, known to be wave-uniform
    const int64_t nblk = (N + NB - 1) / NB;
    // workgroup -> unknown block: dependencies only on lower blockIdx.x, consecutive blocks on one XCD (dev.h)
    const int64_t j = flow_block_of_at(nblk, true, bx, gx);
    if (j < 0)
        return;
    const int64_t j0 = j * NB;
    const int jb = (int)((N - j0 < NB) ? N - j0 : NB);
    const unsigned long long SENT = ~0ull;
    double ld_part = 0.0, oa_part = 0.0; // wave 0: this lane's log L_ii and sum_p om a
    if (part && threadIdx.x < NB && lane < jb)
        ld_part = log(L[(j0 + lane) + (j0 + lane) * ld]);
    FTS(0);
    for (int p = 0; p < P; ++p) {
        double* ap = a + (int64_t)p * ldw;
        if (threadIdx.x < NB)
            wj[lane] = (lane < jb) ? y[(j0 + lane) * ysi + p * ysp] : 0.0;
        // Contributors t = nblk-1 .. j+1, software-pipelined FOUR deep: both the tile T[k][c] = L[t0 + k][j0 + c]
        // (lane = k) and the first look at a_t are requested four folds ahead.  With one tile in flight and
        // the poll issued at the point of use, every fold paid a full memory round trip (~2.5 us), even for
        // contributions published long before — the last workgroup's 63 folds WERE the kernel time.
        double tl[4][FQ];
        unsigned long long pb[4] = {SENT, SENT, SENT, SENT};
        auto fetch = [&](double (&dst)[FQ], unsigned long long& peek, int64_t tt) {
            const int64_t t0 = tt * NB;
            const int tb = (int)((N - t0 < NB) ? N - t0 : NB);
            const int kc = lane < tb ? lane : tb - 1;
#pragma unroll
            for (int q = 0; q < FQ; ++q) {
                // Unconditional load from a clamped (valid) address, masked by a multiplication: written as
                // `cond ? load : 0` the compiler predicates every load and waits for each one in turn
                // (s_cbranch_execz + s_waitcnt vmcnt(0) per load: 8 serial round trips per tile, 3 us a fold).
                // The column is wave-uniform (wvu): scalar base + one 32-bit lane offset, no 64-bit vector
                // address per load (16 of those per tile in flight x 4 tiles spilled to scratch).
                const int c = wvu + FW * q;
                const int cc = c < jb ? c : jb - 1;
                const double* col = L + t0 + (j0 + cc) * ld;
                dst[q] = col[kc]; // (the mask of a ragged tile: in fold, where the tile is used — with the multiplication here the compiler
                                  // serialised the loads in front of the loop, sweep2.hip)
            }
            // first look at a_t (clamped address; lanes past the block read a valid neighbour and ignore it)
#ifdef FLOW_NOPOLL
            peek = 0x3ff0000000000000ull; // timing experiment: pretend every contribution is 1.0 and already there
#else
            peek = __hip_atomic_load((const unsigned long long*)(ap + t0 + kc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        };
        auto fold = [&](const double (&src)[FQ], unsigned long long peek, int64_t t) {
            __syncthreads(); // Stg / xs of the previous contributor are consumed
            const double rowmask = lane < ((N - t * NB < NB) ? (int)(N - t * NB) : NB) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < FQ; ++q)
                Stg[(wv + FW * q) * LSTR + lane] = src[q] * (wvu + FW * q < jb ? rowmask : 0.0); // Stg[c][k]
            if (threadIdx.x < NB) {
                const int64_t t0 = t * NB;
                const int tb = (int)((N - t0 < NB) ? N - t0 : NB);
                double v = 0.0;
                if (lane < tb) {
                    unsigned long long bits = peek;
                    int spins = 0;
                    while (bits == SENT) { // not there four folds ago: poll.  Three looks that this XCD's L2 may
                                           // serve (workgroup scope) for every one that goes to memory (agent
                                           // scope; the producer stores with agent scope): always correct,
                                           // quicker when producer and consumer share an XCD
                        if ((spins & 3) != 3)
                            bits = __hip_atomic_load((const unsigned long long*)(ap + t0 + lane), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP);
                        else
                            bits = __hip_atomic_load((const unsigned long long*)(ap + t0 + lane), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                        if (bits != SENT)
                            break;
                        if (++spins > GPE_FLOW_SPIN_LIMIT) { // ~seconds: a lost producer, never a legal state
                            *err = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    v = __longlong_as_double((long long)bits);
                }
#ifdef FLOW_TIMING
                if (t == j + 1) FTS(7); // the last contributor's values have arrived
#endif
                xs[lane] = v;
            }
            __syncthreads();
            double acc = 0.0;
#pragma unroll
            for (int kk = 0; kk < FQ; ++kk) {
                const int k = FQ * wv + kk;
                acc = fma(Stg[lane * LSTR + k], xs[k], acc);
            }
            part_s[wv][lane] = acc;
            __syncthreads();
            if (threadIdx.x < NB)
                wj[lane] -= flow_sum(part_s, lane);
#ifdef FLOW_TIMING
            {
                const int64_t nf = nblk - t; // folds done so far
                if (nf == 1) FTS(1);
                if (nf == 2) FTS(2);
                if (nf == 8) FTS(3);
                if (t == j + 1) FTS(4);
            }
#endif
        };
        // branch-free main loop: tile indices are clamped instead of guarded, so the compiler knows exactly how
        // many loads are younger than the ones a fold needs and waits for no more than that
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
        // 0-3 contributors left; their tiles are in tl[0..2]
        if (t > j)
            fold(tl[0], pb[0], t);
        if (t - 1 > j)
            fold(tl[1], pb[1], t - 1);
        if (t - 2 > j)
            fold(tl[2], pb[2], t - 2);
        __syncthreads();
        // a_j = X_j^T w_j :  a[c] = sum_r Xt[c + 64 r] w[r]   (coalesced along c)
        const double* Xt = Xt_all + j * (NB * NB);
        double acc = 0.0;
#pragma unroll
        for (int kk = 0; kk < FQ; ++kk) {
            const int r = FQ * wv + kk;
            acc = fma(Xt[lane + NB * r], wj[r], acc);
        }
        part_s[wv][lane] = acc;
        __syncthreads();
        FTS(5);
        if (threadIdx.x < NB && lane < jb) {
            const double v = flow_sum(part_s, lane);
            __hip_atomic_store((unsigned long long*)(ap + j0 + lane), (unsigned long long)__double_as_longlong(v),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (part)
                oa_part = fma(om[j0 + lane + (int64_t)p * ldom], v, oa_part);
        }
        FTS(6);
        __syncthreads();
    }
    if (part && threadIdx.x < NB) { // wave 0: fixed-order butterfly, bitwise reproducible
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ld_part += __shfl_xor(ld_part, o);
            oa_part += __shfl_xor(oa_part, o);
        }
        if (lane == 0) {
            if (!part_acc)
                part[j] = ld_part;
            part[nblk + j] = (part_acc ? part[nblk + j] : 0.0) + oa_part;
        }
    }
}
// entry points: the single-GP launch (exactly the round-1 kernel: no extra parameter, no occupancy bound) and the batched
// one (gridDim.z GPs, pointers rebased per GP; dev.h)
__global__ __launch_bounds__(64 * FW) void k_trsv_bwd_flow(const double* __restrict__ L, int64_t ld, int64_t N,
                                                       const double* __restrict__ Xt_all, const double* __restrict__ y,
                                                       int64_t ysi, int64_t ysp, double* a, int64_t ldw, int P,
                                                       int* __restrict__ err, const double* __restrict__ om,
                                                       int64_t ldom, double* __restrict__ part, int part_acc)
{
    trsv_bwd_flow_body(L, ld, N, Xt_all, y, ysi, ysp, a, ldw, P, err, om, ldom, part, part_acc, (int64_t)blockIdx.x, (int64_t)gridDim.x);
}
// G members in ONE 1-D grid, position by position: blockIdx.x = position * G + member, every member's chain in the plain form (a
// workgroup per block: the eight-per-position form of a single GP would put 8 G nblk workgroups through the dispatcher, seven of
// eight to return at once — 2048 at G = 8, N = 2048, 40 us of a 97 us launch), so that all members' chains advance side by side
// whatever number of workgroups is resident (member after member, gridDim.z, the second half of a batch of 32 started when the first
// had finished).  A wait is for a lower position of the same member: a lower-numbered workgroup.
__global__ __launch_bounds__(64 * FW) void k_trsv_bwd_flow_b(const double* __restrict__ L, int64_t ld, int64_t N,
                                                         const double* __restrict__ Xt_all, const double* __restrict__ y,
                                                         int64_t ysi, int64_t ysp, double* a, int64_t ldw, int P,
                                                         int* __restrict__ err, const double* __restrict__ om,
                                                         int64_t ldom, double* __restrict__ part, int part_acc,
                                                         const BatchTab* __restrict__ bt)
{
    const int G = bt->G, gp = (int)(blockIdx.x % G);
    L = bt_rebase(bt, gp, L);
    Xt_all = bt_rebase(bt, gp, Xt_all);
    y = bt_rebase(bt, gp, y);
    a = bt_rebase(bt, gp, a);
    err = bt_rebase(bt, gp, err);
    om = bt_rebase(bt, gp, om);
    part = bt_rebase(bt, gp, part);
    trsv_bwd_flow_body(L, ld, N, Xt_all, y, ysi, ysp, a, ldw, P, err, om, ldom, part, part_acc, (int64_t)(blockIdx.x / G), (int64_t)(gridDim.x / G));
}

#ifdef FLOW_TIMING
#include <cstdio>
void dump_flow_timing(int nblk)
{
    static long long h[256][8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_flow_ts), sizeof(h));
    long long t0 = h[nblk - 1][0];
    printf("backward flow solve (10 ns ticks since the first workgroup started): block | start | fold1 fold2 fold8 done | last contributor seen | last fold done | solved | published\n");
    for (int j = nblk - 1; j >= 0; j -= (j > nblk - 4 || j < 4) ? 1 : 6)
        printf("  %3d | %5lld | %5lld %5lld %5lld | %6lld | %6lld | %6lld | %6lld\n", j, h[j][0] - t0, h[j][1] - t0, h[j][2] - t0, h[j][3] - t0, h[j][7] - t0,
               h[j][4] - t0, h[j][5] - t0, h[j][6] - t0);
}
#endif
// a <- L^-T y in one launch (nblk <= 256: all workgroups resident); `a` must not alias y
void launch_trsv_bwd_flow(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all, const double* y,
                          int64_t ysi, int64_t ysp, double* a, int64_t ldw, int P, int* err, int prefilled, const double* om,
                          int64_t ldom, double* part, int part_acc)
{
    if (N <= 0)
        return;
    const int64_t nblk = (N + NB - 1) / NB;
    if (!prefilled)
        for (int p = 0; p < P; ++p)
            hipMemsetAsync(a + (int64_t)p * ldw, 0xFF, sizeof(double) * (size_t)N, s);
    FlowGate gate(s); // (one data-flow launch at a time on the device: dev.h)
    // several right-hand sides travel together, four to a pass (solve_mp.hip); a single one takes the kernel above
    for (int p0 = 0; p0 < P; p0 += 4) {
        const int pc = P - p0 < 4 ? P - p0 : 4;
        const double* yc = y + (int64_t)p0 * ysp;
        double* ac = a + (int64_t)p0 * ldw;
        const double* omc = om ? om + (int64_t)p0 * ldom : nullptr;
        const int acc = (part_acc || p0 > 0) ? 1 : 0;
        if (pc == 1)
        {
            if (g_batch.bt)
                GPE_LAUNCH(k_trsv_bwd_flow_b, dim3((unsigned)(nblk * g_batch.G)), dim3(64 * FW), 0, s, L, ld, N, Xt_all, yc,
                                   ysi, ysp, ac, ldw, 1, err, omc, ldom, part, acc, g_batch.bt);
            else
                GPE_LAUNCH(k_trsv_bwd_flow, dim3(GPE_FLOW_GRID(nblk)), dim3(64 * FW), 0, s, L, ld, N, Xt_all, yc, ysi, ysp, ac,
                                   ldw, 1, err, omc, ldom, part, acc);
        }
        else
            launch_trsv_bwd_flow_mp(s, L, ld, N, Xt_all, yc, ysi, ysp, ac, ldw, pc, err, omc, ldom, part, acc);
    }
}

// ---------------------------------------------------------------------------------------------
// k_trsv_fwd_flow — the forward sweep L y = b in one launch: the mirror image of k_trsv_bwd_flow, same
// hand-off (sentinel-prefilled output, one 8-byte agent-scope store per value, value-polling loads, bounded).
// Workgroup j owns unknown block j and folds in L[j, t] y_t for t = 0 .. j-1 as the y_t appear.  The tile
// L[j0 + r][t0 + c] is fetched with lane = r (512 contiguous bytes per column), wave w takes columns
// w, w + 8, ..: the product needs no transposition through LDS, only the sum over the 8 waves.
// Used where a handful of right-hand sides meet a factor that is already there: the per-point query
// (gp.hpp:620) and add_sample's new row of L (gp.hpp:591-594) — 2 N/64 dependent launches otherwise.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * FW) void k_trsv_fwd_flow(const double* __restrict__ L, int64_t ld, int64_t N,
                                                       const double* __restrict__ Xt_all, const double* __restrict__ b,
                                                       int64_t ldb, double* yout, int64_t ldy, int P,
                                                       int* __restrict__ err)
{
    __shared__ double Xs[NB * LSTR]; // Xs[c][k] = (L_jj^-1)[c][k]
    __shared__ double xs[NB];
    __shared__ double wj[NB];
    __shared__ double part_s[FW][NB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int64_t nblk = (N + NB - 1) / NB;
    const int64_t j = flow_block_of(nblk, false); // dev.h
    if (j < 0)
        return;
    const int64_t j0 = j * NB;
    const int jb = (int)((N - j0 < NB) ? N - j0 : NB);
    const unsigned long long SENT = ~0ull;
    {
        const double* Xt = Xt_all + j * (NB * NB); // Xt[k + 64 c] = (L_jj^-1)[c][k], identity-padded past jb
        for (int e = threadIdx.x; e < NB * NB; e += 64 * FW)
            Xs[(e >> 6) * LSTR + (e & 63)] = Xt[e];
    }
    const int rc = lane < jb ? lane : jb - 1;
    const double rowmask = lane < jb ? 1.0 : 0.0;
    for (int p = 0; p < P; ++p) {
        double* yp = yout + (int64_t)p * ldy;
        __syncthreads(); // wj of the previous right-hand side is consumed (and Xs is in place)
        if (threadIdx.x < NB)
            wj[lane] = (lane < jb) ? b[j0 + lane + (int64_t)p * ldb] : 0.0;
        // contributors t = 0 .. j-1 (all of them full blocks), tiles and first looks at y_t four folds ahead
        double tl[4][FQ];
        unsigned long long pb[4] = {SENT, SENT, SENT, SENT};
        auto fetch = [&](double (&dst)[FQ], unsigned long long& peek, int64_t tt) {
            const int64_t t0 = tt * NB;
#pragma unroll
            for (int q = 0; q < FQ; ++q) {
                const double* col = L + j0 + (t0 + wvu + FW * q) * ld; // wave-uniform column base
                dst[q] = col[rc]; // clamped row (no predicated loads); the mask where the tile is used, fold (sweep2.hip says why)
            }
            peek = __hip_atomic_load((const unsigned long long*)(yp + t0 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto fold = [&](const double (&src)[FQ], unsigned long long peek, int64_t t) {
            __syncthreads(); // xs / part_s of the previous contributor are consumed
            if (threadIdx.x < NB) {
                const int64_t t0 = t * NB;
                unsigned long long bits = peek;
                int spins = 0;
                while (bits == SENT) {
                    if ((spins & 3) != 3)
                        bits = __hip_atomic_load((const unsigned long long*)(yp + t0 + lane), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP);
                    else
                        bits = __hip_atomic_load((const unsigned long long*)(yp + t0 + lane), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
                    if (bits != SENT)
                        break;
                    if (++spins > GPE_FLOW_SPIN_LIMIT) { // ~seconds: a lost producer, never a legal state
                        *err = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                xs[lane] = __longlong_as_double((long long)bits);
            }
            __syncthreads();
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < FQ; ++q)
                acc = fma(src[q] * rowmask, xs[wvu + FW * q], acc);
            part_s[wv][lane] = acc;
            __syncthreads();
            if (threadIdx.x < NB)
                wj[lane] -= flow_sum(part_s, lane);
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
        // y_j = L_jj^-1 w_j :  y[c] = sum_k Xs[c][k] w[k], the 8 waves split k
        double acc = 0.0;
#pragma unroll
        for (int kk = 0; kk < FQ; ++kk) {
            const int k = FQ * wv + kk;
            acc = fma(Xs[lane * LSTR + k], wj[k], acc);
        }
        part_s[wv][lane] = acc;
        __syncthreads();
        if (threadIdx.x < NB && lane < jb) {
            const double v = flow_sum(part_s, lane);
            __hip_atomic_store((unsigned long long*)(yp + j0 + lane), (unsigned long long)__double_as_longlong(v),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// y <- L^-1 b in one launch (nblk <= 256: all workgroups resident; the caller checks); y must not alias b
void launch_trsv_fwd_flow(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all, const double* b,
                          int64_t ldb, double* y, int64_t ldy, int P, int* err)
{
    if (N <= 0 || P <= 0)
        return;
    const int64_t nblk = (N + NB - 1) / NB;
    for (int p = 0; p < P; ++p)
        hipMemsetAsync(y + (int64_t)p * ldy, 0xFF, sizeof(double) * (size_t)N, s);
    FlowGate gate(s); // (one data-flow launch at a time on the device: dev.h)
    for (int p0 = 0; p0 < P; p0 += 4) { // four right-hand sides to a pass (solve_mp.hip)
        const int pc = P - p0 < 4 ? P - p0 : 4;
        const double* bc = b + (int64_t)p0 * ldb;
        double* yc = y + (int64_t)p0 * ldy;
        if (pc == 1)
            GPE_LAUNCH(k_trsv_fwd_flow, dim3(GPE_FLOW_GRID(nblk)), dim3(64 * FW), 0, s, L, ld, N, Xt_all, bc, ldb, yc, ldy, 1,
                               err);
        else
            launch_trsv_fwd_flow_mp(s, L, ld, N, Xt_all, bc, ldb, yc, ldy, pc, err);
    }
}

// rows N..N+P-1 of the matrix <- obs_mean^T (before the factorisation) and back (z = L^-1 obs_mean
// after it): the forward substitution rides along the Cholesky as P extra rows of the panel.
// sent (optional): N x P vector (ld = ldv) pre-filled with the all-ones pattern the data-flow backward
// sweep polls for — saves that sweep a memset on the critical path
__global__ void k_cols_to_rows(const double* __restrict__ V, int64_t ldv, int64_t N, int P, double* __restrict__ Arows,
                               int64_t lda, double* __restrict__ sent, const BatchTab* __restrict__ bt)
{
    BT_REBASE(bt, V);
    BT_REBASE(bt, Arows);
    BT_REBASE(bt, sent);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N)
        for (int p = 0; p < P; ++p) {
            Arows[p + i * lda] = V[i + (int64_t)p * ldv];
            if (sent)
                ((unsigned long long*)sent)[i + (int64_t)p * ldv] = ~0ull;
        }
}
__global__ void k_rows_to_cols(const double* __restrict__ Arows, int64_t lda, int64_t N, int P, double* __restrict__ V,
                               int64_t ldv)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N)
        for (int p = 0; p < P; ++p)
            V[i + (int64_t)p * ldv] = Arows[p + i * lda];
}
void launch_cols_to_rows(hipStream_t s, const double* V, int64_t ldv, int64_t N, int P, double* Arows, int64_t lda,
                         double* sent)
{
    if (N > 0 && P > 0)
        GPE_LAUNCH(k_cols_to_rows, dim3((unsigned)((N + 255) / 256), 1, g_batch.G), dim3(256), 0, s, V, ldv, N, P, Arows, lda, sent,
                           g_batch.bt);
}
void launch_rows_to_cols(hipStream_t s, const double* Arows, int64_t lda, int64_t N, int P, double* V, int64_t ldv)
{
    if (N > 0 && P > 0)
        GPE_LAUNCH(k_rows_to_cols, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, Arows, lda, N, P, V, ldv);
}

// ---------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double* sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0)
        sh[wv] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3]; // fixed order: deterministic
}

// out[0] = sum_i log L_ii (gp.hpp:274) ; out[1] = trace(obs_mean^T alpha) (gp.hpp:276-277)
__global__ __launch_bounds__(256) void k_loglik_terms(const double* __restrict__ L, int64_t ldl, int64_t N,
                                                      const double* __restrict__ om, const double* __restrict__ al,
                                                      int64_t ldv, int P, double* __restrict__ out)
{
    __shared__ double sh[4];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += 256)
        s += log(L[i + i * ldl]);
    s = block_sum_256(s, sh);
    double a = 0.0;
    for (int p = 0; p < P; ++p)
        for (int64_t i = threadIdx.x; i < N; i += 256)
            a = fma(om[i + (int64_t)p * ldv], al[i + (int64_t)p * ldv], a);
    a = block_sum_256(a, sh);
    if (threadIdx.x == 0) {
        out[0] = s;
        out[1] = a;
    }
}
void launch_loglik_terms(hipStream_t s, const double* L, int64_t ldl, int64_t N, const double* om, const double* alpha,
                         int64_t ldv, int P, double* out)
{
    GPE_LAUNCH(k_loglik_terms, dim3(1), dim3(256), 0, s, L, ldl, N, om, alpha, ldv, P, out);
}

// var[m] = kvv[m] - sum_i Z[i,m]^2   (gp.hpp:621)
__global__ __launch_bounds__(256) void k_col_var(const double* __restrict__ Z, int64_t ldz, int64_t N, int64_t M,
                                                 const double* __restrict__ kvv, double* __restrict__ var)
{
    __shared__ double sh[4];
    for (int64_t m = blockIdx.x; m < M; m += gridDim.x) {
        const double* z = Z + m * ldz;
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < N; i += 256) {
            const double v = z[i];
            s = fma(v, v, s);
        }
        s = block_sum_256(s, sh);
        if (threadIdx.x == 0)
            var[m] = kvv[m] - s;
    }
}
void launch_col_var(hipStream_t s, const double* Z, int64_t ldz, int64_t N, int64_t M, const double* kvv, double* var)
{
    if (M <= 0)
        return;
    unsigned grid = (unsigned)(M < 4096 ? M : 4096);
    GPE_LAUNCH(k_col_var, dim3(grid), dim3(256), 0, s, Z, ldz, N, M, kvv, var);
}

// kta[m, p] = sum_i Ks[i, m] alpha[i, p]   (gp.hpp:615)
__global__ __launch_bounds__(256) void k_kta(const double* __restrict__ Ks, int64_t ldk, int64_t N, int64_t M,
                                             const double* __restrict__ alpha, int64_t lda, int P,
                                             double* __restrict__ kta, int64_t ldo)
{
    __shared__ double sh[4];
    for (int64_t m = blockIdx.x; m < M; m += gridDim.x) {
        const double* k = Ks + m * ldk;
        for (int p = 0; p < P; ++p) {
            const double* a = alpha + (int64_t)p * lda;
            double s = 0.0;
            for (int64_t i = threadIdx.x; i < N; i += 256)
                s = fma(k[i], a[i], s);
            s = block_sum_256(s, sh);
            if (threadIdx.x == 0)
                kta[m + (int64_t)p * ldo] = s;
        }
    }
}
void launch_kta(hipStream_t s, const double* Ks, int64_t ldk, int64_t N, int64_t M, const double* alpha, int64_t lda,
                int P, double* kta, int64_t ldo)
{
    if (M <= 0)
        return;
    unsigned grid = (unsigned)(M < 4096 ? M : 4096);
    GPE_LAUNCH(k_kta, dim3(grid), dim3(256), 0, s, Ks, ldk, N, M, alpha, lda, P, kta, ldo);
}

// ---- the same two reductions for the TRANSPOSED layout of the batched query path (points contiguous: Zt[m + i ldz]) ----
// partial[(seg * nout + q) * ldp + m] = sum over i in segment seg of  Zt[m, i]^2          (A == nullptr, nout = 1)
//                                                                   or  Zt[m, i] A[i, q]   (A: N x nout, lda)
// thread = point (coalesced along m), the N columns cut into gridDim.y segments; k_rows_finish adds the segments in order.
template <int NQ>
__global__ __launch_bounds__(256) void k_rows_partial_t(const double* __restrict__ Zt, int64_t ldz, int64_t N, int64_t M,
                                                        const double* __restrict__ A, int64_t lda, int nout,
                                                        double* __restrict__ partial, int64_t ldp)
{
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t seg = blockIdx.y, nseg = gridDim.y;
    const int64_t len = (N + nseg - 1) / nseg, i0 = seg * len, i1 = i0 + len < N ? i0 + len : N;
    const int64_t mc = m < M ? m : M - 1;
    for (int q0 = 0; q0 < nout; q0 += NQ) {
        double acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            acc[q] = 0.0;
        for (int64_t i = i0; i < i1; ++i) {
            const double v = Zt[mc + i * ldz];
            if (A) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    acc[q] = fma(v, A[i + (int64_t)(q0 + q < nout ? q0 + q : nout - 1) * lda], acc[q]);
            }
            else
                acc[0] = fma(v, v, acc[0]);
        }
        if (m < M)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q0 + q < nout)
                    partial[(seg * nout + q0 + q) * ldp + m] = acc[q];
    }
}
// out[m + q ldo] = (base ? base[m] - sum : sum) of the nseg partials, in order
__global__ void k_rows_finish_t(const double* __restrict__ partial, int64_t ldp, int nseg, int nout, int64_t M,
                                const double* __restrict__ base, double* __restrict__ out, int64_t ldo)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = blockIdx.y;
    if (m >= M)
        return;
    double s = 0.0;
    for (int seg = 0; seg < nseg; ++seg)
        s += partial[((int64_t)seg * nout + q) * ldp + m];
    out[m + (int64_t)q * ldo] = base ? base[m] - s : s;
}
// var[m] = kvv[m] - sum_i Zt[m, i]^2   (gp.hpp:621);  partial: nseg x ldp doubles of scratch (ldp >= M)
void launch_row_var_t(hipStream_t s, const double* Zt, int64_t ldz, int64_t N, int64_t M, const double* kvv, double* var,
                      double* partial, int64_t ldp, int nseg)
{
    if (M <= 0)
        return;
    GPE_LAUNCH((k_rows_partial_t<1>), dim3((unsigned)((M + 255) / 256), (unsigned)nseg), dim3(256), 0, s, Zt, ldz, N, M,
                       (const double*)nullptr, (int64_t)0, 1, partial, ldp);
    GPE_LAUNCH(k_rows_finish_t, dim3((unsigned)((M + 255) / 256), 1), dim3(256), 0, s, partial, ldp, nseg, 1, M, kvv, var, M);
}
// kta[m + p ldo] = sum_i Kst[m, i] alpha[i, p]   (gp.hpp:615);  partial: nseg x P x ldp doubles
void launch_kta_t(hipStream_t s, const double* Kst, int64_t ldk, int64_t N, int64_t M, const double* alpha, int64_t lda, int P,
                  double* kta, int64_t ldo, double* partial, int64_t ldp, int nseg)
{
    if (M <= 0 || P <= 0)
        return;
    GPE_LAUNCH((k_rows_partial_t<4>), dim3((unsigned)((M + 255) / 256), (unsigned)nseg), dim3(256), 0, s, Kst, ldk, N, M, alpha,
                       lda, P, partial, ldp);
    GPE_LAUNCH(k_rows_finish_t, dim3((unsigned)((M + 255) / 256), (unsigned)P), dim3(256), 0, s, partial, ldp, nseg, P, M,
                       (const double*)nullptr, kta, ldo);
}

// ---------------------------------------------------------------------------------------
// small utilities
// ---------------------------------------------------------------------------------------
__global__ void k_set_identity(double* __restrict__ A, int64_t lda, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = blockIdx.y;
    if (i < n)
        A[i + j * lda] = (i == j) ? 1.0 : 0.0;
}
void launch_set_identity(hipStream_t s, double* A, int64_t lda, int64_t n)
{
    if (n <= 0)
        return;
    GPE_LAUNCH(k_set_identity, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, s, A, lda, n);
}

__global__ void k_zero_upper(double* __restrict__ A, int64_t lda, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = blockIdx.y;
    if (i < n && i < j)
        A[i + j * lda] = 0.0;
}
void launch_zero_upper(hipStream_t s, double* A, int64_t lda, int64_t n)
{
    if (n <= 0)
        return;
    GPE_LAUNCH(k_zero_upper, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, s, A, lda, n);
}

// A[i][j] = A[j][i] for i < j (32x32 LDS-transposed tiles so both sides are coalesced)
__global__ __launch_bounds__(256) void k_symmetrize(double* __restrict__ A, int64_t lda, int64_t n)
{
    __shared__ double T[32][33];
    const int bi = blockIdx.x, bj = blockIdx.y; // source tile (rows bi, cols bj) with bi >= bj
    if (bi < bj)
        return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    for (int q = 0; q < 4; ++q) {
        const int64_t i = (int64_t)bi * 32 + tx, j = (int64_t)bj * 32 + ty + 8 * q;
        T[ty + 8 * q][tx] = (i < n && j < n) ? A[i + j * lda] : 0.0;
    }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        // destination element (row = bj*32 + tx, col = bi*32 + ty+8q) = source (col, row)
        const int64_t r = (int64_t)bj * 32 + tx, c = (int64_t)bi * 32 + ty + 8 * q;
        if (r < n && c < n && r < c)
            A[r + c * lda] = T[tx][ty + 8 * q];
    }
}
void launch_symmetrize_from_lower(hipStream_t s, double* A, int64_t lda, int64_t n)
{
    if (n <= 0)
        return;
    unsigned t = (unsigned)((n + 31) / 32);
    GPE_LAUNCH(k_symmetrize, dim3(t, t), dim3(256), 0, s, A, lda, n);
}

__global__ void k_copy2d(const double* __restrict__ src, int64_t lds_, double* __restrict__ dst, int64_t ldd,
                         int64_t rows, int64_t cols)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows)
        return;
    for (int64_t j = blockIdx.y; j < cols; j += gridDim.y)
        dst[i + j * ldd] = src[i + j * lds_];
}
void launch_copy2d(hipStream_t s, const double* src, int64_t lds_, double* dst, int64_t ldd, int64_t rows, int64_t cols)
{
    if (rows <= 0 || cols <= 0)
        return;
    unsigned gy = (unsigned)(cols < 65535 ? cols : 65535);
    GPE_LAUNCH(k_copy2d, dim3((unsigned)((rows + 255) / 256), gy), dim3(256), 0, s, src, lds_, dst, ldd, rows,
                       cols);
}

// add_sample tail (gp.hpp:596-597): L[n][n] = sqrt(K_nn - ||L[n][0:n]||^2); Lrow points at L[n][0]
__global__ __launch_bounds__(256) void k_append_diag(double* __restrict__ Lrow, int64_t ldl, int64_t n,
                                                     const double* __restrict__ knn, int* __restrict__ info)
{
    __shared__ double sh[4];
    double s = 0.0;
    for (int64_t k = threadIdx.x; k < n; k += 256) {
        const double v = Lrow[k * ldl];
        s = fma(v, v, s);
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) {
        const double d = knn[0] - s;
        if (!(d > 0.0) && *info == 0)
            *info = (int)(n + 1);
        Lrow[n * ldl] = sqrt(d);
    }
}
void launch_append_diag(hipStream_t s, double* Lrow, int64_t ldl, int64_t n, const double* knn, int* info)
{
    GPE_LAUNCH(k_append_diag, dim3(1), dim3(256), 0, s, Lrow, ldl, n, knn, info);
}
