// small.hip — the Bayesian-optimisation inner loop below ~256 samples, ONE launch per call, no copies.
//
//   add_sample  (src/limbo/model/gp.hpp:126-152 + :573-603 + :605-611)      k_small_add<P>
//   query of a handful of points (gp.hpp:159-191, :613-632)                   k_small_query
//
// At these sizes (BASELINE configs[4]: n = 10 -> 200; bayes_opt/boptimizer.hpp:148-161 calls add_sample once and
// query() thousands of times per iteration) the general path is all fixed cost: >= 10 launches, two staged
// host-to-device copies and a stream synchronisation per add_sample (105 us, one-point query 72 us — slower than
// the reference on one CPU core).  Here the whole call is one workgroup on one CU:
//
//   * the new sample travels as a kernel argument; obs_mean / the query points are read by the kernel straight from
//     pinned host memory, results and a sequence word are written straight back to pinned host memory, the host
//     spins on that word: no hipMemcpy, no event, no stream synchronisation;
//   * the strictly-lower 64 x 64 tiles of the old factor (<= 6 of them) are loaded ONCE into registers (lane = row,
//     wave w holds columns w, w+8, ..: 512 contiguous bytes per load) together with the <= 4 block inverses (LDS),
//     all loads issued before the first dependent instruction — the dependency chain then runs on registers and LDS;
//   * the forward substitution carries the new kernel column AND obs_mean together; the appended row enters
//     analytically (L_new = [[L, 0], [row, l_nn]]:  y_n = (b_n - row.y) / l_nn,  a_n = y_n / l_nn,
//     a = L^-T (y - row^T a_n)), so both sweeps use the OLD factor and its old block inverses: the new row, the new
//     diagonal entry and the updated block inverse are written at the end, off the chain;
//   * the backward sweep re-uses the same register tiles transposed: per column a 64-lane sum by DPP row
//     operations (no LDS traffic, fixed order: bitwise reproducible).
#include "dev.h"

#define NB 64
#define XS 65
#define SM_T 512                     // threads of the one workgroup
#define SM_W (SM_T / 64)             // waves
#define SM_Q (NB / SM_W)             // tile columns per wave per 64-column block
#define SM_NBLK 4                    // old factor: <= 4 row blocks (n <= 256)
#define SM_NT (SM_Q * (SM_NBLK * (SM_NBLK - 1) / 2)) // register tile elements per thread: 8 (1 + 2 + 3) = 48
#define SM_RMAX 4                    // right-hand sides carried together

// ---- 64-lane sum by DPP (VALU only) --------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// total over the 64 lanes, valid in lanes 48..63 (fixed order of additions)
static __device__ __forceinline__ double wave_sum_row3(double v)
{
    v += dpp_f64<0xB1, 0xF>(v);  // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xF>(v);  // quad_perm [2,3,0,1]
    v += dpp_f64<0x141, 0xF>(v); // row_half_mirror: the other quad of the 8
    v += dpp_f64<0x140, 0xF>(v); // row_mirror: the other half of the 16
    v += dpp_f64<0x142, 0xA>(v); // row_bcast15 into rows 1, 3
    v += dpp_f64<0x143, 0xC>(v); // row_bcast31 into rows 2, 3
    return v;
}
// sum over the whole workgroup, returned to every thread (red: SM_W doubles of LDS; two barriers)
static __device__ __forceinline__ double block_sum(double v, double* red)
{
    const double t = wave_sum_row3(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 63)
        red[threadIdx.x >> 6] = t;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int w = 1; w < SM_W; ++w)
        s += red[w];
    return s;
}

struct SmallCtx {
    const double* L;   // factor, column-major
    int64_t ld;
    const double* Xinv; // block inverses, Xt[k + 64 c] = (L_bb^-1)[c][k]
    int n;             // rows of the (old) factor
};

// LDS of the small kernels
struct SmallLds {
    double Xs[SM_NBLK][NB * XS];      // block inverses, Xs[b][c * XS + k] = (L_bb^-1)[c][k]
    double part[SM_W][SM_RMAX][NB];   // per-wave partial sums (also the second-stage partials)
    double v[SM_RMAX][SM_NBLK * NB];  // right-hand sides / solutions
    double w[SM_RMAX][NB];
    double x[GPE_MAX_THETA];
    double omn[SM_RMAX]; // obs_mean of the new sample
    double red[SM_W];
};

static __device__ __forceinline__ int tile_off(int j) { return SM_Q * (j * (j - 1) / 2); }

// all loads of the old factor, issued back to back: T[tile_off(j) + q] = L[64 j + lane][wave + 8 q], j = 1 .. nblk-1
static __device__ __forceinline__ void load_tiles(const SmallCtx& c, double (&T)[SM_NT], SmallLds& S)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nblk = (c.n + NB - 1) / NB;
#pragma unroll
    for (int j = 1; j < SM_NBLK; ++j) {
        const int r = NB * j + lane;
        const int rc = r < c.n ? r : (c.n > 0 ? c.n - 1 : 0);
        const double m = (j < nblk && r < c.n) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < SM_Q * j; ++q) {
            const int col = wv + SM_W * q;
            // unconditional load from a valid (clamped) address, masked by a multiplication (see solve.hip)
            T[tile_off(j) + q] = (j < nblk ? c.L[rc + (int64_t)col * c.ld] : 0.0) * m;
        }
    }
    for (int b = 0; b < nblk; ++b) {
        const double* Xt = c.Xinv + (int64_t)b * (NB * NB);
#pragma unroll
        for (int i = 0; i < NB * NB / SM_T; ++i) {
            const int e = threadIdx.x + SM_T * i;
            S.Xs[b][(e >> 6) * XS + (e & 63)] = Xt[e];
        }
    }
}

// S.v[rho][0:n] <- L^-1 S.v[rho][0:n] for rho in [0, R); entries n .. 64 nblk - 1 must be zero on entry
template <int R>
static __device__ __forceinline__ void small_fwd(const SmallCtx& c, const double (&T)[SM_NT], SmallLds& S)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nblk = (c.n + NB - 1) / NB;
#pragma unroll
    for (int j = 0; j < SM_NBLK; ++j) {
        if (j >= nblk)
            break;
        if (j > 0) {
            double acc[R];
#pragma unroll
            for (int rho = 0; rho < R; ++rho)
                acc[rho] = 0.0;
#pragma unroll
            for (int q = 0; q < SM_Q * j; ++q) {
                const int col = wv + SM_W * q;
#pragma unroll
                for (int rho = 0; rho < R; ++rho)
                    acc[rho] = fma(T[tile_off(j) + q], S.v[rho][col], acc[rho]);
            }
#pragma unroll
            for (int rho = 0; rho < R; ++rho)
                S.part[wv][rho][lane] = acc[rho];
        }
        __syncthreads();
        if (threadIdx.x < NB * R) {
            const int rho = threadIdx.x >> 6;
            double s = S.v[rho][NB * j + lane];
            if (j > 0) {
#pragma unroll
                for (int w = 0; w < SM_W; ++w)
                    s -= S.part[w][rho][lane];
            }
            S.w[rho][lane] = s;
        }
        __syncthreads();
        { // z = X_j w: thread (c = lane, k-slice = wave)
            const double* Xr = &S.Xs[j][lane * XS + SM_Q * wv];
#pragma unroll
            for (int rho = 0; rho < R; ++rho) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < SM_Q; ++k)
                    s = fma(Xr[k], S.w[rho][SM_Q * wv + k], s);
                S.part[wv][rho][lane] = s;
            }
        }
        __syncthreads();
        if (threadIdx.x < NB * R) {
            const int rho = threadIdx.x >> 6;
            double s = S.part[0][rho][lane];
#pragma unroll
            for (int w = 1; w < SM_W; ++w)
                s += S.part[w][rho][lane];
            S.v[rho][NB * j + lane] = s;
        }
        __syncthreads();
    }
}

// S.v[rho0 + rho][0:n] <- L^-T S.v[rho0 + rho][0:n], rho in [0, R)
template <int R>
static __device__ __forceinline__ void small_bwd(const SmallCtx& c, const double (&T)[SM_NT], SmallLds& S, int rho0)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nblk = (c.n + NB - 1) / NB;
#pragma unroll
    for (int j = SM_NBLK - 1; j >= 0; --j) {
        if (j >= nblk)
            continue;
        { // a_j = X_j^T w_j: thread (r = lane, k-slice = wave)
#pragma unroll
            for (int rho = 0; rho < R; ++rho) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < SM_Q; ++k)
                    s = fma(S.Xs[j][(SM_Q * wv + k) * XS + lane], S.v[rho0 + rho][NB * j + SM_Q * wv + k], s);
                S.part[wv][rho][lane] = s;
            }
        }
        __syncthreads();
        if (threadIdx.x < NB * R) {
            const int rho = threadIdx.x >> 6;
            double s = S.part[0][rho][lane];
#pragma unroll
            for (int w = 1; w < SM_W; ++w)
                s += S.part[w][rho][lane];
            S.v[rho0 + rho][NB * j + lane] = s;
        }
        __syncthreads();
        if (j > 0) { // v[col] -= sum_r L[64 j + r][col] a_j[r] for col < 64 j: the register tiles, transposed
            double a[R];
#pragma unroll
            for (int rho = 0; rho < R; ++rho)
                a[rho] = S.v[rho0 + rho][NB * j + lane];
#pragma unroll
            for (int q = 0; q < SM_Q * j; ++q) {
                const int col = wv + SM_W * q;
#pragma unroll
                for (int rho = 0; rho < R; ++rho) {
                    const double t = wave_sum_row3(T[tile_off(j) + q] * a[rho]);
                    if (lane == 63)
                        S.v[rho0 + rho][col] -= t; // (wave, q) owns this column: no other writer
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct SmallX {
    double v[GPE_MAX_THETA];
};

// The bodies of the three calls (`reload` = false: the factor's tiles T and the block inverses S.Xs are where an earlier
// call of the same workgroup left them — the resident-workgroup form of round 3, measured 2 us slower per call than a launch
// and removed in round 4: profiles/r03_small_server_latency.log; the kernels below always reload).
template <int P>
static __device__ __forceinline__ void small_add_body(const SmallAddArgs& g, const KParams& kp, const LamParams& lp,
                                                      const double* xnew, SmallLds& S, double (&T)[SM_NT], bool reload)
{
    const int tid = threadIdx.x;
    const int n = g.n;
    SmallCtx c{g.A, g.ld, g.Xinv, n};
    if (reload)
        load_tiles(c, T, S);
    // old diagonal (log-likelihood term), this thread's sample, obs_mean from pinned host memory
    const int ic = tid < n ? tid : 0;
    const double ldiag = (tid < n) ? g.A[ic + (int64_t)ic * g.ld] : 1.0;
    double om_i[P];
#pragma unroll
    for (int p = 0; p < P; ++p)
        om_i[p] = (tid <= n) ? g.om_host[tid + (int64_t)p * (n + 1)] : 0.0;
    if (tid < kp.Din)
        S.x[tid] = xnew[tid];
    if (tid == n) {
#pragma unroll
        for (int p = 0; p < P; ++p)
            S.omn[p] = om_i[p];
    }
    __syncthreads();
    if (tid < lp.k) { // projections Lambda^T x of the new sample (squared_exp_ard.hpp:142-146), as k_lambda_rows
        double f = 0.0;
        for (int d = 0; d < lp.D; ++d)
            f = fma(S.x[d], lp.A[d + tid * lp.D], f);
        S.x[lp.D + tid] = f;
    }
    __syncthreads();
    // k(x_i, x_new), no noise (gp.hpp:583-586 with i != n); k(x_new, x_new) + noise + 1e-8 (kernel.hpp:83)
    double kv = 0.0;
    if (tid < n) {
        double z = 0.0;
        for (int d = 0; d < kp.D; ++d) {
            const double q = (g.Xt[(int64_t)d * g.ldx + tid] - S.x[d]) * kp.inv_ell[d];
            z = fma(q, q, z);
        }
        kv = kfun(kp.kind, z, kp.sf2);
    }
    const double knn = kfun(kp.kind, 0.0, kp.sf2) + kp.diag_add;
    if (tid < SM_NBLK * NB) {
        S.v[0][tid] = kv;
#pragma unroll
        for (int p = 0; p < P; ++p)
            S.v[1 + p][tid] = tid < n ? om_i[p] : 0.0;
    }
    __syncthreads();
    small_fwd<P + 1>(c, T, S); // v[0] = new row of L (gp.hpp:591-594), v[1 + p] = (L^-1 obs_mean)[0:n]
    const double zk = tid < n ? S.v[0][tid] : 0.0;
    const double d2 = knn - block_sum(zk * zk, S.red); // gp.hpp:596
    const double lnn = sqrt(d2);                       // gp.hpp:597 (NaN when K is not positive definite, as there)
    double yn[P], an[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const double dot = block_sum(tid < n ? zk * S.v[1 + p][tid] : 0.0, S.red);
        const double omn = S.omn[p];
        yn[p] = (omn - dot) / lnn;
        an[p] = yn[p] / lnn;
    }
    __syncthreads();
    if (tid < n) {
#pragma unroll
        for (int p = 0; p < P; ++p)
            S.v[1 + p][tid] -= zk * an[p];
    }
    __syncthreads();
    small_bwd<P>(c, T, S, 1); // alpha[0:n] (gp.hpp:605-611 for the extended factor)
    // ---- everything below is off the dependency chain: results and the new state ----
    double s_oa = 0.0;
    if (tid < n) {
        g.A[n + (int64_t)tid * g.ld] = zk;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const double a = S.v[1 + p][tid];
            g.Al[tid + (int64_t)p * g.ld] = a;
            g.Om[tid + (int64_t)p * g.ld] = om_i[p];
            s_oa = fma(om_i[p], a, s_oa);
        }
    }
    if (tid == n) {
        g.A[n + (int64_t)n * g.ld] = lnn;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            g.Al[n + (int64_t)p * g.ld] = an[p];
            g.Om[n + (int64_t)p * g.ld] = om_i[p];
            s_oa = fma(om_i[p], an[p], s_oa);
        }
    }
    if (tid < kp.D)
        g.Xt[(int64_t)tid * g.ldx + n] = S.x[tid];
    { // block inverse of the block that gained row q = n mod 64
        const int jb = n >> 6, q = n & 63;
        double* Xt = g.Xinv + (int64_t)jb * (NB * NB);
        const double inv = 1.0 / lnn;
        if (q == 0) { // a new block: identity-padded
            for (int e = tid; e < NB * NB; e += SM_T)
                Xt[e] = ((e & 63) == (e >> 6)) ? ((e == 0) ? inv : 1.0) : 0.0;
        }
        else if (tid <= q) {
            // X_new[q][k] = -(1 / l_nn) sum_{c = k}^{q-1} L[n][64 jb + c] X[c][k],  X_new[q][q] = 1 / l_nn
            double s = 0.0;
            for (int cc = tid; cc < q; ++cc)
                s = fma(S.v[0][NB * jb + cc], S.Xs[jb][cc * XS + tid], s);
            Xt[tid + NB * q] = (tid == q) ? inv : -s * inv;
        }
    }
    const double s_ld = block_sum(tid < n ? log(ldiag) : (tid == n ? log(lnn) : 0.0), S.red);
    const double s_tot = block_sum(s_oa, S.red);
    if (tid == 0) {
        g.out[0] = s_ld;
        g.out[1] = s_tot;
        if (!(d2 > 0.0) && *g.info == 0)
            *g.info = n + 1;
    }
    __threadfence_system(); // device results before the word the host spins on; every thread's global stores are out
    __syncthreads();
    if (tid == 0) {
        __threadfence_system();
        __hip_atomic_store(g.seq, g.seq_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
template <int P>
__global__ __launch_bounds__(SM_T) void k_small_add(SmallAddArgs g, KParams kp, LamParams lp, SmallX xnew)
{
    __shared__ SmallLds S;
    double T[SM_NT];
    small_add_body<P>(g, kp, lp, xnew.v, S, T, true);
}

// ---------------------------------------------------------------------------------------------------------------------
// one workgroup per query point: k* -> k*^T alpha (gp.hpp:615) -> z = L^-1 k* (:620) -> k(v,v) - z.z (:621)
static __device__ __forceinline__ void small_query_body(const SmallQueryArgs& g, const KParams& kp, const LamParams& lp, int m,
                                                        SmallLds& S, double (&T)[SM_NT], bool reload)
{
    const int tid = threadIdx.x;
    const int n = g.n;
    SmallCtx c{g.L, g.ld, g.Xinv, n};
    if (g.want_var && reload)
        load_tiles(c, T, S);
    if (tid < g.D)
        S.x[tid] = g.xq_host[(int64_t)m * g.D + tid];
    __syncthreads();
    if (tid < lp.k) {
        double f = 0.0;
        for (int d = 0; d < lp.D; ++d)
            f = fma(S.x[d], lp.A[d + tid * lp.D], f);
        S.x[lp.D + tid] = f;
    }
    __syncthreads();
    double kv = 0.0;
    if (tid < n) {
        double z = 0.0;
        for (int d = 0; d < kp.D; ++d) {
            const double q = (g.Xt[(int64_t)d * g.ldx + tid] - S.x[d]) * kp.inv_ell[d];
            z = fma(q, q, z);
        }
        kv = kfun(kp.kind, z, kp.sf2);
    }
    if (g.want_kta) {
        for (int p = 0; p < g.P; ++p) {
            const double s = block_sum(tid < n ? kv * g.Al[tid + (int64_t)p * g.ld] : 0.0, S.red);
            if (tid == 0)
                g.kta_host[m + (int64_t)p * g.M] = s;
        }
    }
    if (g.want_var) {
        if (tid < SM_NBLK * NB)
            S.v[0][tid] = kv;
        __syncthreads();
        small_fwd<1>(c, T, S);
        const double z = tid < n ? S.v[0][tid] : 0.0;
        const double zz = block_sum(z * z, S.red);
        if (tid == 0)
            g.var_host[m] = kfun(kp.kind, 0.0, kp.sf2) - zz;
    }
    __syncthreads();
    if (tid == 0) {
        __threadfence_system();
        __hip_atomic_store(g.seq + m, g.seq_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(SM_T) void k_small_query(SmallQueryArgs g, KParams kp, LamParams lp)
{
    __shared__ SmallLds S;
    double T[SM_NT];
    small_query_body(g, kp, lp, (int)blockIdx.x, S, T, true);
}


// ---------------------------------------------------------------------------------------------------------------------
// recompute(., false) / _compute_alpha (gp.hpp:241-252, :605-611) below 256 samples: new obs_mean, same factor.
// alpha = L^-T L^-1 obs_mean and the two log-likelihood sums, one launch, obs_mean read from pinned host memory.
template <int P>
static __device__ __forceinline__ void small_alpha_body(const SmallAlphaArgs& g, SmallLds& S, double (&T)[SM_NT], bool reload)
{
    const int tid = threadIdx.x;
    const int n = g.n;
    SmallCtx c{g.L, g.ld, g.Xinv, n};
    if (reload)
        load_tiles(c, T, S);
    const int ic = tid < n ? tid : 0;
    const double ldiag = (tid < n) ? g.L[ic + (int64_t)ic * g.ld] : 1.0;
    double om_i[P];
#pragma unroll
    for (int p = 0; p < P; ++p)
        om_i[p] = (tid < n) ? g.om_src[tid + (int64_t)p * g.ldom] : 0.0;
    if (tid < SM_NBLK * NB) {
#pragma unroll
        for (int p = 0; p < P; ++p)
            S.v[p][tid] = om_i[p];
    }
    __syncthreads();
    small_fwd<P>(c, T, S);
    small_bwd<P>(c, T, S, 0);
    double s_oa = 0.0;
    if (tid < n) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const double a = S.v[p][tid];
            g.Al[tid + (int64_t)p * g.ld] = a;
            if (g.Om)
                g.Om[tid + (int64_t)p * g.ld] = om_i[p];
            s_oa = fma(om_i[p], a, s_oa);
        }
    }
    const double s_ld = block_sum(tid < n ? log(ldiag) : 0.0, S.red);
    const double s_tot = block_sum(s_oa, S.red);
    if (tid == 0) {
        g.out[0] = s_ld;
        g.out[1] = s_tot;
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        __threadfence_system();
        __hip_atomic_store(g.seq, g.seq_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int P>
__global__ __launch_bounds__(SM_T) void k_small_alpha(SmallAlphaArgs g)
{
    __shared__ SmallLds S;
    double T[SM_NT];
    small_alpha_body<P>(g, S, T, true);
}

// ---------------------------------------------------------------------------------------------------------------------
// ---- host side ------------------------------------------------------------------------------------------------------
int small_max_n() { return SM_NBLK * NB; }

void launch_small_add(hipStream_t s, const SmallAddArgs& g, int P, const KParams& kp, const LamParams& lp, const double* x)
{
    SmallX xv;
    for (int d = 0; d < GPE_MAX_THETA; ++d)
        xv.v[d] = d < kp.Din ? x[d] : 0.0;
    if (P == 1)
        GPE_LAUNCH(k_small_add<1>, dim3(1), dim3(SM_T), 0, s, g, kp, lp, xv);
    else if (P == 2)
        GPE_LAUNCH(k_small_add<2>, dim3(1), dim3(SM_T), 0, s, g, kp, lp, xv);
    else
        GPE_LAUNCH(k_small_add<3>, dim3(1), dim3(SM_T), 0, s, g, kp, lp, xv);
}

void launch_small_query(hipStream_t s, const SmallQueryArgs& g, const KParams& kp, const LamParams& lp)
{
    GPE_LAUNCH(k_small_query, dim3((unsigned)g.M), dim3(SM_T), 0, s, g, kp, lp);
}

void launch_small_alpha(hipStream_t s, const SmallAlphaArgs& g, int P)
{
    if (P == 1)
        GPE_LAUNCH(k_small_alpha<1>, dim3(1), dim3(SM_T), 0, s, g);
    else if (P == 2)
        GPE_LAUNCH(k_small_alpha<2>, dim3(1), dim3(SM_T), 0, s, g);
    else if (P == 3)
        GPE_LAUNCH(k_small_alpha<3>, dim3(1), dim3(SM_T), 0, s, g);
    else
        GPE_LAUNCH(k_small_alpha<4>, dim3(1), dim3(SM_T), 0, s, g);
}
