// dev.h — internal declarations shared by the HIP translation units of libgpengine.so.
// gfx950 (MI355X, CDNA4) only: wave = 64 lanes, v_mfma_f64_16x16x4_f64, 160 KiB LDS/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#define GPE_MAX_THETA 64
#define GPE_MAX_P 8 // outputs handled per solve launch

// kernel-functor parameters, pre-digested on the host (kernel/kernel.hpp:116-123 and the
// per-kernel set_params: squared_exp_ard.hpp:96-105, matern_five_halves.hpp:97-102)
struct KParams {
    int kind;  // gpe_kernel_kind
    int D;     // rows of the SoA sample matrix that enter the distance: input dimension (+ k projection rows, below)
    int Din;   // input dimension
    int k_lam; // SE-ARD with M = Lambda Lambda^T + diag(ell^-2), Lambda D x k (squared_exp_ard.hpp:109-126,142-146):
               // rows Din .. Din+k-1 of the sample matrix hold the projections Lambda^T x (inv_ell = 1 there), so
               // (x1-x2)^T M (x1-x2) is the same sum of squares over D = Din + k rows
    double sf2;                 // sigma_f^2 = exp(2 p_last)
    double inv_l;               // 1/l (isotropic kernels)
    double diag_add;            // noise + 1e-8 (kernel.hpp:83)
    double noise;               // sigma_n^2
    double inv_ell[GPE_MAX_THETA]; // 1/ell_d (SE-ARD)
};

typedef double d4_t __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ d4_t mfma_f64(double a, double b, d4_t c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---- batched launches: G independent GPs of the same shape stepped by ONE launch per stage ------------------------
// (multi_gp.hpp:124-126 / parallel_repeater.hpp:86-105 as gpe_batch_compute runs them: config 4, 64 x N = 2048.)
// Every GP has its own buffers, allocated separately.  The launch sequence is built once, from GP 0's pointers, with
// gridDim.z = G; workgroup z translates each pointer argument from GP 0's buffer into GP z's buffer of the same class
// (wave-uniform: a handful of scalar compares per pointer).  The per-GP kernel parameters travel in the same table.
#define GPE_BT_CLS 13
#define GPE_BT_MAXG 64
struct BatchTab {
    int G;
    int ncls;
    const char* base0[GPE_BT_CLS];              // GP 0's buffers ...
    unsigned long long size[GPE_BT_CLS];         // ... and their sizes in bytes
    const char* base[GPE_BT_CLS][GPE_BT_MAXG];   // the same buffer of every GP
    KParams kp[GPE_BT_MAXG];
};
template <class T>
static __device__ __forceinline__ T* bt_rebase(const BatchTab* __restrict__ bt, int g, T* p)
{
    const char* q = (const char*)p;
#pragma unroll
    for (int c = 0; c < GPE_BT_CLS; ++c) {
        const unsigned long long off = (unsigned long long)(q - bt->base0[c]);
        if (off < bt->size[c])
            return (T*)(bt->base[c][g] + off);
    }
    return p;
}
#define BT_REBASE(bt, p)                          \
    do {                                          \
        if (bt)                                   \
            p = bt_rebase(bt, (int)blockIdx.z, p); \
    } while (0)
// what the launch wrappers of this thread add to every launch (set by gpe_batch_compute around the enqueue)
struct BatchLaunch {
    const BatchTab* bt = nullptr; // device copy
    int G = 1;
};
extern thread_local BatchLaunch g_batch;

// the kernel functors' value from z = sum_d ((x1_d - x2_d) / ell_d)^2 (kbuild.hip, small.hip)
static __device__ __forceinline__ double kfun(int kind, double z, double sf2)
{
    // z = sum_d ((x1_d - x2_d) / ell_d)^2   (isotropic kernels: ell_d = l for every d)
    switch (kind) {
    case 0: // SquaredExpARD::kernel, squared_exp_ard.hpp:148-150
    case 3: // Exp::kernel, exp.hpp:97-102
        return sf2 * exp(-0.5 * z);
    case 1: { // MaternFiveHalves::kernel, matern_five_halves.hpp:104-113
        double r = sqrt(z);
        double term1 = 2.23606797749978969641 * r; // sqrt(5) d / l
        double term2 = (5.0 / 3.0) * z;            // 5 d^2 / (3 l^2)
        return sf2 * (1.0 + term1 + term2) * exp(-term1);
    }
    default: { // MaternThreeHalves::kernel, matern_three_halves.hpp:101-107
        double term = 1.73205080756887729353 * sqrt(z);
        return sf2 * (1.0 + term) * exp(-term);
    }
    }
}

// ---- workgroup -> unknown block of a one-launch data-flow sweep (solve.hip, solve_mp.hip) ------------------
// Two requirements meet here.  (1) Progress without assuming that every workgroup is resident: a workgroup may
// only ever wait for workgroups with a LOWER blockIdx.x — the dispatcher hands workgroups out in that order, so
// whatever a resident workgroup polls for is resident or finished, however many CUs other streams (other handles'
// sweeps, 147 KB-LDS GEMMs) hold.  (2) Consecutive blocks of the chain on one XCD (hops served by that XCD's L2).
// The dispatcher deals blockIdx.x round-robin over the 8 XCDs, so a grid of nblk workgroups cannot have both.
// The grid therefore has 8 workgroups per chain position s = blockIdx.x / 8 (s-th block to be solved: s going up
// for the forward sweep, nblk-1-s for the backward one); only the one that sits on the XCD owning that block's
// contiguous range of the chain works, the other seven return at once (-1).
#define GPE_FLOW_SPIN_LIMIT (1 << 22) // bounded poll: ~a second; raises *err, and the host re-runs block by block
// (bx of gx: the workgroup's place in ITS chain's grid — blockIdx.x of gridDim.x for a single GP; a batched sweep interleaves its
// members' chains in one 1-D grid, batch.hpp / solve.hip)
static __device__ __forceinline__ int64_t flow_block_of_at(int64_t nblk, bool backward, int64_t bx, int64_t gx)
{
    if (gx == nblk) // plain form (GPE_FLOW_XCD=0, batched sweeps): chain position = dispatch position, no XCD locality
        return backward ? nblk - 1 - bx : bx;
    const int64_t s = bx >> 3;
    const int64_t x = bx & 7;
    const int64_t j = backward ? nblk - 1 - s : s;
    const int64_t q = nblk / 8, r = nblk % 8, split = r * (q + 1);
    const int64_t owner = j < split ? j / (q + 1) : r + (j - split) / (q > 0 ? q : 1);
    return x == owner ? j : -1;
}
static __device__ __forceinline__ int64_t flow_block_of(int64_t nblk, bool backward)
{
    return flow_block_of_at(nblk, backward, (int64_t)blockIdx.x, (int64_t)gridDim.x);
}
unsigned flow_grid(int64_t nblk); // 8 nblk (XCD-local chains) or nblk (GPE_FLOW_XCD=0)
#define GPE_FLOW_GRID(nblk) flow_grid(nblk)

// ---- launch tracing (GPE_TRACE=1 / gpe_trace(1)) ------------------------------------------------------------------
// rocprofv3's kernel trace delays every dispatch that carries its own completion event by ~100 us, so the PRODUCTION
// schedule of the factorisation (look-ahead on two streams, the fused next-panel update + diagonal block) cannot be traced
// with it.  With tracing on, every launch of the library carries a start and a stop event of its own instead
// (hipExtLaunchKernelGGL: the dispatch packet's own timestamps, no marker packets, the mechanism the look-ahead already
// uses for ordering), and gpe_trace_dump() writes (kernel, stream, start, end) relative to the first launch.
bool gpe_trace_on();
hipEvent_t gpe_trace_event();
void gpe_trace_add(const char* name, hipStream_t s, hipEvent_t e0, hipEvent_t e1, dim3 grid, dim3 block);
#define GPE_LAUNCH_NAMED(name, kern, grid, block, shmem, stream, ...)                                      \
    do {                                                                                                   \
        if (gpe_trace_on()) {                                                                              \
            hipEvent_t e0_ = gpe_trace_event(), e1_ = gpe_trace_event();                                   \
            hipExtLaunchKernelGGL(kern, grid, block, shmem, stream, e0_, e1_, 0, __VA_ARGS__);             \
            gpe_trace_add(name, stream, e0_, e1_, grid, block);                                            \
        }                                                                                                  \
        else                                                                                               \
            hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);                             \
    } while (0)
#define GPE_LAUNCH(kern, grid, block, shmem, stream, ...) GPE_LAUNCH_NAMED(#kern, kern, grid, block, shmem, stream, __VA_ARGS__)
// the same for a launch that signals `stop` (an event of the caller's: the look-ahead's ordering) through its dispatch
#define GPE_LAUNCH_STOP(name, kern, grid, block, shmem, stream, stop, ...)                                 \
    do {                                                                                                   \
        if (gpe_trace_on()) { /* (the caller's event has no timestamps: own events + a marker packet for `stop`) */ \
            hipEvent_t e0_ = gpe_trace_event(), e1_ = gpe_trace_event();                                   \
            hipExtLaunchKernelGGL(kern, grid, block, shmem, stream, e0_, e1_, 0, __VA_ARGS__);             \
            hipEventRecord(stop, stream);                                                                  \
            gpe_trace_add(name, stream, e0_, e1_, grid, block);                                            \
        }                                                                                                  \
        else                                                                                               \
            hipExtLaunchKernelGGL(kern, grid, block, shmem, stream, nullptr, stop, 0, __VA_ARGS__);        \
    } while (0)

// ---- kernel-matrix build (kbuild.hip) ----------------------------------------------
// Xt: SoA, D x ldx (sample index contiguous).  Writes the LOWER triangle (incl. diagonal,
// + diag_add) of K into A (col-major, lda).  gp.hpp:556-558 + kernel.hpp:81-84.
// tail (optional): what launch_cols_to_rows does, by extra workgroups of the same launch (kbuild.hip); returns whether it was
// taken (the wide build kernel, single-GP launches) — otherwise the caller launches it itself
struct BuildRowsTail {
    const double* V; // N x P (ld ldv)
    int64_t ldv;
    int P;
    double* Arows;   // A + N: row p of column i at Arows[p + i * lda]
    double* sent;    // optional, as for launch_cols_to_rows
    int64_t first;   // (set by the launcher)
};
bool launch_build_K(hipStream_t s, const double* Xt, int64_t ldx, int64_t N, const KParams& kp,
                    double* A, int64_t lda, const BuildRowsTail* tail = nullptr);
// full symmetric K (tests / gpe_get_K)
void launch_build_K_full(hipStream_t s, const double* Xt, int64_t ldx, int64_t N, const KParams& kp,
                         double* A, int64_t lda);
// cross kernel Ks[i, m] = k(x_i, q_m), no noise (gp.hpp:626-632); Qt SoA D x ldq; Ks col-major N x M
void launch_build_Ks(hipStream_t s, const double* Xt, int64_t ldx, int64_t N, const double* Qt, int64_t ldq,
                     int64_t M, const KParams& kp, double* Ks, int64_t ldk);
// kvv[m] = k(q_m, q_m)
void launch_kvv(hipStream_t s, const double* Qt, int64_t ldq, int64_t M, const KParams& kp, double* kvv);
// row-major (n x D) -> SoA (D x ld), writing columns [col0, col0+n)
void launch_transpose_x(hipStream_t s, const double* Xrm, int64_t n, int D, double* Xt, int64_t ld, int64_t col0);
// Lambda (D x k, column-major as in the parameter vector: squared_exp_ard.hpp:100-102)
struct LamParams {
    int D, k;
    double A[GPE_MAX_THETA];
};
// rows D .. D+k-1 of Xt, columns [col0, col0+n):  Xt[(D+j) ld + i] = sum_d A[d + j D] Xt[d ld + i]
void launch_lambda_rows(hipStream_t s, double* Xt, int64_t ld, int64_t col0, int64_t n, const LamParams& lp);

// ---- Cholesky diagonal block (potrf.hip) --------------------------------------------
// factor the jb x jb (jb <= 64) diagonal block at A (in place, lower) and write the transposed
// inverse Xt[k + 64 c] = (L11^-1)[c][k] (64 x 64, identity-padded for jb < 64).  info: first bad
// pivot (1-based, global index = goff + j + 1), written only if *info == 0.
// half_form (jb == 64 only): only the inverses of the two 32 x 32 diagonal half-blocks are produced
// (round 1; the data-flow block kernel of diag_flow.h leaves ALL of the inverse and ignores the flag).
void launch_diag(hipStream_t s, double* A, int64_t lda, int jb, double* Xt, int* info, int64_t goff, int half_form);
// one fused 64-column step below the factored diagonal block at (j0, j0): L21 = A21 X^T for rows
// j0+64 .. M-1, the updates of the next `nt` 64-column blocks of the outer panel, and (do_next) the
// factorisation + inversion of the next diagonal block (-> Xt_next).  Full 64-blocks only.
typedef unsigned long long gpe_epoch_t; // the value of a hand-over flag word: the serial number of the launch that set it
// Hs: nt scratch tiles (64 x 64 each) for the L of the first nt row blocks; launch_head_copy moves
// the tiles of the nf fused steps of the panel at p0 (nt0 = tiles of its first step) into A.
// dnext / dfirst (>= 0 to enable): see k_panel_step — pieces of the next outer panel's first diagonal block
void launch_panel_step(hipStream_t s, double* A, int64_t lda, int64_t j0, int64_t M, int nt, const double* Xt_cur,
                       double* Xt_next, int do_next, int* info, double* Hs, int64_t dnext, int64_t dfirst, int dinit,
                       double* Dacc, gpe_epoch_t* hflag);
// all steps of the 256-column outer panel at p0 in one launch (potrf.hip: k_panel256); Xt = inverse of the diagonal block at
// p0 (the next three follow at + 4096 each and are written), dnext / Dacc as above
void launch_panel256(hipStream_t s, double* A, int64_t lda, int64_t p0, int64_t M, double* Xt, int* info, int64_t dnext,
                     double* Dacc, double* S22, double* S22_next, hipEvent_t stop = nullptr);
// nt tile columns of a panel of nb row strips in one data-flow launch (potrf.hip: k_tail) — the closing columns of the
// factorisation (nb = nt [+ 1 for the right-hand-side rows]) or a tall head panel (nb > nt); a buffer = the polled quarters
// of nt block inverses, then a slot per tile
#define GPE_TAIL_MAX 8192 // (upper bound of the setting; the default: engine.hip)
// Data-flow launches (k_tail, k_panel256, the one-launch sweeps) wait INSIDE the launch for lower-numbered workgroups of the same
// launch.  That is deadlock-free for one such launch at a time — its lowest unfinished workgroup is always resident — but not
// for two from different streams: each XCD dispatches its share of every launch in order, and the CUs of the XCD that the
// lowest unfinished workgroup of launch A belongs to can all be held by waiting workgroups of launch B and vice versa (round
// 4: four handles evaluated from four host threads ran into the bounded polls and the re-run path, 1 evaluation/s instead of
// 1300; profiles/r04_concurrent_flow_launches.log).  So a data-flow launch from one stream waits for the previous one on the
// device when that one went to another stream (an event recorded at that stream's end at that moment; nothing otherwise).  engine.hip.
void flow_gate_enter(hipStream_t s);
void flow_gate_leave(hipStream_t s);
void flow_gate_forget(hipStream_t s);
struct FlowGate {
    hipStream_t s;
    bool on;
    explicit FlowGate(hipStream_t s_, bool engage = true) : s(s_), on(engage)
    {
        if (on)
            flow_gate_enter(s);
    }
    ~FlowGate()
    {
        if (on)
            flow_gate_leave(s);
    }
    FlowGate(const FlowGate&) = delete;
    FlowGate& operator=(const FlowGate&) = delete;
};
int debug_tail_order(int nt, int nb, int lag, int pair); // potrf.hip (host only)
void debug_chain_split(int wave, int* units10, int* cols); // potrf.hip (host only): the chain workgroup's work split
static inline int64_t tail_tiles(int64_t nt, int64_t nb) { return nt * nb - nt * (nt - 1) / 2; }
static inline int64_t tail_buf_doubles(int64_t nt, int64_t nb) { return nt * 3072 + tail_tiles(nt, nb) * 4096; }
// gen (optional): the launch generates its tiles of K from the samples (and obs_mean's rows from Om) instead of reading A,
// and pre-fills the sweep's sentinel in Al for its columns — no kernel-matrix build in front of it
struct TailGen {
    const double* Xg; // SoA samples
    int64_t ldx, Ns;
    const double* Om;
    int64_t ldom;
    double* Al; // may be null
    int64_t ldal;
    int P;
    const KParams* kp; // host copy (single launches pass it by value; batched ones read the batch table)
};
int ragged_split(int64_t k, int64_t scratch_doubles, int* kc_out); // potrf.hip (host only)
bool launch_ragged_finish(hipStream_t s, double* C, int64_t ldc, const double* A, int64_t ld, int64_t jb, int64_t P, int64_t k,
                          double* scratch, int64_t scratch_doubles, double* Xt, int* info, int64_t goff); // potrf.hip: update + factor + solve
bool launch_ragged_update(hipStream_t s, double* C, int64_t ldc, const double* A, int64_t ld, int64_t m, int64_t n, int64_t k,
                          double* scratch, int64_t scratch_doubles); // potrf.hip: a ragged order's last block behind k_tail
void launch_tail(hipStream_t s, double* A, int64_t lda, int64_t t0, int64_t t1, int64_t N64, int64_t M, double* Xt_all, int* info,
                 double* buf_cur, double* buf_next, const TailGen* gen = nullptr);
// S22 / S22_next: the polled hand-over buffers (block inverses + head tiles), 33,792 doubles each, holding the all-ones
// pattern when the launch starts (the launch arms S22_next)
#define GPE_S22_TILE 20 // the two buffers sit in tiles 20..28 of either half of gpe_ctx::dHead
#define GPE_S22_TILES 9
// the device block behind gpe_ctx::dHead: 32 + 32 head tiles (by panel parity), the scratch sum of the next diagonal
// block, and one tile's worth of hand-over flag words (hflag: one unsigned per head tile, same indexing as the tiles)
#define GPE_HEAD_TILES 66
void launch_head_copy(hipStream_t s, double* A, int64_t lda, int64_t p0, int nt0, int nf, const double* H);
// inverses of diagonal blocks b0 .. b0+nblocks-1 of an existing factor L (order N) into
// Xt_all + 4096 b
void launch_diag_inv(hipStream_t s, const double* L, int64_t ldl, int64_t N, int64_t b0, int64_t nblocks,
                     double* Xt_all);

// ---- fp64 MFMA GEMM update (gemm.hip) ----------------------------------------------
// C[m x n] -= opA[m x k] * opB[n x k]^T
//   a_kmajor = 0: opA(i,kk) = A[i + kk*lda]   (row index contiguous)
//   a_kmajor = 1: opA(i,kk) = A[kk + i*lda]   (k contiguous)          likewise for B.
// tri: 0 = all tiles; 1 = skip tiles entirely above the diagonal, where the diagonal is
//      (global row grow0 + i) == (global col gcol0 + j).
// ktri: 1 = per-tile k range starts at max(tile row0, tile col0) + koff (LAUUM: X^T X with X lower)
struct GemmArgs {
    double* C; int64_t ldc;
    const double* A; int64_t lda; int a_kmajor;
    const double* B; int64_t ldb; int b_kmajor;
    int64_t m, n, k;
    int tri; int64_t grow0, gcol0;
    int ktri;
    int overwrite; // 0: C -= A*B^T; 1: C = +A*B^T (no read of C); 2: C += A*B^T
    int fold_len;  // set by launch_gemm_sub (tri): live tiles per folded tile-column pair
    int total;     // set by launch_gemm_sub: logical workgroups (glds kernel)
    int grid_limit; // > 0: at most this many physical workgroups (they loop) — leaves CUs to another stream
    int tile;      // 0: pick by problem size; 128 / 64 / 32: force the 128x128 / 64x64 / 32x64 tile
    int rhs_rows;  // the LAST rhs_rows of the m rows of C / A are right-hand-side rows (engine.hip): the direct-to-LDS
                   // kernels update them with plain FMAs instead of a tile row (gemm_glds64.h: gemm_rhs_rows); inside those
                   // kernels m counts the main rows only
    const int* tile_map; // set by launch_gemm_sub (tri, 128 x 128 direct-to-LDS kernel): workgroup -> tile (ti | tj << 16, or -1: none)
                         // in an order that keeps the operand panels of an XCD's resident workgroups few (gemm.hip: tri_tile_map)
    void* stop_event; // host side only: hipEvent_t completed by this launch (null: none)
    const BatchTab* bt; // batched launch (gridDim.z GPs): set by the launch wrapper, null otherwise
};
static __device__ __forceinline__ void gemm_rebase(GemmArgs& g)
{
    if (g.bt) {
        const int z = (int)blockIdx.z;
        g.C = bt_rebase(g.bt, z, g.C);
        g.A = bt_rebase(g.bt, z, g.A);
        g.B = bt_rebase(g.bt, z, g.B);
    }
}
void launch_gemm_sub(hipStream_t s, const GemmArgs& g);
int debug_tri_tile_map(int64_t m, int64_t n, int64_t grow0, int64_t gcol0, int* out, int cap); // gemm.hip (host only)
// the next-panel update g (as for launch_gemm_sub: C = A[pe:, pe:pe2], k = pe - p0, tri) and, in the same launch, the
// update + factorisation + half-inversion of the next diagonal block A[pe:pe+64, pe:pe+64] (-> Xt_next)
// (Dacc: sum of the pieces the panel steps already formed, subtracted as well; p0 == pe: no products here)
void launch_upd_fused(hipStream_t s, const GemmArgs& g, double* A, int64_t lda, int64_t p0, int64_t pe, double* Xt_next,
                      int* info, const double* Dacc);
double gemm_flops(const GemmArgs& g);

// ---- K^-1 by recursion (inv2.hip): lists of 128 x 128 x k products and of tile folds ------------------------------
// One launch = a LIST of independent tile products C = (+/-) A B^T, A and B contiguous along their non-k index (the operand
// form of k_gemm_glds), each with its own k; persistent workgroups, each with a host-built share of the list (longest first).
// A product whose k range was cut into chunks leaves chunk c > 0 in a partial buffer; a fold list adds them up in fixed
// order (bitwise reproducible) and, where asked, also writes the tile transposed.
struct GemmItem {
    const double* A; // tile row origin at the first k of the chunk: A[r + kk * ld]
    const double* B; // tile column origin likewise: B[c + kk * ld]
    double* C;       // where THIS chunk's product goes (its first mr rows x nc columns: a ragged last tile stores nothing else); never
                     // read by the product: the tile itself for chunk 0 (and for a whole k range), partial buffer c for chunk c > 0
    double* T;       // optional: the finished tile transposed as well, T[c + r * ld] = D[r + c * ld]
    double* D;       // a cut k range: the tile (= chunk 0's C)
    const double* P1; // ... and its place in partial buffer 1 (buffer c: P1 + (c - 1) * pstride)
    int32_t k;       // multiple of 16
    int32_t flags;   // bit 0: the product enters negated; bits 8..15: mr, bits 16..23: nc — valid rows / columns (1 .. tile edge)
    int32_t slot, nch; // nch > 1: one of the nch chunks of a cut k range; `slot` = the tile's counter word.  Every chunk stores its
                     // product write-through and counts; the one that counts LAST adds them up, D + P1 + P2 + .. in that order
                     // whoever it is (bitwise reproducible), stores the tile and its transposed copy: nobody waits for anybody
};
#define GEMM_ITEM_NEG 1
// (batched launches, g_batch: the lists hold member 0's pointers, workgroup z rebases them — dev.h: bt_rebase; member z counts in
// the words z * nslots ..)
void launch_gemm_items(hipStream_t s, const GemmItem* items, const int32_t* bin_start, int nbins, int64_t ld, int tile, int* counters,
                       int nslots, int64_t pstride); // tile: 128 or 64
// the plan of one inversion (inv2.hip): host-built once per (N, ld, buffers), resident on the device
struct Inv2Plan;
Inv2Plan* inv2_plan_get(Inv2Plan* old, int64_t N, int64_t ld, const double* L, double* U, double* Kinv, double* S, int64_t pstride,
                        int members = 1, bool* rebuilt = nullptr); // members: GPs stepped by the launches (a batched sequence: another
                                                                   // plan from four on); rebuilt: the caller zero-fills U and S then
void inv2_plan_free(Inv2Plan* p);
bool inv2_supported(int64_t N);
int inv2_partials();                // N x N partial buffers behind the W / T-form buffer in S (S holds 1 + this many)
double inv2_flops(const Inv2Plan* p); // algorithmic flops of the products (2 N^3 / 3 less the leaves)
void inv2_run(hipStream_t s, Inv2Plan* p, const double* Xt_all, int part = 0); // enqueues the launches (part: inv2.hip)
int inv2_debug_plan(int64_t N, int64_t ld, int nbins, int load_pct, int64_t* out, int64_t cap_rows); // host only (tests)

// ---- vector solves, reductions (solve.hip) -----------------------------------------
// one full triangular sweep over P <= GPE_MAX_P right-hand sides (ceil(N/64) launches):
// trans = 0: out = L^-1 w (top down);  trans = 1: out = L^-T w (bottom up).  w is destroyed.
// Xt_all: inverses of the 64 x 64 diagonal blocks (launch_diag / launch_diag_inv).
void launch_trsv_sweep(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all, double* w,
                       double* out, int64_t ldw, int P, int trans);
// the whole backward sweep a = L^-T y in one data-flow launch for nblk = ceil(N/64) <= 256 (the caller
// checks).  y[i * ysi + p * ysp]; err: set to 1 if a hand-off timed out; prefilled: `a` already holds
// the all-ones sentinel; part (optional, 2 nblk doubles): per-block partial sums of log L_ii and
// om . a (part_acc: add to the second half instead of overwriting)
void launch_trsv_bwd_flow(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all, const double* y,
                          int64_t ysi, int64_t ysp, double* a, int64_t ldw, int P, int* err, int prefilled, const double* om,
                          int64_t ldom, double* part, int part_acc);
// the same for ONE right-hand side with a_j = g_j - M_j a_{j+1}: the hop is one matrix-vector product (sweep2.hip, round 6)
void launch_trsv_bwd_m(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all, const double* y, int64_t ysi, double* a,
                       int* err, int prefilled, const double* om, double* part);
// the whole forward sweep y = L^-1 b in one data-flow launch, nblk <= 256 (the caller checks); b: N x P (ldb),
// y: N x P (ldy), must not alias b; err as above
void launch_trsv_fwd_flow(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* Xt_all, const double* b,
                          int64_t ldb, double* y, int64_t ldy, int P, int* err);
// Arows[p + i*lda] = V[i + p*ldv] (P rows appended under the matrix) and back
void launch_cols_to_rows(hipStream_t s, const double* V, int64_t ldv, int64_t N, int P, double* Arows, int64_t lda,
                         double* sent = nullptr);
void launch_rows_to_cols(hipStream_t s, const double* Arows, int64_t lda, int64_t N, int P, double* V, int64_t ldv);
// out[0] = sum_i log L_ii ; out[1] = sum_{i,p} obs_mean * alpha   (gp.hpp:274-277)
void launch_loglik_terms(hipStream_t s, const double* L, int64_t ldl, int64_t N, const double* om, const double* alpha,
                         int64_t ldv, int P, double* out);
// colsq[m] = sum_i Z[i,m]^2 ; var[m] = kvv[m] - colsq[m]
void launch_col_var(hipStream_t s, const double* Z, int64_t ldz, int64_t N, int64_t M, const double* kvv, double* var);
// kta[m, p] = sum_i Ks[i, m] alpha[i, p]
void launch_kta(hipStream_t s, const double* Ks, int64_t ldk, int64_t N, int64_t M, const double* alpha, int64_t lda,
                int P, double* kta, int64_t ldo);
// the same two for the transposed layout of the batched query path (points contiguous, Zt[m + i ldz]); partial: scratch of
// nseg x (P or 1) x ldp doubles, the segments are added in order (bitwise reproducible)
void launch_row_var_t(hipStream_t s, const double* Zt, int64_t ldz, int64_t N, int64_t M, const double* kvv, double* var,
                      double* partial, int64_t ldp, int nseg);
void launch_kta_t(hipStream_t s, const double* Kst, int64_t ldk, int64_t N, int64_t M, const double* alpha, int64_t lda, int P,
                  double* kta, int64_t ldo, double* partial, int64_t ldp, int nseg);
// misc
void launch_set_identity(hipStream_t s, double* A, int64_t lda, int64_t n);
// inv.hip: Out[o0:o0+pw, o0:o0+pw] = inv(L[o0:o0+pw, o0:o0+pw]) (full square) for every outer panel of nbo <= 256
// columns, from the 64 x 64 block inverses Xt_all
// OutT (optional): the same blocks transposed
void launch_inv_panels(hipStream_t s, const double* L, int64_t ld, int64_t N, int nbo, const double* Xt_all, double* Out,
                       int64_t ldo, double* OutT, int64_t ldt);
// A[0 : rows, 0 : cols] = 0 (column-major, lda) — a kernel, not hipMemsetAsync: it takes part in batched launches
void launch_zero2d(hipStream_t s, double* A, int64_t lda, int64_t rows, int64_t cols);
void launch_zero_upper(hipStream_t s, double* A, int64_t lda, int64_t n);
void launch_symmetrize_from_lower(hipStream_t s, double* A, int64_t lda, int64_t n);
void launch_copy2d(hipStream_t s, const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols);
// add_sample tail: L[n,n] = sqrt(knn - ||row||^2)   (gp.hpp:596-597)
void launch_append_diag(hipStream_t s, double* Lrow, int64_t ldl, int64_t n, const double* knn, int* info);

// ---- gradient of the log-likelihood (grad.hpp:285-311) (grad.hip) ------------------
// partial[b, t] per lower-triangle tile b; then a fixed-order final reduction into grad[T]
void launch_grad_loglik(hipStream_t s, const double* Xt, int64_t ldx, int64_t N, const KParams& kp, const double* Kinv,
                        int64_t ldk, const double* alpha, int64_t lda, const double* uvec, int P, int n_theta,
                        int optimize_noise, double* partial, double* grad);
// leave-one-out (gp.hpp:339-402): v = alpha / kappa, sc = sqrt(c), val[i] per-sample terms, out[0] = LOO value
// (v, sc may be null: value only); S = sym(Kl) diag(sc); g[0..n) *= f
void launch_loo_prep(hipStream_t s, const double* Kinv, int64_t ldk, int64_t N, const double* alpha, int64_t lda, int P,
                     double* v, double* sc, double* val, double* out);
void launch_sym_colscale(hipStream_t s, const double* Kl, int64_t ldk, int64_t N, const double* sc, double* S,
                         int64_t lds_);
void launch_scale_vec(hipStream_t s, double* g, int n, double f);
int64_t grad_partial_size(int64_t N, int T);

// ---- one-launch small-N path (small.hip): add_sample and point queries below ~256 samples ------------
struct SmallAddArgs {
    double* A;          // factor (and, on exit, its new row n)
    int64_t ld;
    double* Xinv;
    double* Xt;         // SoA samples, column n is written
    int64_t ldx;
    double* Om;         // device obs_mean (ld), rewritten
    double* Al;         // device alpha (ld), rewritten
    const double* om_host; // pinned: (n + 1) x P, column-major, ld = n + 1
    double* out;        // pinned: [0] sum log L_ii, [1] sum obs_mean . alpha
    int* info;          // pinned: first non-positive pivot (1-based), written if zero
    unsigned long long* seq; // pinned: sequence word, written last
    unsigned long long seq_val;
    int n;              // samples before this call = index of the new one
};
struct SmallQueryArgs {
    const double* L;
    int64_t ld;
    const double* Xinv;
    const double* Xt;
    int64_t ldx;
    const double* Al;
    int P, n, M, D;
    const double* xq_host;  // pinned: M x D row-major query points
    double* kta_host;       // pinned: M x P, column-major (ld = M), may be null
    double* var_host;       // pinned: M, may be null
    unsigned long long* seq; // pinned: M sequence words (one per point)
    unsigned long long seq_val;
    int want_kta, want_var;
};

struct SmallAlphaArgs {
    const double* L;
    int64_t ld;
    const double* Xinv;
    const double* om_src; // obs_mean to solve for: pinned host memory (ldom = n) or the device copy (ldom = ld)
    int64_t ldom;
    double* Om;           // device obs_mean to refresh (null: om_src IS the device copy)
    double* Al;
    double* out;          // pinned: [0] sum log L_ii, [1] sum obs_mean . alpha
    unsigned long long* seq;
    unsigned long long seq_val;
    int n;
};
int small_max_n();
void launch_small_alpha(hipStream_t s, const SmallAlphaArgs& g, int P);
void launch_small_add(hipStream_t s, const SmallAddArgs& g, int P, const KParams& kp, const LamParams& lp, const double* x);
void launch_small_query(hipStream_t s, const SmallQueryArgs& g, const KParams& kp, const LamParams& lp);

// ---- micro-benchmarks (microbench.hip) ---------------------------------------------
double run_mfma_f64_peak(hipStream_t s);
double run_hbm_stream_peak(hipStream_t s);
