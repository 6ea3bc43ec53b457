// engine.hip — host side of libgpengine.so: the handle, HBM residency, the blocked algorithms'
// launch sequences and the C-ABI of include/gpe.h.
//
// What lives in HBM per handle (all fp64, column-major, leading dimension `ld`):
//   Xt    D x ld      samples, SoA (sample index contiguous)          gp.hpp:520 `_samples`
//   A     ld x cap    K, factored IN PLACE into L (lower)              gp.hpp:528/:530 `_kernel`/`_matrixL`
//   Om    ld x P      obs_mean = Y - m(X)                              gp.hpp:523 `_obs_mean`
//   Al    ld x P      alpha                                            gp.hpp:525 `_alpha`
//   Xinv  cap/64 x 64 x 64   L_bb^-T of every diagonal block (by-product of the factorisation)
// Rows N..N+P-1 of A carry obs_mean^T during the factorisation: the forward substitution
// z = L^-1 obs_mean (gp.hpp:608) rides along as P extra rows of every panel and update.
//   Linv  ld x cap    L^-1   (only once K^-1 is asked for)
//   Kinv  ld x cap    K^-1 lower triangle                              gp.hpp:528 `_inv_kernel`
// The reference keeps K, L and K^-1 as three N x N host matrices and deep-copies all of them for
// every hyper-parameter objective evaluation (kernel_lf_opt.hpp:79); here an evaluation is
// "same X, new theta" on resident buffers.
#include "../../include/gpe.h"
#include "dev.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <atomic>
#include <chrono>
#include <map>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <mutex>
#include <string>
#include <vector>

thread_local BatchLaunch g_batch; // dev.h: what this thread's launch wrappers add to every launch

// ---- launch tracing (dev.h: GPE_LAUNCH) --------------------------------------------------------------------------
namespace {
struct TraceRec {
    const char* name;
    hipStream_t stream;
    hipEvent_t e0, e1;
    unsigned gx, gy, gz, bx;
    bool own_stop; // e1 came from the trace pool (not the look-ahead's)
};
std::mutex g_trace_mu;
std::vector<TraceRec> g_trace;
std::vector<hipEvent_t> g_trace_pool;
std::atomic<int> g_trace_state{-1}; // -1: not looked at yet (GPE_TRACE), 0 off, 1 on
} // namespace
bool gpe_trace_on()
{
    int st = g_trace_state.load(std::memory_order_relaxed);
    if (st < 0) {
        const char* e = getenv("GPE_TRACE");
        st = e && atoi(e) != 0 ? 1 : 0;
        g_trace_state.store(st);
    }
    return st == 1;
}
hipEvent_t gpe_trace_event()
{
    std::lock_guard<std::mutex> lk(g_trace_mu);
    if (!g_trace_pool.empty()) {
        hipEvent_t e = g_trace_pool.back();
        g_trace_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
void gpe_trace_add(const char* name, hipStream_t s, hipEvent_t e0, hipEvent_t e1, dim3 grid, dim3 block)
{
    std::lock_guard<std::mutex> lk(g_trace_mu);
    g_trace.push_back(TraceRec{name, s, e0, e1, grid.x, grid.y, grid.z, block.x, false});
}

#define NB 64
// pinned staging of the small path: results in [0, 256), inputs (obs_mean: (n + 1) x P <= 257 x 3; query points: 8 x 64) from 256 on
#define SMALL_STAGE_DOUBLES (256 + 1024)

namespace {

struct PhaseRec {
    int phase;
    hipEvent_t e0, e1;
    double flops;
};

} // namespace

struct gpe_ctx {
    std::atomic<uint64_t> epoch{1}; // bumped by every call that can change what a query answers (gpe_epoch: round 6 — the C++ drop-in's
                                    // per-device query replicas are valid exactly as long as this has not moved)
    int device = 0;  // physical HIP device
    int ldevice = 0; // the device id the caller used (differs from `device` only under GPE_VIRTUAL_DEVICES)
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;      // look-ahead: bulk of a trailing update runs here, behind the next panel
    std::vector<hipEvent_t> la_events; // untimed events ordering the two streams
    int64_t tail_max = 2816;           // the last <= this many columns by one launch (k_tail; GPE_TAIL_MAX=0: by panels to the end);
                                       // 2816 against 2560: the one update in front of it has 253 tiles instead of 230 for the 256 CUs
                                       // (0.64 against 0.57 of the fp64 peak), profiles/r04_tail_max_sizes.log
    int64_t tail_single = 3328;        // round 6: an order up to this is ONE data-flow launch, whatever tail_max says (measured after the
                                       // chain work of rounds 5-6, profiles/r06_split_retune.log: N = 2880 0.657 -> 0.600 ms, 3072 0.689 ->
                                       // 0.646, 3200 0.792 -> 0.688, 3328 0.776 -> 0.732; 3584 and up: the three launches); 0 when
                                       // GPE_TAIL_MAX is set (the switch then means what it says)
    int64_t tall_max = 1536;           // ... and up to this many columns in FRONT of them as one tall data-flow launch (all rows
                                       // below ride along) followed by ONE update with k = its width (GPE_TALL=0: 256-column panels
                                       // with look-ahead all the way to the closing launch, the round-3 schedule)
    int64_t batch_tail_max = 1536;     // tail_max of a batched launch sequence (GPE_BATCH_TAIL_MAX): G members share the chip, so
                                       // more of the work belongs in the one update between the two data-flow launches (measured,
                                       // profiles/r04_batch_split_ab.log: 8 x N = 2048 1.29 ms per batch at 1536 against 1.48 at 2560)
    // the polled hand-over buffers of the data-flow launches, ONE allocation: [closing 0 | closing 1 | tall 0 | tall 1].  Of a
    // pair, the buffer of parity `count & 1` is armed (all-ones) for the slot layout of the pair's previous launch; a launch
    // with another layout (another N or P on this handle) first puts the whole pair back to all-ones (prepare_tail)
    double* dTail = nullptr;
    int64_t tail_cap = 0, tall_cap = 0; // doubles per buffer of the closing / tall pair
    unsigned tail_count = 0, tall_count = 0;
    int64_t tail_lay = -1, tall_lay = -1; // nt * 65536 + nb of the pair's previous launch (-1: both buffers entirely all-ones)
    int gen_mode = 0; // this evaluation's data-flow launches generate their own tiles of K (compute_enqueue): 0 no, 1 the one
                      // launch that is the whole factorisation, 2 the tall launch (the rest of K is built beside it)
    hipEvent_t gen_ev = nullptr;
    unsigned p256_count = 0;           // launches of k_panel256 so far: its polled X22 copies alternate between two buffers
    std::vector<hipEvent_t> pl_events; // ... one per outer panel: the one-launch panel is complete (early release of the look-ahead stream)
    int64_t early_bulk = 100;          // release the look-ahead stream at the END OF THE PANEL (not of the fused next-panel update)
                                       // when the far update has at least this many 128 x 128 tiles (GPE_EARLY_BULK_TILES; -1: never)
    bool stop_events = true;    // next-panel update signals through its own dispatch (hipExtLaunchKernel stop event)
    bool fuse_diag = true;      // next diagonal block factored inside the next-panel update launch (k_upd_fused)
    bool lookahead = true;             // GPE_LOOKAHEAD=0 disables
    int bulk_wgs = 192;                // physical workgroups of a look-ahead bulk update
    int near_wgs = 0;                  // workgroups of the "near" part of a look-ahead update (0: unrestricted — it is what the
                                       // next panel's update waits for; -1: bulk_wgs)
    int64_t bulk_free_tiles = 0;       // ... unless it has at least this many 128 x 128 tiles.  250 (the
                                       // first three far updates at N = 4096) while the panels ran step by step; with the
                                       // one-launch panels (64 CUs for ~55 us) every far update is better off unrestricted:
                                       // 640 -> 651/s at N = 4096 for 0..100, round 3
    std::mutex mu;
    int64_t N = 0, cap = 0, ld = 0;
    int D = 0, P = 0;
    int kind = GPE_KERNEL_SE_ARD, n_theta = 0;
    double theta[GPE_MAX_THETA] = {0};
    double noise = 0.01; // defaults::kernel::noise (kernel/kernel.hpp:57)
    KParams kp;
    double *dXt = nullptr, *dA = nullptr, *dOm = nullptr, *dAl = nullptr, *dW = nullptr, *dY = nullptr;
    double *dLinv = nullptr, *dKinv = nullptr, *dKhost = nullptr, *dGradPartial = nullptr, *dGrad = nullptr;
    double *dLooS = nullptr, *dLooV = nullptr; // leave-one-out scratch: N x N and N x (P + 2) (+8)
    double* dQuery = nullptr; // query scratch kept between calls while it is small (single-point queries: no malloc/free)
    size_t query_bytes = 0;
    double* dHead = nullptr; // scratch tiles of the fused panel steps (k_panel_step)
    bool panel_handover = true; // head tiles of a panel step change hands (potrf.hip); GPE_PANEL_HANDOVER=0: re-derived
    bool panel_handover_cfg = true; // what the caller / environment chose: a hand-over timeout switches panel_handover off
    int handover_off_left = 0;      // ... for this many evaluations only, then it is re-armed (one hiccup is not forever)
    int64_t handover_reruns = 0;    // evaluations re-run after a hand-over timeout (gpe_handover_reruns)
    double* dXinv = nullptr; // transposed inverses of the 64 x 64 diagonal blocks of L, 4096 doubles each
    bool panel256 = true;    // all steps of a 256-column outer panel in one data-flow launch (GPE_PANEL256=0: step by step)
    double* dXp = nullptr;   // inverses of the nbo x nbo diagonal panels of L, compact (ensure_inv with the overlapped product)
    double* dInvS = nullptr; // the recursive K^-1's scratch (inv2.hip): T-forms / W | three partial buffers, ld x cap each
    int invS_bufs = 0;       // ... how many ld x cap buffers it holds: 1 + inv2_partials() for a single handle, 1 for a member of a
                             // batch of >= 4 (whose plan cuts no k range: ADVICE r5 — 64 x N = 4096 used to reserve 26 GB it never touched)
    Inv2Plan* inv2 = nullptr; // ... and its plan, rebuilt when N, ld or a buffer changes
    int64_t inv_pad_n = -1;           // U and the T-form / W buffer read as zero beyond the inv_pad_n x inv_pad_n part (-1: unknown)
    Inv2Plan* inv2_batched = nullptr; // ... the plan of a batched sequence of >= 4 members led by this handle (no chunked k ranges)
    hipEvent_t chain_ev = nullptr; // the end of this handle's last evaluation chain when that ran on a CU-masked stream (ChainScope) ...
    bool chain_pending = false;    // ... and nobody has waited for it yet: the HOST does (wait_chain), never the handle's own stream
    bool inv_early = false;   // set by gpe_hp_objective around compute_enqueue: start K^-1's lowest level beside the sweep
    bool inv_prefix_done = false; // ... done on stream2 for the factor at hand; inv_ev completes behind it
    hipEvent_t inv_ev = nullptr, inv_ev0 = nullptr;
    size_t xp_cap = 0;
    int64_t grad_partial_cap = 0;
    int* dInfo = nullptr; // = hInfo: pinned host memory the kernels write directly (no copy-back, no device memset)
    double* dScal = nullptr; // [0] sum log L_ii, [1] trace(om^T alpha), [2] knn scratch, [8 .. 8 + 2 nblk) per-block partials
    int ll_partials = 0;     // > 0: the backward sweep left that many per-block partial sums instead of [0], [1]
    bool al_prefilled = false; // alpha holds the sentinel pattern of the data-flow sweep
    int* hInfo = nullptr;    // pinned
    double* hScal = nullptr; // pinned
    bool have_L = false, inv_ok = false, host_K = false, ll_ok = false;
    int nbo = 256; // outer panel width of the two-level blocked algorithms
    // one-launch small-N path (small.hip): pinned staging the kernels read / write directly, the word the host spins on
    char* hPinned = nullptr;             // the one pinned allocation behind hInfo / hSmallSeq / hScal / hSmall
    double* hSmall = nullptr;            // [0..2): log-lik terms | [16 .. 16+8*GPE_MAX_P+8): kta, var | [256..): obs_mean / query points in
    unsigned long long* hSmallSeq = nullptr; // 8 sequence words (one per query point; word 0 for add_sample)
    unsigned long long small_seq = 0;
    bool small_path = true;              // GPE_SMALL=0 disables
    int64_t small_calls = 0;             // calls served by the small path (instrumentation / tests)
    int64_t flow_retries = 0; // sweeps re-run block by block after a hand-off timeout (never expected; see flow_failed)
    bool flow_solve = true; // one data-flow launch for the backward sweep (GPE_FLOW_SOLVE=0: per-block launches)
    bool fuse_panel = true; // k_panel_step instead of the three-launch panel step (GPE_FUSE_PANEL=0 disables)
    // instrumentation
    bool prof = false;
    std::vector<PhaseRec> pending;
    std::vector<hipEvent_t> pool;
    double ph_ms[GPE_PH_COUNT] = {0}, ph_flops[GPE_PH_COUNT] = {0};
    int64_t ph_launches[GPE_PH_COUNT] = {0};
    std::string err;
};

namespace {

#define HIPCHK(c, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (c)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                            \
            return GPE_ERR_HIP;                                                                      \
        }                                                                                            \
    } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

int ld_pad() { return 16; } // break power-of-two column strides (HBM channel camping)

// leading dimension: cap rows + room for the P right-hand-side rows + a pad that breaks
// power-of-two column strides
int64_t ld_for(int64_t cap, int P) { return cap + round_up((int64_t)P + ld_pad(), 16); }

hipEvent_t get_event(gpe_ctx* c)
{
    if (!c->pool.empty()) {
        hipEvent_t e = c->pool.back();
        c->pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

// roctx ranges around every phase (tracing), so that a rocprofv3 --marker-trace timeline carries the phase
// names.  Off unless GPE_ROCTX=1; libroctx64 is looked up at run time (no link-time dependency of the product on it).
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char* e = getenv("GPE_ROCTX");
        if (!e || atoi(e) == 0)
            return;
        void* h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h)
            h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h)
            return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop)
            push = nullptr, pop = nullptr;
    }
};
const Roctx& roctx()
{
    static const Roctx r;
    return r;
}
const char* const kPhaseNames[GPE_PH_COUNT] = {"gpe:kernel_build", "gpe:potrf_panel", "gpe:potrf_update", "gpe:solve",
                                               "gpe:loglik",       "gpe:inv",         "gpe:grad",         "gpe:query",
                                               "gpe:potrf_tall",   "gpe:potrf_tail"};

struct PhaseScope {
    gpe_ctx* c;
    int phase;
    double flops;
    hipEvent_t e0 = nullptr;
    PhaseScope(gpe_ctx* c_, int ph, double fl = 0.0) : c(c_), phase(ph), flops(fl)
    {
        if (roctx().push)
            roctx().push(kPhaseNames[ph]);
        if (c->prof) {
            e0 = get_event(c);
            hipEventRecord(e0, c->stream);
        }
    }
    ~PhaseScope()
    {
        if (roctx().pop)
            roctx().pop();
        if (c->prof) {
            hipEvent_t e1 = get_event(c);
            hipEventRecord(e1, c->stream);
            c->pending.push_back(PhaseRec{phase, e0, e1, flops});
        }
    }
};

void drain_phases(gpe_ctx* c)
{
    for (auto& r : c->pending) {
        float ms = 0.f;
        hipEventSynchronize(r.e1);
        hipEventElapsedTime(&ms, r.e0, r.e1);
        c->ph_ms[r.phase] += ms;
        c->ph_flops[r.phase] += r.flops;
        c->ph_launches[r.phase] += 1;
        c->pool.push_back(r.e0);
        c->pool.push_back(r.e1);
    }
    c->pending.clear();
}

// rows of the SoA sample matrix: D inputs + room for the k <= D projection rows of SE-ARD's Lambda
int max_lam(int D) { return std::min(D, (GPE_MAX_THETA - 1 - D) / std::max(D, 1)); }
int xt_rows(int D) { return D + std::max(0, max_lam(D)); }
// SE-ARD: D + D k + 1 log-parameters (squared_exp_ard.hpp:94); the isotropic kernels: 2.  Returns k, or -1
int lam_columns(int kind, int n_theta, int D)
{
    if (kind != GPE_KERNEL_SE_ARD)
        return n_theta == 2 ? 0 : -1;
    if (D <= 0 || n_theta < D + 1 || (n_theta - 1) % D != 0)
        return -1;
    const int k = (n_theta - 1) / D - 1;
    return k <= max_lam(D) ? k : -1;
}

void free_dev(gpe_ctx* c)
{
    double** ps[] = {&c->dXt, &c->dA, &c->dOm, &c->dAl, &c->dW, &c->dY, &c->dLinv, &c->dKinv, &c->dKhost,
                     &c->dGradPartial, &c->dXinv, &c->dLooS, &c->dLooV};
    for (auto p : ps) {
        if (*p)
            hipFree(*p);
        *p = nullptr;
    }
    if (c->dQuery)
        hipFree(c->dQuery);
    c->dQuery = nullptr;
    c->query_bytes = 0;
    if (c->dXp)
        hipFree(c->dXp);
    c->dXp = nullptr;
    c->xp_cap = 0;
    if (c->dInvS)
        hipFree(c->dInvS);
    c->dInvS = nullptr;
    c->invS_bufs = 0;
    c->inv_pad_n = -1;
    inv2_plan_free(c->inv2);
    inv2_plan_free(c->inv2_batched);
    c->inv2 = c->inv2_batched = nullptr;
    c->grad_partial_cap = 0;
    c->cap = c->ld = 0;
}

// (re)allocate for capacity `cap` samples, dimension D, P outputs.  Existing contents are NOT kept.
int alloc_dev(gpe_ctx* c, int64_t cap, int D, int P)
{
    free_dev(c);
    cap = round_up(std::max<int64_t>(cap, NB), NB);
    int64_t ld = ld_for(cap, P);
    c->cap = cap;
    c->ld = ld;
    HIPCHK(c, hipMalloc(&c->dXt, sizeof(double) * (size_t)(ld * xt_rows(D))));
    HIPCHK(c, hipMalloc(&c->dA, sizeof(double) * (size_t)(ld * cap)));
    HIPCHK(c, hipMalloc(&c->dOm, sizeof(double) * (size_t)(ld * P)));
    HIPCHK(c, hipMalloc(&c->dAl, sizeof(double) * (size_t)(ld * P)));
    HIPCHK(c, hipMalloc(&c->dW, sizeof(double) * (size_t)(ld * std::max(P, 1))));
    HIPCHK(c, hipMalloc(&c->dY, sizeof(double) * (size_t)(ld * std::max(P, 1))));
    HIPCHK(c, hipMalloc(&c->dXinv, sizeof(double) * (size_t)(cap / NB) * NB * NB));
    HIPCHK(c, hipMemsetAsync(c->dXinv, 0, sizeof(double) * (size_t)(cap / NB) * NB * NB, c->stream));
    HIPCHK(c, hipMemsetAsync(c->dXt, 0, sizeof(double) * (size_t)(ld * xt_rows(D)), c->stream));
    return GPE_OK;
}

// grow capacity keeping X, L, Om (add_sample path; the reference reallocates K and L on every
// add_sample — gp.hpp:581,:588 conservativeResize — here capacity doubles)
int grow_dev(gpe_ctx* c, int64_t need)
{
    if (need <= c->cap)
        return GPE_OK;
    int64_t ncap = round_up(std::max<int64_t>(need, 2 * c->cap), NB);
    int D = c->D, P = c->P;
    int64_t nld = ld_for(ncap, P);
    double *nXt = nullptr, *nA = nullptr, *nOm = nullptr, *nAl = nullptr, *nW = nullptr, *nY = nullptr, *nXi = nullptr;
    HIPCHK(c, hipMalloc(&nXt, sizeof(double) * (size_t)(nld * xt_rows(D))));
    HIPCHK(c, hipMalloc(&nA, sizeof(double) * (size_t)(nld * ncap)));
    HIPCHK(c, hipMalloc(&nOm, sizeof(double) * (size_t)(nld * P)));
    HIPCHK(c, hipMalloc(&nAl, sizeof(double) * (size_t)(nld * P)));
    HIPCHK(c, hipMalloc(&nW, sizeof(double) * (size_t)(nld * P)));
    HIPCHK(c, hipMalloc(&nY, sizeof(double) * (size_t)(nld * P)));
    HIPCHK(c, hipMalloc(&nXi, sizeof(double) * (size_t)(ncap / NB) * NB * NB));
    HIPCHK(c, hipMemsetAsync(nXi, 0, sizeof(double) * (size_t)(ncap / NB) * NB * NB, c->stream));
    HIPCHK(c, hipMemsetAsync(nXt, 0, sizeof(double) * (size_t)(nld * xt_rows(D)), c->stream));
    if (c->N > 0) {
        launch_copy2d(c->stream, c->dXt, c->ld, nXt, nld, c->N, xt_rows(D));
        launch_copy2d(c->stream, c->dA, c->ld, nA, nld, c->N, c->N);
        hipMemcpyAsync(nXi, c->dXinv, sizeof(double) * (size_t)(c->cap / NB) * NB * NB, hipMemcpyDeviceToDevice,
                       c->stream);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double* old[] = {c->dXt, c->dA, c->dOm, c->dAl, c->dW, c->dY, c->dLinv, c->dKinv, c->dXinv, c->dLooS, c->dLooV, c->dInvS};
    for (double* p : old)
        if (p)
            hipFree(p);
    c->dInvS = nullptr;
    c->invS_bufs = 0;
    c->dXt = nXt;
    c->dA = nA;
    c->dOm = nOm;
    c->dAl = nAl;
    c->dW = nW;
    c->dY = nY;
    c->dXinv = nXi;
    c->dLinv = c->dKinv = c->dLooS = c->dLooV = nullptr;
    c->inv_pad_n = -1;
    c->inv_ok = false;
    c->cap = ncap;
    c->ld = nld;
    return GPE_OK;
}

void digest_kernel(gpe_ctx* c)
{
    KParams& k = c->kp;
    memset(&k, 0, sizeof(k));
    k.kind = c->kind;
    k.D = k.Din = c->D;
    k.noise = c->noise;
    k.diag_add = c->noise + 1e-8; // kernel.hpp:83
    if (c->kind == GPE_KERNEL_SE_ARD) {
        // SquaredExpARD::set_params, squared_exp_ard.hpp:96-105
        for (int d = 0; d < c->D && d < GPE_MAX_THETA; ++d)
            k.inv_ell[d] = 1.0 / std::exp(c->theta[d]);
        k.k_lam = std::max(0, lam_columns(c->kind, c->n_theta, c->D));
        k.D = c->D + k.k_lam; // the projections Lambda^T x are extra rows with unit length scale
        for (int j = 0; j < k.k_lam; ++j)
            k.inv_ell[c->D + j] = 1.0;
        k.sf2 = std::exp(2.0 * c->theta[c->n_theta - 1]);
        k.inv_l = 1.0;
    }
    else {
        // MaternFiveHalves::set_params (matern_five_halves.hpp:97-102), same for Matern3/2, Exp
        double l = std::exp(c->theta[0]);
        k.inv_l = 1.0 / l;
        k.sf2 = std::exp(2.0 * c->theta[1]);
        for (int d = 0; d < c->D && d < GPE_MAX_THETA; ++d)
            k.inv_ell[d] = k.inv_l;
    }
}

// rows D .. D+k-1 of a SoA point matrix <- Lambda^T x for columns [col0, col0 + n)  (no-op for k = 0)
void project_lambda(gpe_ctx* c, hipStream_t s, double* Xt, int64_t ld, int64_t col0, int64_t n)
{
    if (c->kp.k_lam <= 0)
        return;
    LamParams lp;
    lp.D = c->D;
    lp.k = c->kp.k_lam;
    for (int q = 0; q < lp.D * lp.k; ++q)
        lp.A[q] = c->theta[c->D + q]; // squared_exp_ard.hpp:100-102: _A(i, j) = p((j + 1) D + i), not in log-space
    launch_lambda_rows(s, Xt, ld, col0, n, lp);
}


#include "schedule.hpp" // the launch schedule of ONE evaluation

// The small kernels write their results and then a sequence word straight into pinned host memory: spin on the
// word(s) instead of synchronising the stream (an event round trip costs more than the kernel).  Falls back to a
// stream synchronisation after 50 ms (a fault, or a debugger).
static int small_wait(gpe_ctx* c, int nwords, unsigned long long want)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        bool all = true;
        for (int i = 0; i < nwords; ++i)
            all = all && (__atomic_load_n(c->hSmallSeq + i, __ATOMIC_ACQUIRE) == want);
        if (all)
            return GPE_OK;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipGetLastError());
            for (int i = 0; i < nwords; ++i)
                if (__atomic_load_n(c->hSmallSeq + i, __ATOMIC_ACQUIRE) != want) {
                    c->err = "small path: the kernel finished without publishing its results";
                    return GPE_ERR_HIP;
                }
            return GPE_OK;
        }
    }
}

static LamParams lam_params(const gpe_ctx* c)
{
    LamParams lp;
    lp.D = c->D;
    lp.k = c->kp.k_lam;
    for (int q = 0; q < lp.D * lp.k && q < GPE_MAX_THETA; ++q)
        lp.A[q] = c->theta[c->D + q];
    return lp;
}

#include "inverse.hpp" // K^-1 (the recursion on the factor and the panel form), the LOO weight matrix, the gradient objectives' enqueue

// The look-ahead stream runs the bulk of a trailing update while the main stream factors the next
// panel.  A GEMM workgroup (147 KB LDS) and a panel-step workgroup (115 KB) cannot share a CU, so a
// bulk update that owns all 256 CUs would simply delay the panel: the bulk update is launched with
// `bulk_wgs` < 256 looping workgroups (gemm.hip, GemmArgs::grid_limit), the other CUs stay free for
// the critical path.  (A CU mask on the stream was tried first and had no effect.)
// The main stream is created at the device's HIGHEST priority, the look-ahead stream at the default one: the runtime keeps a
// pool of hardware queues per priority, so the two can never share a hardware queue.  With both at the default priority the
// runtime mapped them onto the SAME queue for some creation histories (a third handle alive, sixteen streams created and
// destroyed before: profiles/r04_stream_queue_mapping.log) and every overlap of this file — look-ahead updates, K^-1 beside
// the gradient's pair sums — silently serialised: gpe_hp_objective 3.22 ms instead of 2.85.  (The priority itself has no
// measurable effect on how the chip schedules the two; GPE_STREAM_PRIO=0 restores two default-priority streams.)
hipError_t create_main_stream(hipStream_t* st)
{
    static const bool use = !(getenv("GPE_STREAM_PRIO") && atoi(getenv("GPE_STREAM_PRIO")) == 0);
    int lo = 0, hi = 0;
    if (use && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
        return hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi);
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
hipError_t create_bulk_stream(hipStream_t* st) { return hipStreamCreateWithFlags(st, hipStreamNonBlocking); }

struct DevGuard {
    explicit DevGuard(gpe_ctx* c) { hipSetDevice(c->device); }
};


// what survives a handle: see gpe_create
struct HandleShell {
    hipStream_t stream, stream2;
    double* dScal;
    char* hPinned;
};
std::mutex g_shell_mu;
std::vector<HandleShell> g_shells[16];

// Devices as the callers count them.  GPE_VIRTUAL_DEVICES=n (tests): n logical devices dealt round-robin over the
// physical ones, so that the multi-device placement of the C++ policies (clones of one GP on several devices,
// gpe_clone_to) is exercised on a one-GPU box.
int physical_devices()
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
int logical_devices()
{
    const int phys = physical_devices();
    if (const char* e = getenv("GPE_VIRTUAL_DEVICES")) {
        const int v = atoi(e);
        if (v > 0 && phys > 0)
            return v;
    }
    return phys;
}

} // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
#include "partitions.hpp" // who may launch what when

extern "C" {

const char* gpe_version(void) { return "limbo_amd-gpe 0.1 (gfx950)"; }

int gpe_create(int device_id, gpe_handle* out)
{
    if (!out)
        return GPE_ERR_ARG;
    const int phys = physical_devices();
    if (phys <= 0 || device_id < 0 || device_id >= logical_devices())
        return GPE_ERR_HIP;
    gpe_ctx* c = new gpe_ctx();
    c->ldevice = device_id;
    c->device = device_id % phys;
    if (hipSetDevice(c->device) != hipSuccess) {
        delete c;
        return GPE_ERR_HIP;
    }
    // Streams, the scratch block and the pinned block of a destroyed handle are kept for the next one on that device
    // (limbo creates and drops GPs freely: value semantics, one clone per hyper-parameter fit and thread —
    // kernel_lf_opt.hpp:79; creating two streams and a pinned allocation costs milliseconds).
    bool reused = false;
    if (c->device < 16) {
        std::lock_guard<std::mutex> lk(g_shell_mu);
        auto& pool = g_shells[c->device];
        if (!pool.empty()) {
            const HandleShell sh = pool.back();
            pool.pop_back();
            c->stream = sh.stream;
            c->stream2 = sh.stream2;
            c->dScal = sh.dScal;
            c->hPinned = sh.hPinned;
            reused = true;
        }
    }
    if (!reused
        && (create_main_stream(&c->stream) != hipSuccess || create_bulk_stream(&c->stream2) != hipSuccess
            // one device block [dScal 8 KiB | dHead GPE_HEAD_TILES tiles] and one coherent (fine-grained) pinned block
            // [hInfo 64 B | hSmallSeq 64 B | hScal 8 KiB | hSmall]: the small path's host side reads the pinned words while
            // the stream is still busy
            || hipMalloc(&c->dScal, 8192 + sizeof(double) * GPE_HEAD_TILES * NB * NB) != hipSuccess
            // the hand-over flag words start from zero, in the order of the stream the panel steps run on
            || hipMemsetAsync(c->dScal + 1024 + 65 * NB * NB, 0, sizeof(double) * NB * NB, c->stream) != hipSuccess
            || hipHostMalloc(&c->hPinned, 128 + 8192 + sizeof(double) * SMALL_STAGE_DOUBLES, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)) {
        delete c;
        return GPE_ERR_HIP;
    }
    c->dHead = c->dScal + 1024;
    {
        // the CU-masked stream pairs of the chain partitions (ChainScope) are created when a device has TWO live handles for the
        // first time: creating them takes tens of milliseconds, which must not fall into the first evaluation that meets another
        // one.  (the count is of LIVE handles — gpe_destroy takes its handle off it again.)  The first live handle
        // also shows the process in the GPU's users file (other processes then take turns with it, xproc_enter).
        if (c->device < 16) {
            const int now_live = g_live[c->device].fetch_add(1) + 1;
            GateDev& gd = gate_dev();
            std::lock_guard<std::recursive_mutex> lk(gd.mu);
            if (now_live >= 1)
                xproc_show(gd);
            if (now_live >= 2 && gate_on() && partitions_on())
                (void)partition_streams(gd);
        }
    }
    // the polled X22 copies of k_panel256 (both buffers: a block from the pool may have been left in either state)
    hipMemsetAsync(c->dHead + GPE_S22_TILE * (NB * NB), 0xFF, sizeof(double) * GPE_S22_TILES * NB * NB, c->stream);
    hipMemsetAsync(c->dHead + (32 + GPE_S22_TILE) * (NB * NB), 0xFF, sizeof(double) * GPE_S22_TILES * NB * NB, c->stream);
    c->hInfo = (int*)c->hPinned;
    c->hSmallSeq = (unsigned long long*)(c->hPinned + 64);
    c->hScal = (double*)(c->hPinned + 128);
    c->hSmall = c->hScal + 1024;
    memset(c->hPinned, 0, 128);
    if (const char* f = getenv("GPE_SMALL"))
        c->small_path = atoi(f) != 0;
    c->dInfo = c->hInfo; // mapped pinned memory: same address on the device (unified addressing)
    if (const char* f = getenv("GPE_PANEL_HANDOVER"))
        c->panel_handover = atoi(f) != 0;
    c->panel_handover_cfg = c->panel_handover;
    if (const char* f = getenv("GPE_FUSE_DIAG"))
        c->fuse_diag = atoi(f) != 0;
    if (const char* f = getenv("GPE_PANEL256"))
        c->panel256 = atoi(f) != 0;
    if (const char* f = getenv("GPE_EARLY_BULK_TILES"))
        c->early_bulk = atoll(f);
    if (const char* f = getenv("GPE_TAIL_MAX")) {
        c->tail_max = std::min<int64_t>(std::max<int64_t>(atoll(f), 0), GPE_TAIL_MAX);
        c->tail_single = 0;
    }
    if (const char* f = getenv("GPE_TALL"))
        c->tall_max = std::min<int64_t>(std::max<int64_t>(atoll(f), 0), GPE_TAIL_MAX);
    c->batch_tail_max = std::min(c->batch_tail_max, c->tail_max);
    if (const char* f = getenv("GPE_BATCH_TAIL_MAX"))
        c->batch_tail_max = std::min<int64_t>(std::max<int64_t>(atoll(f), 0), c->tail_max);
    if (const char* f = getenv("GPE_STOP_EVENT"))
        c->stop_events = atoi(f) != 0;
    if (const char* f = getenv("GPE_LOOKAHEAD"))
        c->lookahead = atoi(f) != 0;
    if (const char* f = getenv("GPE_FLOW_SOLVE"))
        c->flow_solve = atoi(f) != 0;
    if (const char* f = getenv("GPE_FUSE_PANEL"))
        c->fuse_panel = atoi(f) != 0;
    const char* e = getenv("GPE_NBO");
    if (e) {
        int v = atoi(e);
        if (v >= 64 && v % 64 == 0)
            c->nbo = v;
    }
    *out = c;
    return GPE_OK;
}

int gpe_destroy(gpe_handle c)
{
    if (!c)
        return GPE_ERR_ARG;
    DevGuard g(c);
    hipStreamSynchronize(c->stream);
    if (c->gen_ev)
        hipEventDestroy(c->gen_ev);
    if (c->inv_ev) {
        hipEventDestroy(c->inv_ev);
        hipEventDestroy(c->inv_ev0);
    }
    if (c->chain_ev) {
        (void)wait_chain(c);
        hipEventDestroy(c->chain_ev);
    }
    drain_phases(c);
    for (auto e : c->pool)
        hipEventDestroy(e);
    free_dev(c);
    if (c->dTail)
        hipFree(c->dTail);
    for (auto e : c->pl_events)
        hipEventDestroy(e);
    for (auto e : c->la_events)
        hipEventDestroy(e);
    hipStreamSynchronize(c->stream2);
    bool kept = false;
    if (c->device < 16) {
        std::lock_guard<std::mutex> lk(g_shell_mu);
        auto& pool = g_shells[c->device];
        if (pool.size() < 64) {
            pool.push_back(HandleShell{c->stream, c->stream2, c->dScal, c->hPinned});
            kept = true;
        }
    }
    if (!kept) {
        hipFree(c->dScal);
        hipHostFree(c->hPinned);
        flow_gate_forget(c->stream2);
        flow_gate_forget(c->stream);
        hipStreamDestroy(c->stream2);
        hipStreamDestroy(c->stream);
    }
    if (c->device < 16 && g_live[c->device].fetch_sub(1) == 1) { // the process's last handle on this GPU: nothing of it can be in flight
        GateDev& gd = gate_dev();
        std::lock_guard<std::recursive_mutex> lk(gd.mu);
        if (g_live[c->device].load() == 0)
            xproc_hide(gd);
    }
    delete c;
    return GPE_OK;
}

const char* gpe_last_error(gpe_handle c) { return c ? c->err.c_str() : "null handle"; }

int gpe_set_data(gpe_handle c, const double* X, int64_t N, int D, const double* obs_mean, int P)
{
    if (c)
        ++c->epoch;
    if (!c || !X || !obs_mean || N <= 0 || D <= 0 || D > GPE_MAX_THETA - 2 || P <= 0)
        return GPE_ERR_ARG;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (N > c->cap || D != c->D || P != c->P || !c->dA) {
        int rc = alloc_dev(c, N, D, P);
        if (rc)
            return rc;
    }
    if (c->dLinv) {
        hipFree(c->dLinv);
        c->dLinv = nullptr;
    }
    if (c->dKinv) {
        hipFree(c->dKinv);
        c->dKinv = nullptr;
    }
    for (double** q : {&c->dLooS, &c->dLooV})
        if (*q) {
            hipFree(*q);
            *q = nullptr;
        }
    c->N = N;
    c->D = D;
    c->P = P;
    c->have_L = c->inv_ok = c->ll_ok = false;
    c->host_K = (c->kind == GPE_KERNEL_HOST_K);
    // stage X through the (not yet used) matrix buffer, then transpose to SoA on the device
    double* tmp = c->dA;
    HIPCHK(c, hipMemcpyAsync(tmp, X, sizeof(double) * (size_t)(N * D), hipMemcpyHostToDevice, c->stream));
    launch_transpose_x(c->stream, tmp, N, D, c->dXt, c->ld, 0);
    HIPCHK(c, hipMemcpy2DAsync(c->dOm, sizeof(double) * c->ld, obs_mean, sizeof(double) * N, sizeof(double) * N, P,
                               hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GPE_OK;
}

int gpe_set_data_device(gpe_handle c, const double* dX, int64_t N, int D, const double* dOm, int P)
{
    if (c)
        ++c->epoch;
    if (!c || !dX || !dOm || N <= 0 || D <= 0 || D > GPE_MAX_THETA - 2 || P <= 0)
        return GPE_ERR_ARG;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (N > c->cap || D != c->D || P != c->P || !c->dA) {
        int rc = alloc_dev(c, N, D, P);
        if (rc)
            return rc;
    }
    c->N = N;
    c->D = D;
    c->P = P;
    c->have_L = c->inv_ok = c->ll_ok = false;
    c->host_K = (c->kind == GPE_KERNEL_HOST_K);
    launch_transpose_x(c->stream, dX, N, D, c->dXt, c->ld, 0);
    launch_copy2d(c->stream, dOm, N, c->dOm, c->ld, N, P);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GPE_OK;
}

int gpe_set_kernel(gpe_handle c, int kind, const double* th, int n_theta, double noise)
{
    if (c)
        ++c->epoch;
    if (!c || kind < 0 || kind > GPE_KERNEL_HOST_K || n_theta < 0 || n_theta > GPE_MAX_THETA)
        return GPE_ERR_ARG;
    if (n_theta > 0 && !th)
        return GPE_ERR_ARG;
    c->kind = kind;
    c->n_theta = n_theta;
    for (int i = 0; i < n_theta; ++i)
        c->theta[i] = th[i];
    c->noise = noise;
    c->host_K = (kind == GPE_KERNEL_HOST_K);
    return GPE_OK;
}

int gpe_set_K_host(gpe_handle c, const double* K, int64_t ldk)
{
    if (c)
        ++c->epoch;
    if (!c || !K || c->N <= 0 || ldk < c->N)
        return GPE_ERR_ARG;
    DevGuard g(c);
    {
        std::lock_guard<std::mutex> lk(c->mu);
    }
    if (!c->dKhost)
        HIPCHK(c, hipMalloc(&c->dKhost, sizeof(double) * (size_t)(c->ld * c->cap)));
    HIPCHK(c, hipMemcpy2D(c->dKhost, sizeof(double) * c->ld, K, sizeof(double) * ldk, sizeof(double) * c->N, c->N,
                          hipMemcpyHostToDevice));
    c->host_K = true;
    c->kind = GPE_KERNEL_HOST_K;
    return GPE_OK;
}

int gpe_compute(gpe_handle c)
{
    if (c)
        ++c->epoch;
    if (!c)
        return GPE_ERR_ARG;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->host_K) {
        if (lam_columns(c->kind, c->n_theta, c->D) < 0) {
            c->err = "set_kernel: wrong number of hyper-parameters for this kernel/dimension";
            return GPE_ERR_ARG;
        }
    }
    int rc = compute_enqueue(c);
    if (rc)
        return rc;
    return compute_finish(c);
}

int gpe_update_alpha(gpe_handle c, const double* obs_mean)
{
    if (c)
        ++c->epoch;
    if (!c)
        return GPE_ERR_ARG;
    if (!c->have_L)
        return GPE_ERR_STATE;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->small_path && c->N <= small_max_n() && c->P <= 4 && (int64_t)c->N * c->P <= 1024) {
        // one launch, no copies (small.hip): obs_mean read from pinned memory (or the device copy when none is given)
        SmallAlphaArgs a{};
        a.L = c->dA;
        a.ld = c->ld;
        a.Xinv = c->dXinv;
        if (obs_mean) {
            memcpy(c->hSmall + 256, obs_mean, sizeof(double) * (size_t)(c->N * c->P));
            a.om_src = c->hSmall + 256;
            a.ldom = c->N;
            a.Om = c->dOm;
        }
        else {
            a.om_src = c->dOm;
            a.ldom = c->ld;
            a.Om = nullptr;
        }
        a.Al = c->dAl;
        a.out = c->hSmall;
        a.seq = c->hSmallSeq;
        a.seq_val = ++c->small_seq;
        a.n = (int)c->N;
        {
            PhaseScope ps(c, GPE_PH_SOLVE, 2.0 * (double)c->N * c->N * c->P);
            launch_small_alpha(c->stream, a, c->P);
        }
        c->al_prefilled = false;
        c->ll_partials = 0;
        ++c->small_calls;
        int rc = small_wait(c, 1, a.seq_val);
        drain_phases(c);
        if (rc)
            return rc;
        c->hScal[0] = c->hSmall[0];
        c->hScal[1] = c->hSmall[1];
        c->ll_ok = true;
        return GPE_OK;
    }
    if (obs_mean)
        HIPCHK(c, hipMemcpy2DAsync(c->dOm, sizeof(double) * c->ld, obs_mean, sizeof(double) * c->N,
                                   sizeof(double) * c->N, c->P, hipMemcpyHostToDevice, c->stream));
    solve_alpha(c);
    enqueue_loglik_terms(c);
    int rc = compute_finish(c);
    return rc < 0 ? rc : GPE_OK;
}

__global__ void k_vec_to_row(const double* __restrict__ v, double* __restrict__ row, int64_t ld, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        row[i * ld] = v[i];
}
__global__ void k_knn(const double* __restrict__ kcol, int64_t n, double diag_add, double* __restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        out[0] = kcol[n] + diag_add;
}

int gpe_add_sample(gpe_handle c, const double* x, int D, const double* obs_mean, int P)
{
    if (c)
        ++c->epoch;
    if (!c || !x || !obs_mean || D <= 0 || P <= 0)
        return GPE_ERR_ARG;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->host_K)
        return GPE_ERR_UNSUPPORTED;
    hipStream_t s = c->stream;
    if (c->N == 0) { // gp.hpp:128-137
        if (D > GPE_MAX_THETA - 2)
            return GPE_ERR_ARG;
        int rc = alloc_dev(c, 256, D, P);
        if (rc)
            return rc;
        c->D = D;
        c->P = P;
    }
    else {
        if (D != c->D || P != c->P) // gp.hpp:139-140
            return GPE_ERR_ARG;
        if (!c->have_L)
            return GPE_ERR_STATE;
        int rc = grow_dev(c, c->N + 1);
        if (rc)
            return rc;
    }
    {
        if (lam_columns(c->kind, c->n_theta, c->D) < 0) {
            c->err = "set_kernel: wrong number of hyper-parameters for this kernel/dimension";
            return GPE_ERR_ARG;
        }
    }
    const int64_t n = c->N; // index of the new sample
    const int64_t ld = c->ld;
    digest_kernel(c);
    if (c->small_path && n >= 1 && n <= small_max_n() && P <= 3 && c->have_L) {
        // one launch, no copies (small.hip): x travels as a kernel argument, obs_mean is read from pinned memory
        double* om_stage = c->hSmall + 256;
        memcpy(om_stage, obs_mean, sizeof(double) * (size_t)((n + 1) * P));
        c->hInfo[0] = c->hInfo[1] = 0;
        SmallAddArgs a{};
        a.A = c->dA;
        a.ld = ld;
        a.Xinv = c->dXinv;
        a.Xt = c->dXt;
        a.ldx = ld;
        a.Om = c->dOm;
        a.Al = c->dAl;
        a.om_host = om_stage;
        a.out = c->hSmall;
        a.info = c->hInfo;
        a.seq = c->hSmallSeq;
        a.seq_val = ++c->small_seq;
        a.n = (int)n;
        {
            PhaseScope ps(c, GPE_PH_POTRF_PANEL, (double)n * n + 2.0 * (double)n * n * P);
            launch_small_add(s, a, P, c->kp, lam_params(c), x);
        }
        c->N = n + 1;
        c->have_L = true;
        c->inv_ok = false; // gp.hpp:602
        c->al_prefilled = false;
        c->ll_partials = 0;
        ++c->small_calls;
        int rc = small_wait(c, 1, a.seq_val);
        drain_phases(c);
        if (rc)
            return rc;
        c->hScal[0] = c->hSmall[0]; // sum log L_ii
        c->hScal[1] = c->hSmall[1]; // sum obs_mean . alpha
        c->ll_ok = true;
        return *c->hInfo;
    }
    // new sample -> column n of Xt (staged through dY)
    HIPCHK(c, hipMemcpyAsync(c->dY, x, sizeof(double) * D, hipMemcpyHostToDevice, s));
    launch_transpose_x(s, c->dY, 1, D, c->dXt, ld, n);
    project_lambda(c, s, c->dXt, ld, n, 1);
    HIPCHK(c, hipMemcpy2DAsync(c->dOm, sizeof(double) * ld, obs_mean, sizeof(double) * (n + 1),
                               sizeof(double) * (n + 1), P, hipMemcpyHostToDevice, s));
    c->hInfo[0] = c->hInfo[1] = 0; // nothing of this handle is in flight here
    // k(x_i, x_new) for i = 0..n (gp.hpp:583-586), no noise yet
    launch_build_Ks(s, c->dXt, ld, n + 1, c->dXt + n, ld, 1, c->kp, c->dW, ld);
    GPE_LAUNCH(k_knn, dim3(1), dim3(1), 0, s, c->dW, n, c->kp.diag_add, c->dScal + 2);
    auto new_row = [c, s, n, ld] {
        PhaseScope ps(c, GPE_PH_POTRF_PANEL, (double)n * n);
        if (n > 0) {
            // new row of L by forward substitution (gp.hpp:591-594): L[n, 0:n] = (L^-1 k[0:n])^T
            if (c->flow_solve && (n + NB - 1) / NB <= 256)
                launch_trsv_fwd_flow(s, c->dA, ld, n, c->dXinv, c->dW, ld, c->dY, ld, 1, c->dInfo + 1);
            else
                launch_trsv_sweep(s, c->dA, ld, n, c->dXinv, c->dW, c->dY, ld, 1, 0);
            GPE_LAUNCH(k_vec_to_row, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->dY, c->dA + n, ld,
                               n);
        }
        launch_append_diag(s, c->dA + n, ld, n, c->dScal + 2, c->dInfo); // gp.hpp:596-597
        launch_diag_inv(s, c->dA, ld, n + 1, n / NB, 1, c->dXinv); // the last block gained a row
    };
    new_row();
    c->N = n + 1;
    c->have_L = true;
    c->inv_ok = false; // gp.hpp:602
    solve_alpha(c);    // gp.hpp:599
    enqueue_loglik_terms(c);
    return compute_finish(c, [c, new_row] {
        c->hInfo[0] = 0; // the pivot word of the failed attempt came from a row that was never completed
        new_row();       // (the one-launch sweep leaves its input c->dW untouched)
        solve_alpha(c);
        enqueue_loglik_terms(c);
    });
}

int gpe_log_lik(gpe_handle c, double* out)
{
    if (!c || !out)
        return GPE_ERR_ARG;
    if (!c->have_L)
        return GPE_ERR_STATE;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->ll_ok) {
        enqueue_loglik_terms(c);
        int rc = compute_finish(c);
        if (rc < 0)
            return rc;
    }
    // gp.hpp:274-279 (P-quirk: logdet and n log 2 pi are not multiplied by P)
    long double logdet = 2 * c->hScal[0];
    double a = c->hScal[1];
    *out = (double)(-0.5 * a - 0.5 * logdet - 0.5 * c->N * std::log(2 * M_PI));
    return GPE_OK;
}

int gpe_compute_inv_kernel(gpe_handle c)
{
    if (!c)
        return GPE_ERR_ARG;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ensure_inv(c);
    if (rc)
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    drain_phases(c);
    return GPE_OK;
}

// shared by the two gradient entry points: enqueue, fetch, and — should a one-launch sweep have given up — once more
// with one launch per block
static int grad_fetch(gpe_ctx* c, double* grad, int n_grad, int optimize_noise, bool loo)
{
    auto once = [&]() -> int {
        int rc = grad_enqueue(c, n_grad, optimize_noise, loo);
        if (rc)
            return rc;
        HIPCHK(c, hipMemcpyAsync(grad, c->dGrad, sizeof(double) * n_grad, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        drain_phases(c);
        return GPE_OK;
    };
    int rc = once();
    if (rc || !flow_failed(c))
        return rc;
    NoFlowScope off(c);
    rc = once();
    c->hInfo[1] = 0;
    return rc;
}

int gpe_log_lik_grad(gpe_handle c, double* grad, int n_grad, int optimize_noise)
{
    if (!c || !grad)
        return GPE_ERR_ARG;
    if (!c->have_L)
        return GPE_ERR_STATE;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    digest_kernel(c);
    return grad_fetch(c, grad, n_grad, optimize_noise, false);
}

// GP::compute_log_loo_cv (gp.hpp:339-351)
int gpe_log_loo_cv(gpe_handle c, double* out)
{
    if (!c || !out)
        return GPE_ERR_ARG;
    if (!c->have_L)
        return GPE_ERR_STATE;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ensure_inv(c);
    if (rc)
        return rc;
    rc = ensure_loo_bufs(c, false);
    if (rc)
        return rc;
    double* outp = c->dLooV + c->ld * (c->P + 2);
    {
        PhaseScope ps(c, GPE_PH_LOGLIK, 0.0);
        launch_loo_prep(c->stream, c->dKinv, c->ld, c->N, c->dAl, c->ld, c->P, nullptr, nullptr,
                        c->dLooV + c->ld * (c->P + 1), outp);
    }
    HIPCHK(c, hipMemcpyAsync(out, outp, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    drain_phases(c);
    return GPE_OK;
}

// GP::compute_kernel_grad_log_loo_cv (gp.hpp:354-402)
int gpe_log_loo_cv_grad(gpe_handle c, double* grad, int n_grad, int optimize_noise)
{
    if (!c || !grad)
        return GPE_ERR_ARG;
    if (!c->have_L)
        return GPE_ERR_STATE;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    digest_kernel(c);
    return grad_fetch(c, grad, n_grad, optimize_noise, true);
}

int gpe_hp_objective(gpe_handle c, int kind, const double* th, int n_theta, double noise, int optimize_noise,
                     int want_grad, double* lik, double* grad)
{
    if (!c || !lik)
        return GPE_ERR_ARG;
    int rc = gpe_set_kernel(c, kind, th, n_theta, noise); // kernel_lf_opt.hpp:80
    if (rc)
        return rc;
    static const bool fused_ok = !(getenv("GPE_HP_FUSED") && atoi(getenv("GPE_HP_FUSED")) == 0);
    if (want_grad && grad && fused_ok) {
        // Round 5: ONE enqueue for the whole objective — factorisation, alpha, log-lik terms, K^-1, gradient — and one wait.
        // The separate calls below cost a host round trip between the sweep and K^-1 (~20 us of idle chip), and K^-1's
        // lowest level (latency-bound launches on an eighth of the chip) can now run on the second stream BESIDE the sweep
        // (inv2_start_early).  Results are those of the separate calls bit for bit (same kernels, same order per buffer).
        DevGuard g(c);
        std::lock_guard<std::mutex> lk(c->mu);
        if (!c->host_K && lam_columns(c->kind, c->n_theta, c->D) < 0) {
            c->err = "set_kernel: wrong number of hyper-parameters for this kernel/dimension";
            return GPE_ERR_ARG;
        }
        const int n_grad = n_theta + (optimize_noise ? 1 : 0);
        const int64_t retries0 = c->flow_retries;
        c->inv_early = true;
        rc = compute_enqueue(c);
        c->inv_early = false;
        if (rc == GPE_OK && c->chain_pending) // (the chain ran on a masked stream: K^-1 and the gradient follow it by a host wait)
            rc = wait_chain(c) == hipSuccess ? GPE_OK : GPE_ERR_HIP;
        if (rc == GPE_OK)
            rc = grad_enqueue(c, n_grad, optimize_noise, false);
        if (c->inv_prefix_done) { // (not consumed: an error on the way) nothing may outlive this call on the second stream
            hipStreamWaitEvent(c->stream, c->inv_ev, 0);
            c->inv_prefix_done = false;
        }
        if (rc)
            return rc;
        HIPCHK(c, hipMemcpyAsync(grad, c->dGrad, sizeof(double) * n_grad, hipMemcpyDeviceToHost, c->stream));
        const int info = compute_finish(c); // one wait; a hand-over / sweep timeout (never expected) re-runs the evaluation
        if (info < 0)
            return info;
        if (c->flow_retries != retries0) { // ... and then the gradient belongs to the first attempt: again, on its own
            rc = grad_fetch(c, grad, n_grad, optimize_noise, false);
            if (rc)
                return rc;
        }
        const long double logdet = 2 * c->hScal[0]; // gp.hpp:274-279, as gpe_log_lik
        *lik = (double)(-0.5 * c->hScal[1] - 0.5 * logdet - 0.5 * c->N * std::log(2 * M_PI));
        return info;
    }
    int info = gpe_compute(c); // :82 recompute(false)
    if (info < 0)
        return info;
    rc = gpe_log_lik(c, lik); // :84
    if (rc)
        return rc;
    if (want_grad) { // :89
        if (!grad)
            return GPE_ERR_ARG;
        rc = gpe_log_lik_grad(c, grad, n_theta + (optimize_noise ? 1 : 0), optimize_noise);
        if (rc)
            return rc;
    }
    return info;
}

#include "query.hpp" // the batched query (gp.hpp

int gpe_query_batch(gpe_handle c, const double* Xq, int64_t M, double* kta, double* var)
{
    if (!c || !Xq || M < 0)
        return GPE_ERR_ARG;
    if (!c->have_L)
        return GPE_ERR_STATE;
    if (c->host_K)
        return GPE_ERR_UNSUPPORTED;
    if (M == 0)
        return GPE_OK;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu); // const queries from several host threads serialise here
    return query_impl(c, Xq, nullptr, M, kta, var);
}

int gpe_query_batch_cross(gpe_handle c, const double* Ks, int64_t M, double* kta, double* zz)
{
    if (!c || !Ks || M < 0)
        return GPE_ERR_ARG;
    if (!c->have_L)
        return GPE_ERR_STATE;
    if (M == 0)
        return GPE_OK;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = query_impl(c, nullptr, Ks, M, kta, zz);
    if (rc == GPE_OK && zz)
        for (int64_t m = 0; m < M; ++m)
            zz[m] = -zz[m]; // query_impl returned 0 - |L^-1 k*|^2
    return rc;
}

int gpe_set_obs_mean(gpe_handle c, const double* obs_mean)
{
    if (c)
        ++c->epoch;
    if (!c || !obs_mean || c->N <= 0 || !c->dOm)
        return GPE_ERR_ARG;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipMemcpy2DAsync(c->dOm, sizeof(double) * c->ld, obs_mean, sizeof(double) * c->N, sizeof(double) * c->N,
                               c->P, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->ll_ok = false;
    return GPE_OK;
}

int gpe_nb_samples(gpe_handle c, int64_t* N)
{
    if (!c || !N)
        return GPE_ERR_ARG;
    *N = c->N;
    return GPE_OK;
}

int gpe_get_L(gpe_handle c, double* L, int64_t ldh)
{
    if (!c || !L || ldh < c->N)
        return GPE_ERR_ARG;
    if (!c->have_L)
        return GPE_ERR_STATE;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    const int64_t N = c->N;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy2D(L, sizeof(double) * ldh, c->dA, sizeof(double) * c->ld, sizeof(double) * N, N,
                          hipMemcpyDeviceToHost));
    for (int64_t j = 1; j < N; ++j) // matrixL(): zero upper triangle (gp.hpp:411)
        memset(L + j * ldh, 0, sizeof(double) * (size_t)j);
    return GPE_OK;
}

int gpe_set_L(gpe_handle c, const double* L, int64_t ldh)
{
    if (c)
        ++c->epoch;
    if (!c || !L || c->N <= 0 || ldh < c->N)
        return GPE_ERR_ARG;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipMemcpy2D(c->dA, sizeof(double) * c->ld, L, sizeof(double) * ldh, sizeof(double) * c->N, c->N,
                          hipMemcpyHostToDevice));
    launch_diag_inv(c->stream, c->dA, c->ld, c->N, 0, (c->N + NB - 1) / NB, c->dXinv);
    if (!c->host_K && lam_columns(c->kind, c->n_theta, c->D) > 0) {
        // compute() never ran on this handle: the Lambda^T x rows of the training samples that the cross-kernel
        // and gradient kernels read (squared_exp_ard.hpp:142-146) are still to be formed
        digest_kernel(c);
        project_lambda(c, c->stream, c->dXt, c->ld, 0, c->N);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_L = true;
    c->inv_ok = false;
    c->ll_ok = false;
    return GPE_OK;
}

int gpe_get_alpha(gpe_handle c, double* a)
{
    if (!c || !a)
        return GPE_ERR_ARG;
    if (!c->have_L)
        return GPE_ERR_STATE;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy2D(a, sizeof(double) * c->N, c->dAl, sizeof(double) * c->ld, sizeof(double) * c->N, c->P,
                          hipMemcpyDeviceToHost));
    return GPE_OK;
}

int gpe_set_alpha(gpe_handle c, const double* a)
{
    if (c)
        ++c->epoch;
    if (!c || !a || c->N <= 0)
        return GPE_ERR_ARG;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipMemcpy2D(c->dAl, sizeof(double) * c->ld, a, sizeof(double) * c->N, sizeof(double) * c->N, c->P,
                          hipMemcpyHostToDevice));
    c->ll_ok = false;
    return GPE_OK;
}

int gpe_get_Kinv(gpe_handle c, double* Kinv, int64_t ldh)
{
    if (!c || !Kinv || ldh < c->N)
        return GPE_ERR_ARG;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ensure_inv(c);
    if (rc)
        return rc;
    const int64_t N = c->N;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    drain_phases(c);
    HIPCHK(c, hipMemcpy2D(Kinv, sizeof(double) * ldh, c->dKinv, sizeof(double) * c->ld, sizeof(double) * N, N,
                          hipMemcpyDeviceToHost));
    for (int64_t j = 1; j < N; ++j) // mirror the lower triangle
        for (int64_t i = 0; i < j; ++i)
            Kinv[i + j * ldh] = Kinv[j + i * ldh];
    return GPE_OK;
}

// Weight matrix of the leave-one-out gradient (grad.hip header) on the host, for kernels whose
// d k / d theta only exists as a host functor: dLOO/dtheta_j = sum_ab W[a, b] dK_j[a, b].
int gpe_get_loo_weights(gpe_handle c, double* W, int64_t ldh)
{
    if (!c || !W || ldh < c->N)
        return GPE_ERR_ARG;
    if (!c->have_L)
        return GPE_ERR_STATE;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ensure_inv(c);
    if (rc)
        return rc;
    rc = loo_weights(c);
    if (rc)
        return rc;
    const int64_t N = c->N;
    const int P = c->P;
    std::vector<double> u((size_t)N * P), a((size_t)N * P);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    drain_phases(c);
    if (flow_failed(c)) { // once more, one launch per block
        NoFlowScope off(c);
        rc = loo_weights(c);
        if (rc)
            return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        drain_phases(c);
        c->hInfo[1] = 0;
    }
    HIPCHK(c, hipMemcpy2D(W, sizeof(double) * ldh, c->dLinv, sizeof(double) * c->ld, sizeof(double) * N, N,
                          hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy2D(u.data(), sizeof(double) * N, c->dLooV, sizeof(double) * c->ld, sizeof(double) * N, P,
                          hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy2D(a.data(), sizeof(double) * N, c->dAl, sizeof(double) * c->ld, sizeof(double) * N, P,
                          hipMemcpyDeviceToHost));
    for (int64_t j = 0; j < N; ++j)
        for (int64_t i = j; i < N; ++i) {
            double w = -W[i + j * ldh];
            for (int p = 0; p < P; ++p)
                w += 0.5 * (u[i + (size_t)p * N] * a[j + (size_t)p * N] + a[i + (size_t)p * N] * u[j + (size_t)p * N]);
            W[i + j * ldh] = W[j + i * ldh] = w;
        }
    return GPE_OK;
}

int gpe_get_K(gpe_handle c, double* K, int64_t ldh)
{
    if (!c || !K || ldh < c->N || c->N <= 0)
        return GPE_ERR_ARG;
    DevGuard g(c);
    std::lock_guard<std::mutex> lk(c->mu);
    const int64_t N = c->N;
    if (c->host_K) {
        if (!c->dKhost)
            return GPE_ERR_STATE;
        HIPCHK(c, hipMemcpy2D(K, sizeof(double) * ldh, c->dKhost, sizeof(double) * c->ld, sizeof(double) * N, N,
                              hipMemcpyDeviceToHost));
        return GPE_OK;
    }
    digest_kernel(c);
    double* tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, sizeof(double) * (size_t)(c->ld * N)));
    project_lambda(c, c->stream, c->dXt, c->ld, 0, N);
    launch_build_K_full(c->stream, c->dXt, c->ld, N, c->kp, tmp, c->ld);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess)
        e = hipMemcpy2D(K, sizeof(double) * ldh, tmp, sizeof(double) * c->ld, sizeof(double) * N, N,
                        hipMemcpyDeviceToHost);
    hipFree(tmp);
    HIPCHK(c, e);
    return GPE_OK;
}

int gpe_device_count(int* n)
{
    if (!n)
        return GPE_ERR_ARG;
    *n = logical_devices();
    return *n > 0 ? GPE_OK : GPE_ERR_HIP;
}

int gpe_xproc_waits(int64_t* n)
{
    if (!n)
        return GPE_ERR_ARG;
    *n = g_xproc_waits.load();
    return GPE_OK;
}
int gpe_epoch(gpe_handle c, uint64_t* epoch)
{
    if (!c || !epoch)
        return GPE_ERR_ARG;
    *epoch = c->epoch.load();
    return GPE_OK;
}
int gpe_get_device(gpe_handle c, int* device_id)
{
    if (!c || !device_id)
        return GPE_ERR_ARG;
    *device_id = c->ldevice;
    return GPE_OK;
}

int gpe_flow_retries(gpe_handle c, int64_t* n)
{
    if (!c || !n)
        return GPE_ERR_ARG;
    *n = c->flow_retries;
    return GPE_OK;
}

int gpe_handover_reruns(gpe_handle c, int64_t* n)
{
    if (!c || !n)
        return GPE_ERR_ARG;
    *n = c->handover_reruns;
    return GPE_OK;
}

int gpe_small_calls(gpe_handle c, int64_t* n)
{
    if (!c || !n)
        return GPE_ERR_ARG;
    *n = c->small_calls;
    return GPE_OK;
}

int gpe_clone(gpe_handle src, gpe_handle* out) { return src ? gpe_clone_to(src, src->ldevice, out) : GPE_ERR_ARG; }

int gpe_clone_to(gpe_handle src, int device_id, gpe_handle* out)
{
    if (!src || !out)
        return GPE_ERR_ARG;
    gpe_handle c = nullptr;
    int rc = gpe_create(device_id, &c);
    if (rc)
        return rc;
    std::lock_guard<std::mutex> lk(src->mu);
    {
        DevGuard gs(src);
        hipStreamSynchronize(src->stream);
    }
    DevGuard g(c);
    // device-to-device on one GPU, peer copy over xGMI between two (no host staging either way)
    const int sdev = src->device, ddev = c->device;
    auto copy = [&](void* dst, const void* from, size_t bytes) {
        return sdev == ddev ? hipMemcpyAsync(dst, from, bytes, hipMemcpyDeviceToDevice, c->stream)
                            : hipMemcpyPeerAsync(dst, ddev, from, sdev, bytes, c->stream);
    };
    c->kind = src->kind;
    c->n_theta = src->n_theta;
    memcpy(c->theta, src->theta, sizeof(c->theta));
    c->noise = src->noise;
    c->nbo = src->nbo;
    c->fuse_panel = src->fuse_panel;
    c->flow_solve = src->flow_solve;
    c->small_path = src->small_path;
    c->lookahead = src->lookahead;
    c->bulk_wgs = src->bulk_wgs;
    c->bulk_free_tiles = src->bulk_free_tiles;
    c->host_K = src->host_K;
    if (src->dA) {
        rc = alloc_dev(c, src->cap, src->D, src->P);
        if (rc) {
            gpe_destroy(c);
            return rc;
        }
        c->N = src->N;
        c->D = src->D;
        c->P = src->P;
        const size_t mat = sizeof(double) * (size_t)(c->ld * c->cap);
        copy(c->dXt, src->dXt, sizeof(double) * (size_t)(c->ld * xt_rows(c->D)));
        copy(c->dA, src->dA, mat);
        copy(c->dOm, src->dOm, sizeof(double) * (size_t)(c->ld * c->P));
        copy(c->dAl, src->dAl, sizeof(double) * (size_t)(c->ld * c->P));
        copy(c->dXinv, src->dXinv, sizeof(double) * (size_t)(c->cap / NB) * NB * NB);
        if (src->dKhost) {
            hipMalloc(&c->dKhost, mat);
            copy(c->dKhost, src->dKhost, mat);
        }
        if (src->inv_ok && src->dKinv) {
            hipMalloc(&c->dKinv, mat);
            copy(c->dKinv, src->dKinv, mat);
            c->inv_ok = true;
        }
        c->have_L = src->have_L;
        c->ll_ok = src->ll_ok;
        c->hScal[0] = src->hScal[0];
        c->hScal[1] = src->hScal[1];
        *c->hInfo = *src->hInfo;
        if (hipStreamSynchronize(c->stream) != hipSuccess) {
            gpe_destroy(c);
            return GPE_ERR_HIP;
        }
    }
    *out = c;
    return GPE_OK;
}

#include "batch.hpp" // G independent GPs stepped by ONE launch sequence (gridDim.z = GP)

int gpe_batch_compute(gpe_handle* hs, int G, int* status) { return batch_compute_impl(hs, G, status, nullptr); }

// KernelLFOptimization::operator() (kernel_lf_opt.hpp:77-92) for G clones at once — the restarts of
// opt::ParallelRepeater (parallel_repeater.hpp:84-105), the outputs of multi_gp::ParallelLFOpt (parallel_lf_opt.hpp:64-67):
// member g gets log_theta[g n_theta ..] and noise[g]; K -> L -> alpha -> log-lik -> K^-1 -> gradient of ALL members is one
// launch sequence (gridDim.z = member) when the handles agree in shape, per-member chains otherwise.
int gpe_batch_hp_objective(gpe_handle* hs, int G, int kind, const double* log_theta, int n_theta, const double* noise,
                           int optimize_noise, int want_grad, double* lik, double* grad, int* status)
{
    if (!hs || G < 0 || !log_theta || !noise || !lik || (want_grad && !grad))
        return GPE_ERR_ARG;
    for (int g = 0; g < G; ++g) {
        if (!hs[g])
            return GPE_ERR_ARG;
        int rc = gpe_set_kernel(hs[g], kind, log_theta + (size_t)g * n_theta, n_theta, noise[g]); // kernel_lf_opt.hpp:80
        if (rc)
            return rc;
        if (want_grad && hs[g]->host_K)
            return GPE_ERR_UNSUPPORTED;
        if (lam_columns(kind, n_theta, hs[g]->D) < 0) {
            hs[g]->err = "set_kernel: wrong number of hyper-parameters for this kernel/dimension";
            return GPE_ERR_ARG;
        }
    }
    BatchWant want;
    want.grad = want_grad != 0;
    want.optimize_noise = optimize_noise;
    want.n_grad = n_theta + (optimize_noise ? 1 : 0);
    want.grad_out = grad;
    std::vector<int> st(G, 0);
    int worst = batch_compute_impl(hs, G, st.data(), &want); // :82 recompute(false) (+ :89 the gradient)
    for (int g = 0; g < G; ++g) {
        if (status)
            status[g] = st[g];
        if (st[g] >= 0) {
            int rc = gpe_log_lik(hs[g], lik + g); // :84
            if (rc < 0)
                worst = rc;
        }
    }
    return worst;
}

int gpe_batch_log_lik(gpe_handle* hs, int G, double* out)
{
    if (!hs || !out)
        return GPE_ERR_ARG;
    for (int g = 0; g < G; ++g) {
        int rc = gpe_log_lik(hs[g], out + g);
        if (rc)
            return rc;
    }
    return GPE_OK;
}

int gpe_get_stream(gpe_handle c, void** stream)
{
    if (!c || !stream)
        return GPE_ERR_ARG;
    *stream = (void*)c->stream;
    return GPE_OK;
}

int gpe_synchronize(gpe_handle c)
{
    if (!c)
        return GPE_ERR_ARG;
    DevGuard g(c);
    {
        std::lock_guard<std::mutex> lk(c->mu);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GPE_OK;
}

int gpe_set_profiling(gpe_handle c, int on)
{
    if (!c)
        return GPE_ERR_ARG;
    c->prof = on != 0;
    return GPE_OK;
}

int gpe_get_phase_ms(gpe_handle c, double* ms, int64_t* launches, double* flops, int n)
{
    if (!c || n > GPE_PH_COUNT)
        return GPE_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        if (ms)
            ms[i] = c->ph_ms[i];
        if (launches)
            launches[i] = c->ph_launches[i];
        if (flops)
            flops[i] = c->ph_flops[i];
    }
    return GPE_OK;
}

int gpe_reset_phase_ms(gpe_handle c)
{
    if (!c)
        return GPE_ERR_ARG;
    for (int i = 0; i < GPE_PH_COUNT; ++i) {
        c->ph_ms[i] = c->ph_flops[i] = 0.0;
        c->ph_launches[i] = 0;
    }
    return GPE_OK;
}

// launch tracing: on / off (clears what was recorded), and the records so far as text:
//   <start us> <end us> <stream index> <kernel> grid=<x,y,z> block=<x>      (times from the first recorded launch's start)
int gpe_trace(int on)
{
    (void)gpe_trace_on();
    hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_trace_mu);
    for (auto& r : g_trace) {
        g_trace_pool.push_back(r.e0);
        g_trace_pool.push_back(r.e1);
    }
    g_trace.clear();
    g_trace_state.store(on ? 1 : 0);
    return GPE_OK;
}
int gpe_trace_dump(const char* path)
{
    if (!path)
        return GPE_ERR_ARG;
    hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_trace_mu);
    FILE* f = fopen(path, "w");
    if (!f)
        return GPE_ERR_ARG;
    std::map<hipStream_t, int> sid;
    hipEvent_t ref = g_trace.empty() ? nullptr : g_trace[0].e0;
    for (auto& r : g_trace) {
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, ref, r.e0) != hipSuccess || hipEventElapsedTime(&b, ref, r.e1) != hipSuccess)
            continue;
        if (!sid.count(r.stream))
            sid[r.stream] = (int)sid.size();
        fprintf(f, "%10.2f %10.2f %d %s grid=%u,%u,%u block=%u\n", 1e3 * a, 1e3 * b, sid[r.stream], r.name, r.gx, r.gy, r.gz, r.bx);
    }
    fclose(f);
    return GPE_OK;
}

int gpe_debug_tail_order(int nt, int nb, int lag, int pair) { return debug_tail_order(nt, nb, lag, pair); }
int gpe_debug_ragged_split(int64_t k, int64_t scratch_doubles, int* kc)
{
    if (!kc)
        return -1;
    *kc = 0;
    return ragged_split(k, scratch_doubles, kc);
}
int gpe_debug_tri_tile_map(int64_t m, int64_t n, int64_t grow0, int64_t gcol0, int* out, int cap)
{
    if (m <= 0 || n <= 0 || !out || cap <= 0 || (m + 127) / 128 >= 65536 || (n + 127) / 128 >= 32768)
        return -1;
    return debug_tri_tile_map(m, n, grow0, gcol0, out, cap);
}
int gpe_debug_chain_split(int wave, int* units10, int* cols)
{
    if (wave < 0 || wave > 7 || !units10 || !cols)
        return -1;
    debug_chain_split(wave, units10, cols);
    return 0;
}
int gpe_debug_inv_plan(int64_t n, int64_t ld, int nbins, int load_pct, int64_t* out, int64_t cap_rows)
{
    return inv2_debug_plan(n, ld, nbins, load_pct, out, cap_rows);
}

int gpe_debug_tail_plan(int64_t n, int p, int g, int64_t tail_max, int64_t tall_max, int64_t batch_tail_max, int64_t* out8)
{
    if (!out8 || n < 1 || p < 1)
        return GPE_ERR_ARG;
    debug_tail_plan(n, p, g, tail_max, tall_max, batch_tail_max, out8);
    return GPE_OK;
}

int gpe_mfma_f64_peak(int device_id, double* tflops)
{
    if (!tflops || hipSetDevice(device_id) != hipSuccess)
        return GPE_ERR_HIP;
    *tflops = run_mfma_f64_peak(nullptr);
    return *tflops > 0 ? GPE_OK : GPE_ERR_HIP;
}

int gpe_hbm_stream_peak(int device_id, double* gbs)
{
    if (!gbs || hipSetDevice(device_id) != hipSuccess)
        return GPE_ERR_HIP;
    *gbs = run_hbm_stream_peak(nullptr);
    return *gbs > 0 ? GPE_OK : GPE_ERR_HIP;
}

} // extern "C"
