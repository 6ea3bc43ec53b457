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


// ---------------------------------------------------------------------------------------------
// Blocked right-looking Cholesky, two levels (replaces Eigen::LLT at gp.hpp:565):
//   outer panels of `nbo` columns: the trailing update runs with k = nbo so that the matrix-core
//   kernel reads/writes C once per 2*nbo flops per element (k = 64 would be C-traffic bound);
//   inside a panel: 64-column steps  [k_diag: factor + invert | L21 = A21 X^T | in-panel update],
//   the last two being calls of the same matrix-core kernel.
// M >= N rows take part (rows N..M-1 = right-hand sides: they come out as (L^-1 b)^T).
// ---------------------------------------------------------------------------------------------
// Where the data-flow launches of a factorisation of order N (M >= N rows) begin: panels (k_panel256 + look-ahead updates)
// cover [0, e0), a tall launch [e0, t0) followed by one update with k = t0 - e0, the closing launch [t0, N64).
struct TailPlan {
    int64_t e0 = -1, t0 = -1, N64 = 0; // e0 < 0: no tall launch; t0 < 0: neither
    int64_t nt_tall = 0, nb_tall = 0, nt_tail = 0, nb_tail = 0;
    int64_t need_tall = 0, need_tail = 0; // doubles per buffer
};
static TailPlan tail_plan(const gpe_ctx* c, int64_t N, int64_t M)
{
    TailPlan pl;
    const int64_t nbo = c->nbo;
    pl.N64 = N / NB * NB;
    // Batched launches (k_tail_b: the members' tiles interleaved in one grid) take the data-flow launches only while all
    // members' tiles together stay within ~18 rounds of the chip: every member has 256 / G resident workgroups, and a tile
    // holds its CU from dispatch to its last store, mostly waiting — measured (profiles/r04_dispatch_order.log): 8 x N = 2048
    // 1.45 ms per batch against 1.59 through the step-by-step panels, but 64 x 2048 9.0 against 7.0 and 10 x 4096 7.9 against 7.6
    static const int64_t batch_tiles = getenv("GPE_BATCH_TAIL_TILES") ? atoll(getenv("GPE_BATCH_TAIL_TILES")) : 4608;
    int64_t tmax = g_batch.bt ? c->batch_tail_max : c->tail_max;
    if (!g_batch.bt && tmax >= 2 * NB && pl.N64 <= c->tail_single)
        tmax = std::max(tmax, pl.N64);
    if (!(tmax >= 2 * NB && c->panel256 && c->fuse_panel && c->panel_handover && nbo == 4 * NB && M - pl.N64 <= NB))
        return pl;
    const int64_t t0 = pl.N64 > tmax ? (pl.N64 - tmax + nbo - 1) / nbo * nbo : 0;
    if (pl.N64 - t0 < 2 * NB)
        return pl;
    const int64_t rs = M > pl.N64 ? 1 : 0;
    const int nt_tail = (int)((pl.N64 - t0) / NB), nb_tail = nt_tail + (int)rs;
    // the tall launch only from column 0 on: behind 256-column panels the look-ahead schedule in front of the closing launch is
    // the better one (measured, profiles/r04_schedule_ab.log: N = 5000 1.94 against 2.10 ms, 8192 5.14 against 5.34)
    const bool tall = t0 >= 2 * NB && c->tall_max >= 2 * NB && t0 <= c->tall_max;
    const int nt_tall = tall ? (int)(t0 / NB) : 0, nb_tall = tall ? (int)(pl.N64 / NB + rs) : 0;
    if (g_batch.bt) { // (a batch has no look-ahead panels: the data-flow launches cover the matrix from column 0 or not at all)
        if (t0 > 0 && !tall)
            return pl;
        if ((int64_t)g_batch.G * std::max(tail_tiles(nt_tail, nb_tail), tall ? tail_tiles(nt_tall, nb_tall) : (int64_t)0) > batch_tiles)
            return pl;
    }
    pl.t0 = t0;
    pl.nt_tail = nt_tail;
    pl.nb_tail = nb_tail;
    pl.need_tail = tail_buf_doubles(nt_tail, nb_tail);
    if (tall) {
        pl.e0 = 0;
        pl.nt_tall = nt_tall;
        pl.nb_tall = nb_tall;
        pl.need_tall = tail_buf_doubles(nt_tall, nb_tall);
    }
    return pl;
}

// test hook (gpe_debug_tail_plan): the plan for N samples, P outputs, a batch of G members (G <= 1: a single handle) under the
// given widths (<= 0: the defaults); no device is touched
static void debug_tail_plan(int64_t N, int P, int G, int64_t tail_max, int64_t tall_max, int64_t batch_tail_max, int64_t* out)
{
    gpe_ctx c;
    if (tail_max > 0) {
        c.tail_max = tail_max;
        c.tail_single = 0; // (as GPE_TAIL_MAX: the width given is the width used)
    }
    if (tall_max > 0)
        c.tall_max = tall_max;
    c.batch_tail_max = std::min(batch_tail_max > 0 ? batch_tail_max : c.batch_tail_max, c.tail_max);
    const BatchLaunch saved = g_batch;
    if (G > 1) {
        g_batch.G = G;
        g_batch.bt = reinterpret_cast<const BatchTab*>(1);
    }
    const TailPlan pl = tail_plan(&c, N, N + P);
    g_batch = saved;
    out[0] = pl.t0;
    out[1] = pl.e0;
    out[2] = pl.nt_tail;
    out[3] = pl.nb_tail;
    out[4] = pl.nt_tall;
    out[5] = pl.nb_tall;
    out[6] = pl.N64;
    out[7] = c.nbo;
}
// The hand-over buffers of handle c for this plan, on stream s (ordered in front of the launches that poll them).
// like != nullptr (a batched launch built from `like`'s pointers): same capacities and the same armed parity as that handle.
// ... the allocation part: a stream synchronisation, a free and a malloc when the buffers must grow.  compute_enqueue calls it
// BEFORE it enters the device's gate (the gate's mutex must not be held across a device synchronisation — every
// other host thread launching on the device would stall behind it); prepare_tail calls it again, then a no-op.
static bool reserve_tail(gpe_ctx* c, const TailPlan& pl, hipStream_t s, const gpe_ctx* like = nullptr)
{
    if (pl.t0 < 0)
        return true;
    int64_t want_tail = std::max(c->tail_cap, pl.need_tail), want_tall = std::max(c->tall_cap, pl.need_tall);
    if (like) {
        want_tail = like->tail_cap;
        want_tall = like->tall_cap;
        if (want_tail < pl.need_tail || want_tall < pl.need_tall)
            return false;
    }
    if (!c->dTail || c->tail_cap != want_tail || c->tall_cap != want_tall) {
        if (c->dTail) {
            hipStreamSynchronize(c->stream); // (an earlier launch of this handle may still be reading the old one)
            hipFree(c->dTail);
        }
        c->dTail = nullptr;
        c->tail_cap = c->tall_cap = 0;
        const size_t bytes = sizeof(double) * 2 * (size_t)(want_tail + want_tall);
        if (hipMalloc(&c->dTail, bytes) != hipSuccess)
            return false;
        hipMemsetAsync(c->dTail, 0xFF, bytes, s);
        c->tail_cap = want_tail;
        c->tall_cap = want_tall;
        c->tail_lay = c->tall_lay = -1;
    }
    return true;
}
static bool prepare_tail(gpe_ctx* c, const TailPlan& pl, hipStream_t s, const gpe_ctx* like = nullptr)
{
    if (pl.t0 < 0)
        return true;
    if (!reserve_tail(c, pl, s, like))
        return false;
    const int64_t lay_tail = pl.nt_tail * 65536 + pl.nb_tail, lay_tall = pl.e0 >= 0 ? pl.nt_tall * 65536 + pl.nb_tall : -1;
    if (c->tail_lay == -2 || (c->tail_lay >= 0 && c->tail_lay != lay_tail)
        || (like && c->tail_lay >= 0 && ((c->tail_count ^ like->tail_count) & 1))) {
        hipMemsetAsync(c->dTail, 0xFF, sizeof(double) * 2 * (size_t)c->tail_cap, s);
        c->tail_lay = -1;
    }
    if ((c->tall_lay == -2 && c->tall_cap > 0)
        || (pl.e0 >= 0
            && ((c->tall_lay >= 0 && c->tall_lay != lay_tall) || (like && c->tall_lay >= 0 && ((c->tall_count ^ like->tall_count) & 1))))) {
        hipMemsetAsync(c->dTail + 2 * c->tail_cap, 0xFF, sizeof(double) * 2 * (size_t)c->tall_cap, s);
        c->tall_lay = -1;
    }
    if (like) { // (a pair that is all-ones throughout may take any parity)
        c->tail_count = like->tail_count;
        // the tall pair follows member 0 only when this plan HAS a tall launch — only then was its parity checked
        // (and the pair re-armed) above.  A batch without one (N64 <= 1536) leaves the member's tall pair, layout and count
        // as its own last single-handle launch left them.
        if (pl.e0 >= 0)
            c->tall_count = like->tall_count;
    }
    return true;
}

void potrf_blocked(gpe_ctx* c, double* A, int64_t N, int64_t M)
{
    hipStream_t s = c->stream;
    const int64_t ld = c->ld;
    const int64_t nbo = c->nbo;
    bool next_diag_done = false; // the fused next-panel update factored the first diagonal block of the coming panel
    bool la_pending = false; // a bulk update is (possibly) still running on stream2
    size_t la_last = 0;
    // The last <= tail_max columns (all of them when N <= tail_max) go to ONE launch, a tiled data-flow factorisation
    // (potrf.hip: k_tail): the panels end at t0.  Its columns are whole 64-blocks: t0 .. N64; a ragged last block (N64 .. N,
    // fewer than 64 columns) and the right-hand-side rows ride in it as one more row strip and are finished by the panel code
    // below (one small update, the ragged block).  Round 4: up to tall_max columns in front of t0 are one launch of the same
    // kernel too (e0 .. t0, every row strip below riding along), followed by ONE update of everything behind t0 with
    // k = t0 - e0; 256-column panels with look-ahead only in front of e0 (none at N = 4096: three launches factor the matrix).
    TailPlan pl = tail_plan(c, N, M);
    if (pl.t0 >= 0 && !g_batch.bt && !prepare_tail(c, pl, s)) // (a batched launch: batch_enqueue_fused prepared every member)
        pl = TailPlan{};
    // gen_mode (compute_enqueue): the first data-flow launch generates its tiles of K itself — nobody built them
    TailGen gen{c->dXt, ld, N, c->dOm, ld, (c->flow_solve && (N + NB - 1) / NB <= 256) ? c->dAl : nullptr, ld, c->P, &c->kp};
    const int64_t t0 = pl.t0, e0 = pl.e0, N64 = pl.N64;
    const int64_t stop0 = e0 >= 0 ? e0 : t0; // where the panels end: the panel in front of it updates everything left in one piece
    for (int64_t p0 = 0; p0 < N; p0 += nbo) {
        if (e0 >= 0 && p0 == e0) {
            if (la_pending) {
                hipStreamWaitEvent(s, c->la_events[la_last], 0);
                la_pending = false;
            }
            {
                const double w = (double)(t0 - e0), h = (double)(M - e0);
                PhaseScope ps(c, GPE_PH_POTRF_TALL, w * w * w / 3.0 + (h - w) * w * w);
                double* pair = c->dTail + 2 * c->tail_cap;
                launch_tail(s, A, ld, e0, t0, N64, M, c->dXinv, c->dInfo, pair + (c->tall_count & 1) * c->tall_cap,
                            pair + ((c->tall_count + 1) & 1) * c->tall_cap, c->gen_mode == 2 ? &gen : nullptr);
                if (c->gen_mode == 2 && c->gen_ev) // the rest of K, built on the second stream beside this launch
                    hipStreamWaitEvent(s, c->gen_ev, 0);
                ++c->tall_count;
                c->tall_lay = pl.nt_tall * 65536 + pl.nb_tall;
            }
            { // everything behind t0 -= L[t0:M, e0:t0] L[t0:N, e0:t0]^T: one launch, k = t0 - e0
                GemmArgs g{};
                g.C = A + t0 + t0 * ld;
                g.ldc = ld;
                g.A = A + t0 + e0 * ld;
                g.lda = ld;
                g.B = A + t0 + e0 * ld;
                g.ldb = ld;
                g.m = M - t0;
                g.n = N - t0;
                g.k = t0 - e0;
                g.tri = 1;
                g.grow0 = t0;
                g.gcol0 = t0;
                g.rhs_rows = (int)(M - N);
                PhaseScope ps(c, GPE_PH_POTRF_UPDATE, gemm_flops(g));
                launch_gemm_sub(s, g);
            }
            p0 = t0;
            next_diag_done = false;
        }
        if (p0 == t0) {
            if (la_pending) {
                hipStreamWaitEvent(s, c->la_events[la_last], 0);
                la_pending = false;
            }
            {
                PhaseScope ps(c, GPE_PH_POTRF_TAIL, (double)(N64 - t0) * (N64 - t0) * (N64 - t0) / 3.0);
                launch_tail(s, A, ld, t0, N64, N64, M, c->dXinv, c->dInfo, c->dTail + (c->tail_count & 1) * c->tail_cap,
                            c->dTail + ((c->tail_count + 1) & 1) * c->tail_cap, c->gen_mode == 1 ? &gen : nullptr);
                ++c->tail_count;
                c->tail_lay = pl.nt_tail * 65536 + pl.nb_tail;
            }
            if (N64 == N)
                break;
            { // the ragged block and what lies under it: -= L[N64:M, t0:N64] L[N64:N, t0:N64]^T, then the panel code factors it
                GemmArgs g{};
                g.C = A + N64 + N64 * ld;
                g.ldc = ld;
                g.A = A + N64 + t0 * ld;
                g.lda = ld;
                g.B = A + N64 + t0 * ld;
                g.ldb = ld;
                g.m = M - N64;
                g.n = N - N64;
                g.k = N64 - t0;
                g.tri = 1;
                g.grow0 = N64;
                g.gcol0 = N64;
                PhaseScope ps(c, GPE_PH_POTRF_UPDATE, gemm_flops(g));
                // ONE tile with k up to 2816: dealt to up to 32 workgroups + an ordered fold (potrf.hip); its scratch is the pair of polled
                // buffers the closing launch has just used — dead until the next launch arms all of them again
                double* const used = c->dTail + ((c->tail_count - 1) & 1) * c->tail_cap;
                if (!launch_ragged_update(s, g.C, ld, g.A, ld, g.m, g.n, g.k, used, pl.need_tail))
                    launch_gemm_sub(s, g);
            }
            p0 = N64;
            next_diag_done = false;
        }
        const int64_t pw = std::min<int64_t>(nbo, N - p0);
        const int64_t pe = p0 + pw;
        bool diag_done = next_diag_done; // the previous fused step (or fused update) already factored this diagonal block
        next_diag_done = false;
        int nf = 0, nt0 = 0;    // fused steps of this panel and head tiles of the first one
        int64_t htile = 0;
        // head-tile scratch, two halves by panel parity: the copy into A is off the critical path
        // (nothing before the end of the factorisation reads those tiles of A) and may still be
        // pending on the second stream while the next panel is factored
        double* const Hbase = c->dHead + ((p0 / nbo) & 1) * (32 * NB * NB);
        // Will the trailing update of this panel be the fused launch that also factors the next panel's first
        // diagonal block (k_upd_fused)?  Then the steps of this panel pre-apply their pieces of that block.
        const bool fuse_diag = c->lookahead && !c->prof && std::min<int64_t>(pe + nbo, N) < N && c->fuse_panel && c->fuse_diag
            && c->stop_events && pw == nbo && nbo % NB == 0 && nbo >= 2 * NB && ld % 2 == 0
            && std::min<int64_t>(nbo, N - pe) % NB == 0 && pe != stop0;
        // the whole panel in one launch (potrf.hip: k_panel256): full 256 columns, head tiles and block inverses handed over
        // between its workgroups
        const bool p256 = c->panel256 && c->fuse_panel && c->panel_handover && !g_batch.bt && nbo == 4 * NB && pw == nbo && pe <= M;
        // In the first panels of a large factorisation the look-ahead stream is the longer one (N = 4096, panel 1: near + far
        // update 30 + 84 us against 54 + 18 us of chain) and the fused next-panel update, whose 155 KB workgroups need whole CUs,
        // ends up queued behind the far update of the panel before: releasing the stream when the PANEL is complete — its
        // updates need nothing from the fused update — starts every near/far pair one fused update earlier.
        hipEvent_t p_done = nullptr;
        if (p256 && fuse_diag && c->early_bulk >= 0) {
            const int64_t pe2_ = std::min<int64_t>(pe + nbo, N), pe3_ = std::min<int64_t>(pe2_ + nbo, N);
            const int64_t nt128 = (N - pe3_ + 127) / 128, far_tiles = nt128 * (nt128 + 1) / 2;
            if (pe3_ < N && far_tiles >= c->early_bulk) {
                const size_t kp = (size_t)(p0 / nbo);
                while (c->pl_events.size() <= kp) {
                    hipEvent_t e;
                    hipEventCreateWithFlags(&e, hipEventDisableTiming);
                    c->pl_events.push_back(e);
                }
                p_done = c->pl_events[kp];
            }
        }
        if (p256) {
            double* Xt = c->dXinv + (p0 / NB) * (NB * NB);
            if (!diag_done) {
                PhaseScope ps(c, GPE_PH_POTRF_PANEL, (double)NB * NB * NB);
                launch_diag(s, A + p0 + p0 * ld, ld, NB, Xt, c->dInfo, p0, 1);
            }
            PhaseScope ps(c, GPE_PH_POTRF_PANEL, (double)(M - p0 - NB) * NB * NB * 2.5 * 4);
            launch_panel256(s, A, ld, p0, M, Xt, c->dInfo, fuse_diag ? pe : -1, c->dHead + 64 * NB * NB,
                            c->dHead + ((c->p256_count & 1) * 32 + GPE_S22_TILE) * (NB * NB),
                            c->dHead + (((c->p256_count + 1) & 1) * 32 + GPE_S22_TILE) * (NB * NB), p_done);
            ++c->p256_count;
        }
        for (int64_t j0 = p0; j0 < pe && !p256; j0 += NB) {
            const int jb = (int)std::min<int64_t>(NB, pe - j0);
            const int64_t r0 = j0 + jb;
            double* Xt = c->dXinv + (j0 / NB) * (NB * NB);
            // fused step (k_panel_step): full 64-column blocks up to the end of the panel
            const int nt = (int)((pe - r0) / NB);
            const bool fuse = c->fuse_panel && jb == NB && (pe - r0) % NB == 0 && r0 < M && htile + nt <= 32;
            if (!diag_done) {
                PhaseScope ps(c, GPE_PH_POTRF_PANEL, (double)jb * jb * jb);
                launch_diag(s, A + j0 + j0 * ld, ld, jb, Xt, c->dInfo, j0, fuse ? 1 : 0);
            }
            diag_done = false;
            if (fuse) {
                PhaseScope ps(c, GPE_PH_POTRF_PANEL, (double)(M - r0) * NB * NB * (1 + nt));
                if (nf == 0)
                    nt0 = nt;
                // every step but the panel's first adds its own piece of the next panel's first diagonal block to the
                // scratch sum (the second step starts it); the third — whose workgroup there has the most slack —
                // also the first step's piece
                const bool pre = fuse_diag && j0 > p0;
                const int64_t dfirst_at = nbo >= 3 * NB ? p0 + 2 * NB : p0 + NB;
                launch_panel_step(s, A, ld, j0, M, nt, Xt, Xt + NB * NB, nt > 0 ? 1 : 0, c->dInfo, Hbase + htile * NB * NB,
                                  pre ? pe : -1, pre && j0 == dfirst_at ? p0 : -1, j0 == p0 + NB ? 1 : 0,
                                  c->dHead + 64 * NB * NB,
                                  c->panel_handover ? (gpe_epoch_t*)(c->dHead + 65 * NB * NB) + ((p0 / nbo) & 1) * 32 + htile : nullptr);
                htile += nt;
                if (nt > 0)
                    ++nf;
                diag_done = nt > 0;
                continue;
            }
            if (r0 < M) { // L21 = A21 L11^-T, in place (each 32-row workgroup reads only its own rows)
                GemmArgs g{};
                g.C = A + r0 + j0 * ld;
                g.ldc = ld;
                g.A = A + r0 + j0 * ld;
                g.lda = ld;
                g.a_kmajor = 0;
                g.B = Xt;
                g.ldb = NB;
                g.b_kmajor = 1; // opB(col, kk) = X[col][kk] = Xt[kk + 64 col]
                g.m = M - r0;
                g.n = jb;
                g.k = jb;
                g.overwrite = 1;
                g.tile = 32;
                PhaseScope ps(c, GPE_PH_POTRF_PANEL, (double)(M - r0) * jb * jb);
                launch_gemm_sub(s, g);
            }
            if (r0 < pe) { // rest of the panel's columns
                GemmArgs g{};
                g.C = A + r0 + r0 * ld;
                g.ldc = ld;
                g.A = A + r0 + j0 * ld;
                g.lda = ld;
                g.a_kmajor = 0;
                g.B = A + r0 + j0 * ld;
                g.ldb = ld;
                g.b_kmajor = 0;
                g.m = M - r0;
                g.n = pe - r0;
                g.k = jb;
                g.tri = 1;
                g.grow0 = r0;
                g.gcol0 = r0;
                PhaseScope ps(c, GPE_PH_POTRF_PANEL, gemm_flops(g));
                launch_gemm_sub(s, g);
            }
        }
        if (pe < N) { // trailing update, k = pw
            auto upd = [&](hipStream_t st, int64_t c0, int64_t c1, int64_t rlo, int grid_limit = 0,
                           hipEvent_t stop = nullptr, int tile = 0) {
                // C[rlo:M, c0:c1] -= L[rlo:M, p0:pe] L[c0:c1, p0:pe]^T   (elements on/below the diagonal)
                GemmArgs g{};
                g.C = A + rlo + c0 * ld;
                g.ldc = ld;
                g.A = A + rlo + p0 * ld;
                g.lda = ld;
                g.B = A + c0 + p0 * ld;
                g.ldb = ld;
                g.m = M - rlo;
                g.n = c1 - c0;
                g.k = pw;
                g.tri = 1;
                g.grow0 = rlo;
                g.gcol0 = c0;
                g.grid_limit = grid_limit;
                g.stop_event = stop;
                g.rhs_rows = (int)(M - N); // the appended obs_mean rows: FMAs inside the direct-to-LDS kernels, not a tile row
                if (grid_limit > 0)
                    g.tile = tile ? tile : 128; // the direct-to-LDS kernels are the ones that honour grid_limit
                PhaseScope ps(c, GPE_PH_POTRF_UPDATE, gemm_flops(g));
                launch_gemm_sub(st, g);
            };
            const int64_t pe2 = std::min<int64_t>(pe + nbo, N);
            if (c->lookahead && !c->prof && pe2 < N && pe != stop0) {
                // look-ahead: the next panel's columns are updated on the main stream, the rest of the
                // trailing matrix on the second stream while the next panel is factored
                auto ev = [&](size_t i) {
                    while (c->la_events.size() <= i) {
                        hipEvent_t e;
                        hipEventCreateWithFlags(&e, hipEventDisableTiming);
                        c->la_events.push_back(e);
                    }
                    return c->la_events[i];
                };
                // Events per outer panel kp: 3 kp = this panel's next-panel update done (the dispatch's own
                // completion signal: no marker packet on the critical stream), 3 kp + 1 = the bulk update has
                // finished the columns of panel kp + 2 ("near" part, done first), 3 kp + 2 = all of it.
                // The main stream only ever waits for a near part, which completed most of a panel earlier:
                // waiting for an event that fires just in time cost ~10 us per panel in the kernel trace.
                const size_t kp = (size_t)(p0 / nbo);
                if (la_pending)
                    hipStreamWaitEvent(s, ev(3 * (kp - 1) + 1), 0); // the previous bulk update also wrote these columns
                if (fuse_diag) {
                    // the update and, underneath it in the same launch, the factorisation of the next panel's
                    // first diagonal block (k_upd_fused): no k_diag launch at the head of the next panel
                    GemmArgs g{};
                    g.C = A + pe + pe * ld;
                    g.ldc = ld;
                    g.A = A + pe + p0 * ld;
                    g.lda = ld;
                    g.B = A + pe + p0 * ld;
                    g.ldb = ld;
                    g.m = M - pe;
                    g.n = pe2 - pe;
                    g.k = pw;
                    g.tri = 1;
                    g.grow0 = pe;
                    g.gcol0 = pe;
                    g.stop_event = ev(3 * kp);
                    launch_upd_fused(s, g, A, ld, pe, pe, c->dXinv + (pe / NB) * (NB * NB), c->dInfo,
                                     c->dHead + 64 * NB * NB); // the steps summed the pieces: no products here
                    next_diag_done = true;
                }
                else if (c->stop_events)
                    upd(s, pe, pe2, pe, 0, ev(3 * kp));
                else { // GPE_STOP_EVENT=0: a marker packet instead (rocprofv3's kernel trace delays dispatches
                       // that carry their own completion event by ~100 us; use this form under the profiler)
                    upd(s, pe, pe2, pe);
                    hipEventRecord(ev(3 * kp), s);
                }
                hipStreamWaitEvent(c->stream2, p_done ? p_done : ev(3 * kp), 0); // the bulk update starts now and shares the
                                                               // chip with panel kp + 1 only (p_done: and with this update)
                if (nf > 0 && !c->panel_handover) // (with the hand-over the head tiles were written in place too)
                    launch_head_copy(c->stream2, A, ld, p0, nt0, nf, Hbase);
                nf = 0;
                const int64_t pe3 = std::min<int64_t>(pe2 + nbo, N);
                upd(c->stream2, pe2, pe3, pe2, c->near_wgs >= 0 ? c->near_wgs : c->bulk_wgs, nullptr, 64); // near: what panel kp + 1's update needs
                hipEventRecord(ev(3 * kp + 1), c->stream2);
                if (pe3 < N) {
                    // 1 looping workgroup per CU on bulk_wgs CUs leaves 256 - bulk_wgs CUs to the panel.  When the update
                    // is many times longer than a panel (large trailing matrices: N = 16384 has 8 k tiles in its first
                    // ones) the reserve idles most of the time: above bulk_free_tiles tiles the update is dispatched
                    // unrestricted and the panel's workgroups take CUs as tiles retire (43.4 -> 34.6 ms at N = 16384)
                    const int64_t nt128 = (N - pe3 + 127) / 128, far_tiles = nt128 * (nt128 + 1) / 2;
                    upd(c->stream2, pe3, N, pe3, far_tiles >= c->bulk_free_tiles ? 0 : c->bulk_wgs);
                }
                hipEventRecord(ev(3 * kp + 2), c->stream2);
                la_pending = true;
                la_last = 3 * kp + 2;
            }
            else {
                if (la_pending) {
                    hipStreamWaitEvent(s, c->la_events[la_last], 0);
                    la_pending = false;
                }
                if (nf > 0 && !c->panel_handover)
                    launch_head_copy(s, A, ld, p0, nt0, nf, Hbase);
                nf = 0;
                upd(s, pe, N, pe);
            }
        }
        if (nf > 0 && !c->panel_handover) { // last panel: no trailing update
            PhaseScope ps(c, GPE_PH_POTRF_PANEL, 0.0);
            launch_head_copy(s, A, ld, p0, nt0, nf, Hbase);
            nf = 0;
        }
    }
    if (la_pending)
        hipStreamWaitEvent(s, c->la_events[la_last], 0);
}

// Z <- L^-1 B in place, B is N x M (ldb).  identity_structure: B starts as the identity, so at
// step j only columns < j + jb are non-zero (L^-1 is lower triangular) — gp.hpp:260 restricted
// to the triangle.  The 64-row diagonal solves are products with the stored block inverses.
void trsm_left_blocked(gpe_ctx* c, const double* L, double* B, int64_t ldb, int64_t N, int64_t M, bool ident, int ph)
{
    hipStream_t s = c->stream;
    const int64_t ld = c->ld;
    const int64_t nbo = c->nbo;
    for (int64_t o0 = 0; o0 < N; o0 += nbo) {
        const int64_t ow = std::min<int64_t>(nbo, N - o0);
        const int64_t oe = o0 + ow;
        for (int64_t j0 = o0; j0 < oe; j0 += NB) {
            const int jb = (int)std::min<int64_t>(NB, oe - j0);
            const int64_t r0 = j0 + jb;
            const int64_t ncol = ident ? r0 : M;
            {
                // B_j <- X_j B_j, in place: one 64-row tile, every workgroup owns its columns
                GemmArgs g{};
                g.C = B + j0;
                g.ldc = ldb;
                g.A = c->dXinv + (j0 / NB) * (NB * NB);
                g.lda = NB;
                g.a_kmajor = 1; // opA(i, kk) = X[i][kk] = Xt[kk + 64 i]
                g.B = B + j0;
                g.ldb = ldb;
                g.b_kmajor = 1; // opB(n, kk) = B[j0 + kk, n]
                g.m = jb;
                g.n = ncol;
                g.k = jb;
                g.overwrite = 1;
                g.tile = 64;
                PhaseScope ps(c, ph, (double)jb * jb * ncol);
                launch_gemm_sub(s, g);
            }
            if (r0 < oe) {
                GemmArgs g{};
                g.C = B + r0;
                g.ldc = ldb;
                g.A = L + r0 + j0 * ld;
                g.lda = ld;
                g.a_kmajor = 0;
                g.B = B + j0;
                g.ldb = ldb;
                g.b_kmajor = 1; // opB(n, kk) = B[j0 + kk, n]
                g.m = oe - r0;
                g.n = ncol;
                g.k = jb;
                PhaseScope ps(c, ph, gemm_flops(g));
                launch_gemm_sub(s, g);
            }
        }
        if (oe < N) {
            GemmArgs g{};
            g.C = B + oe;
            g.ldc = ldb;
            g.A = L + oe + o0 * ld;
            g.lda = ld;
            g.B = B + o0;
            g.ldb = ldb;
            g.b_kmajor = 1;
            g.m = N - oe;
            g.n = ident ? oe : M;
            g.k = ow;
            PhaseScope ps(c, ph, gemm_flops(g));
            launch_gemm_sub(s, g);
        }
    }
}

// One right-hand side, a single GP: the backward sweep whose hop is one matrix-vector product (sweep2.hip); false: not this
// shape — the caller takes k_trsv_bwd_flow
static bool bwd_chain_sweep(gpe_ctx* c, hipStream_t s, const double* y, int64_t ysi, double* al, int prefilled, const double* om, double* part)
{
    static const bool on = !(getenv("GPE_SWEEP_M") && atoi(getenv("GPE_SWEEP_M")) == 0);
    const int64_t nblk = (c->N + NB - 1) / NB;
    // (below eight blocks the two matrix-core products in front of the chain cost what the shorter hops save: N = 256 0.077 against 0.075 ms)
    if (!on || g_batch.bt || g_batch.G != 1 || nblk < 8 || nblk > 256)
        return false;
    launch_trsv_bwd_m(s, c->dA, c->ld, c->N, c->dXinv, y, ysi, al, c->dInfo + 1, prefilled, om, part);
    return true;
}

// GP::_compute_alpha (gp.hpp:605-611): alpha = L^-T (L^-1 obs_mean)
void solve_alpha(gpe_ctx* c)
{
    hipStream_t s = c->stream;
    c->ll_partials = 0;
    PhaseScope ps(c, GPE_PH_SOLVE, 2.0 * (double)c->N * c->N * c->P);
    const int64_t nblk = (c->N + NB - 1) / NB;
    const bool flow = c->flow_solve && nblk <= 256; // one data-flow launch per sweep instead of one launch per block
    for (int p0 = 0; p0 < c->P; p0 += GPE_MAX_P) {
        int pc = std::min(GPE_MAX_P, c->P - p0);
        const double* om = c->dOm + (int64_t)p0 * c->ld;
        double* al = c->dAl + (int64_t)p0 * c->ld;
        if (flow) {
            launch_trsv_fwd_flow(s, c->dA, c->ld, c->N, c->dXinv, om, c->ld, c->dY, c->ld, pc, c->dInfo + 1);
            if (!(c->P == 1 && bwd_chain_sweep(c, s, c->dY, 1, al, 0, om, c->hScal + 8)))
                launch_trsv_bwd_flow(s, c->dA, c->ld, c->N, c->dXinv, c->dY, 1, c->ld, al, c->ld, pc, c->dInfo + 1, 0, om, c->ld,
                                     c->hScal + 8, p0 > 0 ? 1 : 0);
            continue;
        }
        launch_copy2d(s, om, c->ld, c->dW, c->ld, c->N, pc);
        launch_trsv_sweep(s, c->dA, c->ld, c->N, c->dXinv, c->dW, c->dY, c->ld, pc, 0);
        launch_trsv_sweep(s, c->dA, c->ld, c->N, c->dXinv, c->dY, al, c->ld, pc, 1);
    }
    c->al_prefilled = false;
    c->ll_partials = flow ? (int)nblk : 0;
}

// second half of gp.hpp:605-611 when z = L^-1 obs_mean already sits in rows N.. of A
void solve_alpha_from_z(gpe_ctx* c)
{
    hipStream_t s = c->stream;
    PhaseScope ps(c, GPE_PH_SOLVE, (double)c->N * c->N * c->P);
    const int64_t nblk = (c->N + NB - 1) / NB;
    const bool flow = c->flow_solve && nblk <= 256; // every workgroup of the data-flow sweep must be resident
    for (int p0 = 0; p0 < c->P; p0 += GPE_MAX_P) {
        int pc = std::min(GPE_MAX_P, c->P - p0);
        if (flow) { // reads z straight from the appended rows, leaves the log-likelihood partial sums
            if (!(c->P == 1 && bwd_chain_sweep(c, s, c->dA + c->N, c->ld, c->dAl, c->al_prefilled ? 1 : 0, c->dOm, c->hScal + 8)))
                launch_trsv_bwd_flow(s, c->dA, c->ld, c->N, c->dXinv, c->dA + c->N + p0, c->ld, 1, c->dAl + (int64_t)p0 * c->ld,
                                     c->ld, pc, c->dInfo + 1, c->al_prefilled ? 1 : 0, c->dOm + (int64_t)p0 * c->ld, c->ld,
                                     c->hScal + 8, p0 > 0 ? 1 : 0);
        }
        else {
            launch_rows_to_cols(s, c->dA + c->N + p0, c->ld, c->N, pc, c->dY, c->ld);
            launch_trsv_sweep(s, c->dA, c->ld, c->N, c->dXinv, c->dY, c->dAl + (int64_t)p0 * c->ld, c->ld, pc, 1);
        }
    }
    c->al_prefilled = false;
    c->ll_partials = flow ? (int)nblk : 0;
}

void enqueue_loglik_terms(gpe_ctx* c)
{
    PhaseScope ps(c, GPE_PH_LOGLIK, 0.0);
    // flow path: gp.hpp:274-277 from the sweep's per-block partials, which it wrote straight into the pinned
    // host buffer (hScal + 8); they are added on the host in block order.  Nothing to enqueue.
    if (c->ll_partials == 0) {
        launch_loglik_terms(c->stream, c->dA, c->ld, c->N, c->dOm, c->dAl, c->ld, c->P, c->dScal);
        hipMemcpyAsync(c->hScal, c->dScal, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    }
}

static void inv2_start_early(gpe_ctx* c); // (below, with ensure_inv)
// One evaluation's chain on the device (defined with the gate, below).
struct ChainScope {
    gpe_ctx* c;
    bool on;
    int part = -1; // >= 0: the chain runs on that CU-masked partition's streams
    hipStream_t own = nullptr, own2 = nullptr;
    ChainScope(gpe_ctx* c_, bool engage, bool may_partition);
    ~ChainScope();
    ChainScope(const ChainScope&) = delete;
    ChainScope& operator=(const ChainScope&) = delete;
};
int compute_enqueue(gpe_ctx* c)
{
    if (c->N <= 0 || !c->dA)
        return GPE_ERR_STATE;
    // One evaluation's chain of launches as a unit behind the device's previous data-flow launch (dev.h: FlowGate; the gates of
    // the launches below nest inside this one): two handles evaluated from two threads run chain behind chain — 840
    // evaluations/s in all at N = 4096, where gating launch by launch interleaved their chains at 600.  Not for a batched
    // sequence: the two sub-batches of a batch of 64 overlap on purpose (their data-flow launches are still ordered one by one).
    // Round 5: when another chain is in flight on the device, this one goes to one of two CU-masked streams instead — half
    // of every XCD's CUs each — and the two run side by side (ChainScope, below).
    bool may_partition = false;
    if (!g_batch.bt) {
        const TailPlan pl0 = tail_plan(c, c->N, c->N + c->P);
        may_partition = pl0.t0 == 0 || (pl0.t0 > 0 && pl0.e0 == 0); // data-flow launches from column 0 on: no 256-column panels
        (void)reserve_tail(c, pl0, c->stream); // (a failure shows again, and is handled, where the buffers are prepared)
    }
    ChainScope gate(c, !g_batch.bt, may_partition);
    hipStream_t s = c->stream; // (the handle's own stream, or the partition's for the length of this enqueue)
    if (gate.part >= 0)
        c->inv_early = false; // (its events would make the own stream wait for a masked one: see ChainScope's destructor)
    digest_kernel(c);
    c->hInfo[0] = c->hInfo[1] = 0; // nothing of this handle is in flight here
    if (c->handover_off_left > 0 && --c->handover_off_left == 0)
        c->panel_handover = c->panel_handover_cfg; // re-armed after a run of clean evaluations without it
    const bool flow_al = c->flow_solve && (c->N + NB - 1) / NB <= 256;
    bool rows_done = false; // obs_mean^T under the matrix + the sweep's sentinel: by the build launch itself where it can
    // Round 4: where the data-flow launches begin decides whether K is built at all.  When the first of them starts at
    // column 0 it generates its tiles itself (potrf.hip: tail_gen_tile): for N <= 2560 the kernel matrix is never written,
    // for the tall launch of N = 4096 only the 2560 x 2560 block behind it is — beside the tall launch, on the second stream.
    c->gen_mode = 0;
    {
        // GPE_TAIL_GEN: 0 never; 1 (default) when ONE launch is the whole factorisation (N <= 2560: 0.489 -> 0.477 ms at
        // N = 2048, 0.254 -> 0.247 at 1024); 2 / 3: also the tall launch of N <= 4096, the block behind it built on the second
        // stream beside it / on the main stream in front of it — measured at N = 4096: 2 LOSES (1.253 -> 1.272 ms: the build
        // takes CUs from the first columns of the chain and the update then waits for an event)
        static const int gen_lvl = getenv("GPE_TAIL_GEN") ? atoi(getenv("GPE_TAIL_GEN")) : 1;
        const TailPlan pl = tail_plan(c, c->N, c->N + c->P);
        const int64_t N64 = c->N / NB * NB;
        if (gen_lvl > 0 && !c->host_K && !c->prof && pl.t0 >= 0 && (g_batch.bt || prepare_tail(c, pl, s))) {
            if (pl.t0 == 0 && (N64 == c->N || !g_batch.bt))
                c->gen_mode = 1;
            else if (pl.e0 == 0 && !g_batch.bt && gen_lvl >= 2)
                c->gen_mode = 2;
        }
    }
    if (c->host_K) {
        if (!c->dKhost)
            return GPE_ERR_STATE;
        PhaseScope ps(c, GPE_PH_KERNEL_BUILD, 0.0);
        launch_copy2d(s, c->dKhost, c->ld, c->dA, c->ld, c->N, c->N);
    }
    else if (c->gen_mode != 0) {
        project_lambda(c, s, c->dXt, c->ld, 0, c->N);
        // what is left to build: the ragged last block (mode 1) / everything behind the tall launch (mode 2), with
        // obs_mean's rows and the sweep's sentinel for those columns
        const TailPlan pl = tail_plan(c, c->N, c->N + c->P);
        const int64_t b0 = c->gen_mode == 1 ? c->N / NB * NB : pl.t0;
        rows_done = true;
        if (b0 < c->N) {
            const BuildRowsTail rt{c->dOm + b0, c->ld, c->P, c->dA + c->N + b0 * c->ld, flow_al ? c->dAl + b0 : nullptr, 0};
            hipStream_t sb = s;
            static const bool beside = !(getenv("GPE_TAIL_GEN") && atoi(getenv("GPE_TAIL_GEN")) == 3);
            if (c->gen_mode == 2 && c->kp.k_lam == 0 && beside) { // beside the tall launch (nothing of this handle is in flight on stream2)
                sb = c->stream2;
                if (!c->gen_ev)
                    hipEventCreateWithFlags(&c->gen_ev, hipEventDisableTiming);
            }
            if (!launch_build_K(sb, c->dXt + b0, c->ld, c->N - b0, c->kp, c->dA + b0 + b0 * c->ld, c->ld, &rt))
                launch_cols_to_rows(sb, c->dOm + b0, c->ld, c->N - b0, c->P, c->dA + c->N + b0 * c->ld, c->ld, flow_al ? c->dAl + b0 : nullptr);
            if (sb != s)
                hipEventRecord(c->gen_ev, sb);
            else if (c->gen_ev) { // (mode 2 on the main stream: no event to wait for)
                hipEventDestroy(c->gen_ev);
                c->gen_ev = nullptr;
            }
        }
        c->al_prefilled = flow_al;
    }
    else {
        PhaseScope ps(c, GPE_PH_KERNEL_BUILD, 0.0);
        project_lambda(c, s, c->dXt, c->ld, 0, c->N);
        const BuildRowsTail rt{c->dOm, c->ld, c->P, c->dA + c->N, flow_al ? c->dAl : nullptr, 0};
        static const bool tail = !(getenv("GPE_ROWS_TAIL") && atoi(getenv("GPE_ROWS_TAIL")) == 0);
        rows_done = launch_build_K(s, c->dXt, c->ld, c->N, c->kp, c->dA, c->ld, tail ? &rt : nullptr);
    }
    if (c->gen_mode == 0) {
        if (!rows_done)
            launch_cols_to_rows(s, c->dOm, c->ld, c->N, c->P, c->dA + c->N, c->ld, flow_al ? c->dAl : nullptr);
        c->al_prefilled = flow_al;
    }
    potrf_blocked(c, c->dA, c->N, c->N + c->P);
    c->have_L = true;
    c->inv_ok = false; // gp.hpp:570
    inv2_start_early(c);
    solve_alpha_from_z(c);
    enqueue_loglik_terms(c);
    return GPE_OK;
}

// Host wait for the stream.  A blocking hipStreamSynchronize costs a sleep/wake-up of the calling thread
// (tens of microseconds between back-to-back evaluations of a few milliseconds each); poll for up to
// 20 ms first, then block.  (Round 3 tried blocking straight away from the fifth concurrent waiter on — eight handles in
// flight lose throughput against four, 831 against 930 evaluations/s, and the pollers were the suspects: it made eight in
// flight slower still, 650-710/s.  Restarts that want to share the chip go through gpe_batch_hp_objective instead.)
static hipError_t wait_stream(hipStream_t s)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int i = 0; i < 64; ++i) {
            hipError_t e = hipStreamQuery(s);
            if (e != hipErrorNotReady)
                return e;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
            return hipStreamSynchronize(s);
    }
}

// A data-flow launch that ran into its bounded poll while the device was split into the two CU-masked halves (ChainScope): whatever
// the cause — a runtime that stopped honouring the masks, a queue mapping nobody has seen yet — the halves are given up for the
// rest of the process and evaluations go chain behind chain again (round 4's gate), which needs no assumption about masks.
std::atomic<bool> g_partitions_broken{false};
std::atomic<int> g_masked_chains{0}; // chains enqueued on a masked stream so far
static void partitions_give_up(const char* why)
{
    if (g_masked_chains.load() > 0 && !g_partitions_broken.exchange(true))
        fprintf(stderr, "limbo_amd: %s while evaluations shared the device in CU-masked halves: back to one chain at a time\n", why);
}

// host wait for the end of this handle's chain on a CU-masked stream (ChainScope): polls like wait_stream
static hipError_t wait_chain(gpe_ctx* c)
{
    if (!c->chain_pending)
        return hipSuccess;
    c->chain_pending = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int i = 0; i < 64; ++i) {
            hipError_t e = hipEventQuery(c->chain_ev);
            if (e != hipErrorNotReady)
                return e;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
            return hipEventSynchronize(c->chain_ev);
    }
}

// after a stream sync: did a data-flow sweep give up waiting for a producer?  With the dispatch-ordered block
// mapping (dev.h, flow_block_of) that is not a reachable state; the bounded poll stays as a backstop, and the host
// answers it by running the same work again with one launch per block (GPE_FLOW_FAULT=1 forces that path in tests).
static bool flow_failed(gpe_ctx* c)
{
    static const bool fault = getenv("GPE_FLOW_FAULT") && atoi(getenv("GPE_FLOW_FAULT")) != 0;
    const bool bad = c->hInfo[1] != 0 || (fault && c->flow_solve);
    if (c->hInfo[1] != 0)
        partitions_give_up("a sweep's hand-off timed out");
    c->hInfo[1] = 0;
    if (bad)
        ++c->flow_retries;
    return bad;
}

// scope in which the one-launch sweeps are off (the block-by-block re-run after a hand-off timeout)
struct NoFlowScope {
    gpe_ctx* c;
    bool saved;
    explicit NoFlowScope(gpe_ctx* c_) : c(c_), saved(c_->flow_solve) { c->flow_solve = false; }
    ~NoFlowScope() { c->flow_solve = saved; }
};

static void sum_ll_partials(gpe_ctx* c)
{
    if (c->ll_partials > 0) {
        long double sl = 0.0L, sa = 0.0L;
        for (int j = 0; j < c->ll_partials; ++j) {
            sl += c->hScal[8 + j];
            sa += c->hScal[8 + c->ll_partials + j];
        }
        c->hScal[0] = (double)sl;
        c->hScal[1] = (double)sa;
        c->ll_partials = 0;
    }
}

// redo: re-enqueues, with the one-launch sweeps off, everything that depended on a sweep of this call
template <class Redo> int compute_finish(gpe_ctx* c, Redo redo)
{
    HIPCHK(c, wait_chain(c));
    HIPCHK(c, wait_stream(c->stream));
    HIPCHK(c, hipGetLastError());
    drain_phases(c);
    if (c->hInfo[2] != 0) {
        // a wave of a panel step gave up waiting for a head tile (potrf.hip) — not a reachable state with workgroups
        // dispatched in index order; the bounded poll is a backstop, as for the sweeps.  The factor is unusable: run the
        // whole evaluation again, from K on, with every workgroup deriving the head tiles itself.
        c->hInfo[0] = c->hInfo[1] = c->hInfo[2] = 0;
        c->panel_handover = false;
        // this re-run and the next 16 evaluations re-derive the tiles, then hand over again — twice as many after every further
        // event in the process (a fixed back-off that re-arms for ever is a silent 1000x slowdown when the cause
        // persists), and ONE line on stderr the first time
        static std::atomic<int> events{0};
        const int ev = events.fetch_add(1);
        c->handover_off_left = (16 << std::min(ev, 14)) + 1;
        if (ev == 0)
            fprintf(stderr, "limbo_amd: a hand-over inside a data-flow launch timed out (another process on this GPU that does not take part in "
                            "/dev/shm/limbo_amd.gpu-*.lock, or a runtime that no longer dispatches workgroups in order): the evaluation was run again "
                            "without them; they stay off for 16 evaluations, twice as long after every further event (gpe_handover_reruns counts)\n");
        c->tail_lay = c->tall_lay = -2; // the data-flow launches' buffers are in an unknown state: all-ones again before their next use
        ++c->flow_retries;
        ++c->handover_reruns;
        partitions_give_up("a hand-over of the factorisation timed out");
        const BatchLaunch saved = g_batch;
        g_batch = BatchLaunch{};
        const int e = compute_enqueue(c);
        g_batch = saved;
        if (e != GPE_OK)
            return e;
        HIPCHK(c, wait_chain(c));
        HIPCHK(c, wait_stream(c->stream));
        HIPCHK(c, hipGetLastError());
        drain_phases(c);
    }
    if (flow_failed(c)) {
        NoFlowScope off(c);
        redo();
        HIPCHK(c, wait_stream(c->stream));
        HIPCHK(c, hipGetLastError());
        drain_phases(c);
        if (c->hInfo[1] != 0) { // cannot happen: no data-flow kernel ran
            c->hInfo[1] = 0;
            c->err = "triangular sweep failed twice";
            return GPE_ERR_HIP;
        }
    }
    sum_ll_partials(c);
    c->ll_ok = true;
    return *c->hInfo; // 0 or 1-based index of the first non-positive pivot
}
// the common case: alpha and the log-likelihood terms from L and obs_mean
int compute_finish(gpe_ctx* c)
{
    return compute_finish(c, [c] {
        solve_alpha(c);
        enqueue_loglik_terms(c);
    });
}

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

// What lies beyond the N x N part of U and of the T-form / W buffer must read as zero — the k ranges of a ragged order run to N
// rounded up to 64, and no launch ever writes there (every tile stores its valid part only).  Once per order and allocation.
static void inv2_zero_pads(gpe_ctx* c, hipStream_t s)
{
    if (!c->dLinv || !c->dInvS)
        return;
    if (c->inv_pad_n >= 0 && c->N >= c->inv_pad_n) { // (a larger N: its pads lie inside the pads that are zero already)
        c->inv_pad_n = c->N;
        return;
    }
    hipMemsetAsync(c->dLinv, 0, sizeof(double) * (size_t)(c->ld * c->cap), s);
    hipMemsetAsync(c->dInvS, 0, sizeof(double) * (size_t)(c->ld * c->cap), s);
    c->inv_pad_n = c->N;
}

// buffers and plan of the recursive K^-1 (inv2.hip) for the factor at hand
static int inv2_prepare(gpe_ctx* c)
{
    const int64_t ld = c->ld;
    if (!c->dLinv) {
        HIPCHK(c, hipMalloc(&c->dLinv, sizeof(double) * (size_t)(ld * c->cap)));
        c->inv_pad_n = -1; // (fresh memory: the pads of the recursive K^-1 are to be zero-filled)
    }
    if (!c->dKinv)
        HIPCHK(c, hipMalloc(&c->dKinv, sizeof(double) * (size_t)(ld * c->cap)));
    const int bufs_needed = g_batch.G >= 4 ? 1 : 1 + inv2_partials(); // (the plan of a batch of >= 4 cuts no k range: no partial buffers)
    if (c->dInvS && c->invS_bufs < bufs_needed) { // (a member of an earlier batch, now evaluated alone)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(c->dInvS);
        c->dInvS = nullptr;
    }
    if (!c->dInvS) {
        HIPCHK(c, hipMalloc(&c->dInvS, sizeof(double) * (size_t)(ld * c->cap) * (size_t)bufs_needed));
        c->invS_bufs = bufs_needed;
        c->inv_pad_n = -1; // (fresh memory: the pads of the recursive K^-1 are to be zero-filled)
    }
    Inv2Plan*& slot = g_batch.G >= 4 ? c->inv2_batched : c->inv2;
    bool rebuilt = false;
    slot = inv2_plan_get(slot, c->N, ld, c->dA, c->dLinv, c->dKinv, c->dInvS, ld * c->cap, g_batch.G, &rebuilt);
    (void)rebuilt;
    inv2_zero_pads(c, c->stream);
    if (!slot) {
        c->err = "K^-1: no memory for the plan of the recursion";
        return GPE_ERR_NOMEM;
    }
    return GPE_OK;
}

// gpe_hp_objective: the leaves and the lowest level of K^-1's recursion on the SECOND stream, behind the factorisation and
// beside the backward sweep of alpha (a 64-hop latency chain on 64 CUs: 115 us in which the chip is otherwise idle)
static void inv2_start_early(gpe_ctx* c)
{
    c->inv_prefix_done = false;
    if (!c->inv_early || c->prof || g_batch.bt || !c->stream2 || !inv2_supported(c->N) || inv2_prepare(c) != GPE_OK)
        return;
    if (!c->inv_ev) {
        hipEventCreateWithFlags(&c->inv_ev, hipEventDisableTiming);
        hipEventCreateWithFlags(&c->inv_ev0, hipEventDisableTiming);
    }
    hipEventRecord(c->inv_ev0, c->stream); // the factor is final
    hipStreamWaitEvent(c->stream2, c->inv_ev0, 0);
    inv2_run(c->stream2, c->inv2, c->dXinv, 1);
    hipEventRecord(c->inv_ev, c->stream2);
    c->inv_prefix_done = true;
}

int ensure_inv(gpe_ctx* c)
{
    if (c->inv_ok)
        return GPE_OK;
    if (!c->have_L)
        return GPE_ERR_STATE;
    hipStream_t s = c->stream;
    const int64_t N = c->N, ld = c->ld;
    if (!c->dLinv) {
        HIPCHK(c, hipMalloc(&c->dLinv, sizeof(double) * (size_t)(ld * c->cap)));
        c->inv_pad_n = -1; // (fresh memory: the pads of the recursive K^-1 are to be zero-filled)
    }
    if (!c->dKinv)
        HIPCHK(c, hipMalloc(&c->dKinv, sizeof(double) * (size_t)(ld * c->cap)));
    if (inv2_supported(N)) {
        // Round 5: the recursion of inv2.hip — a dozen launches of tile-product lists with k = 256 .. N / 2 instead of 48
        // launches of k = 256 (N >= 1024, ragged orders included; smaller ones keep the panel form below).  A batched sequence runs the
        // same lists for every member (gridDim.z; batch_enqueue_fused allocated every member's scratch).
        int rc = inv2_prepare(c);
        if (rc)
            return rc;
        {
            Inv2Plan* plan = g_batch.G >= 4 ? c->inv2_batched : c->inv2;
            PhaseScope ps(c, GPE_PH_INV, inv2_flops(plan));
            if (c->inv_prefix_done) { // (inv2_start_early — single handles only: the lowest level ran beside the sweep)
                hipStreamWaitEvent(s, c->inv_ev, 0);
                inv2_run(s, plan, c->dXinv, 2);
                c->inv_prefix_done = false;
            }
            else
                inv2_run(s, plan, c->dXinv, 0);
        }
        HIPCHK(c, hipGetLastError());
        c->inv_ok = true; // gp.hpp:263
        return GPE_OK;
    }
    static const bool inv_panels = !(getenv("GPE_INV_PANELS") && atoi(getenv("GPE_INV_PANELS")) == 0);
    c->inv_pad_n = -1; // (the forms below write whole tiles of the U buffer)
    if (inv_panels && c->nbo % 128 == 0 && c->nbo <= 256) {
        // Transposed formulation: U = L^-T (upper triangular) is built in dLinv, K^-1 = U U^T.  Every product below
        // is C (-/+)= A B^T with A and B contiguous along their non-k index — the operand layout of the LDS-direct
        // matrix-core kernel (gemm.hip) — and k = the panel width, the shape of the Cholesky trailing update.
        //   inv.hip        : X_p = inv(L_pp) for every outer panel p in one launch -> diagonal blocks of T (= the
        //                    K^-1 buffer, free until the last step), X_p^T -> diagonal blocks of U
        //   U[0:o0, p]     = AccT[0:o0, p] X_p^T                       (AccT = -sum_{q<p} U[:, q] L[p, q]^T, in T)
        //   AccT[0:oe, p+1..] -= U[0:oe, p] L[p+1.., p]^T
        //   K^-1 = U U^T                                              (one launch)
        const int64_t nbo = c->nbo;
        const int64_t npan = (N + nbo - 1) / nbo;
        // Round 3: K^-1 = sum_p U[:, p] U[:, p]^T is accumulated panel by panel on the SECOND stream while the main stream
        // is still building the later panels of U — that chain is a string of small dependent launches that leaves most
        // of the chip idle, and panel p's rank-k update only needs U's panel p.  (As a replacement for the one-launch
        // product the 16 accumulating launches were slower, 761 against 704 us: they re-read C; underneath the chain they
        // are free.)  The X_p now live in a compact side buffer, so the K^-1 buffer's diagonal blocks are free from the start.
        // Batched launches, profiling runs and GPE_INV_OVERLAP=0 keep everything on one stream, product last, as before.
        static const bool overlap_ok = !(getenv("GPE_INV_OVERLAP") && atoi(getenv("GPE_INV_OVERLAP")) == 0);
        const bool overlap = overlap_ok && !g_batch.bt && !c->prof && c->stop_events && npan >= 4;
        if (overlap && (int64_t)c->xp_cap < npan * nbo * nbo) {
            if (c->dXp)
                hipFree(c->dXp);
            c->dXp = nullptr;
            c->xp_cap = 0;
            HIPCHK(c, hipMalloc(&c->dXp, sizeof(double) * (size_t)(npan * nbo * nbo)));
            c->xp_cap = (size_t)(npan * nbo * nbo);
        }
        auto ev = [&](size_t i) {
            while (c->la_events.size() <= i) {
                hipEvent_t e;
                hipEventCreateWithFlags(&e, hipEventDisableTiming);
                c->la_events.push_back(e);
            }
            return c->la_events[i];
        };
        {
            PhaseScope ps(c, GPE_PH_INV, 0.0);
            // (a kernel: it takes part in batched launches, dev.h).  With the overlap only the second stream's rank-k updates
            // touch the K^-1 buffer: it is zeroed there, beside the block inverses instead of in front of them (21 us + a
            // launch boundary of every gradient evaluation); that stream's work of the factorisation was joined long ago, and
            // whatever read the buffer last on the main stream precedes the events those launches waited for
            launch_zero2d(overlap ? c->stream2 : s, c->dKinv, ld, N, N);
            if (overlap)
                launch_inv_panels(s, c->dA, ld, N, (int)nbo, c->dXinv, c->dXp, 0, c->dLinv, ld); // X_p compact, X_p^T -> U's diagonal blocks
            else
                launch_inv_panels(s, c->dA, ld, N, (int)nbo, c->dXinv, c->dKinv, ld, c->dLinv, ld);
        }
        auto rank_update = [&](hipStream_t st, int64_t o0, int64_t pw) { // K^-1[0:oe, 0:oe] += U[0:oe, p] U[0:oe, p]^T, lower triangle
            GemmArgs g{};
            g.C = c->dKinv;
            g.ldc = ld;
            g.A = c->dLinv + o0 * ld;
            g.lda = ld;
            g.B = c->dLinv + o0 * ld;
            g.ldb = ld;
            g.m = g.n = o0 + pw;
            g.k = pw;
            g.tri = 1;
            g.overwrite = 2;
            launch_gemm_sub(st, g);
        };
        if (overlap) {
            hipEventRecord(ev(0), s); // zeroed K^-1 buffer, block inverses: panel 0 of U is complete
            hipStreamWaitEvent(c->stream2, ev(0), 0);
            rank_update(c->stream2, 0, std::min<int64_t>(nbo, N));
        }
        for (int64_t o0 = 0; o0 < N; o0 += nbo) {
            const int64_t pw = std::min<int64_t>(nbo, N - o0), oe = o0 + pw;
            if (o0 > 0) {
                GemmArgs g{};
                g.C = c->dLinv + o0 * ld;
                g.ldc = ld;
                g.A = c->dKinv + o0 * ld;
                g.lda = ld;
                g.B = overlap ? c->dXp + (o0 / nbo) * (nbo * nbo) : c->dKinv + o0 + o0 * ld;
                g.ldb = overlap ? nbo : ld;
                g.m = o0;
                g.n = pw;
                g.k = pw;
                g.overwrite = 1;
                if (overlap)
                    g.stop_event = ev((size_t)(o0 / nbo)); // this launch's own completion: panel p of U is final
                PhaseScope ps(c, GPE_PH_INV, gemm_flops(g));
                launch_gemm_sub(s, g);
                if (overlap) {
                    hipStreamWaitEvent(c->stream2, ev((size_t)(o0 / nbo)), 0);
                    rank_update(c->stream2, o0, pw);
                }
            }
            if (oe < N) {
                GemmArgs g{};
                g.C = c->dKinv + oe * ld;
                g.ldc = ld;
                g.A = c->dLinv + o0 * ld;
                g.lda = ld;
                g.B = c->dA + oe + o0 * ld;
                g.ldb = ld;
                g.m = oe;
                g.n = N - oe;
                g.k = pw;
                PhaseScope ps(c, GPE_PH_INV, gemm_flops(g));
                launch_gemm_sub(s, g);
            }
        }
        if (overlap) {
            hipEventRecord(ev((size_t)npan), c->stream2);
            hipStreamWaitEvent(s, ev((size_t)npan), 0);
        }
        else {
            // K^-1 = U U^T (gp.hpp:261) in one launch, lower triangle, k from the tile diagonal on (U is upper
            // triangular).
            GemmArgs g{};
            g.C = c->dKinv;
            g.ldc = ld;
            g.A = c->dLinv;
            g.lda = ld;
            g.B = c->dLinv;
            g.ldb = ld;
            g.m = g.n = g.k = N;
            g.tri = 1;
            g.ktri = 1;
            g.overwrite = 1;
            PhaseScope ps(c, GPE_PH_INV, gemm_flops(g));
            launch_gemm_sub(s, g);
        }
    }
    else {
        {
            PhaseScope ps(c, GPE_PH_INV, 0.0);
            launch_set_identity(s, c->dLinv, ld, N);
        }
        trsm_left_blocked(c, c->dA, c->dLinv, ld, N, N, true, GPE_PH_INV); // L^-1 (gp.hpp:260)
        // K^-1 = L^-T L^-1 (gp.hpp:261), lower triangle, k range from the tile diagonal down
        GemmArgs g{};
        g.C = c->dKinv;
        g.ldc = ld;
        g.A = c->dLinv;
        g.lda = ld;
        g.a_kmajor = 1;
        g.B = c->dLinv;
        g.ldb = ld;
        g.b_kmajor = 1;
        g.m = g.n = g.k = N;
        g.tri = 1;
        g.ktri = 1;
        g.overwrite = 1;
        PhaseScope ps(c, GPE_PH_INV, gemm_flops(g));
        launch_gemm_sub(s, g);
    }
    c->inv_ok = true; // gp.hpp:263
    return GPE_OK;
}

static int ensure_loo_bufs(gpe_ctx* c, bool square)
{
    if (!c->dLooV)
        HIPCHK(c, hipMalloc(&c->dLooV, sizeof(double) * (size_t)(c->ld * (c->P + 2) + 8)));
    if (square && !c->dLooS)
        HIPCHK(c, hipMalloc(&c->dLooS, sizeof(double) * (size_t)(c->ld * c->cap)));
    return GPE_OK;
}

// Weights of the leave-one-out gradient (grad.hip header; gp.hpp:354-402): on return
//   dLooV[:, 0:P] = u = K^-1 (alpha / kappa),   dLinv (lower) = K^-1 diag(c) K^-1,   dLooV[ld (P+2)] = LOO value.
// dLinv (L^-1, only an intermediate of K^-1) is reused as the N x N output.
static int loo_weights(gpe_ctx* c)
{
    int rc = ensure_loo_bufs(c, true);
    if (rc)
        return rc;
    hipStream_t s = c->stream;
    const int64_t N = c->N, ld = c->ld;
    if (!c->dLinv) { // a clone that inherited K^-1 never ran ensure_inv's allocation
        HIPCHK(c, hipMalloc(&c->dLinv, sizeof(double) * (size_t)(ld * c->cap)));
        c->inv_pad_n = -1; // (fresh memory: the pads of the recursive K^-1 are to be zero-filled)
    }
    double *v = c->dLooV, *sc = c->dLooV + ld * c->P, *val = sc + ld, *outp = c->dLooV + ld * (c->P + 2);
    {
        PhaseScope ps(c, GPE_PH_GRAD, 0.0);
        launch_loo_prep(s, c->dKinv, ld, N, c->dAl, ld, c->P, v, sc, val, outp);
        const bool flow = c->flow_solve && (N + NB - 1) / NB <= 256;
        for (int p0 = 0; p0 < c->P; p0 += GPE_MAX_P) { // u = L^-T (L^-1 v), in place
            int pc = std::min(GPE_MAX_P, c->P - p0);
            if (flow) { // one launch per sweep
                launch_trsv_fwd_flow(s, c->dA, ld, N, c->dXinv, v + (int64_t)p0 * ld, ld, c->dY, ld, pc, c->dInfo + 1);
                launch_trsv_bwd_flow(s, c->dA, ld, N, c->dXinv, c->dY, 1, ld, v + (int64_t)p0 * ld, ld, pc, c->dInfo + 1, 0,
                                     nullptr, 0, nullptr, 0);
                continue;
            }
            launch_copy2d(s, v + (int64_t)p0 * ld, ld, c->dW, ld, N, pc);
            launch_trsv_sweep(s, c->dA, ld, N, c->dXinv, c->dW, c->dY, ld, pc, 0);
            launch_trsv_sweep(s, c->dA, ld, N, c->dXinv, c->dY, v + (int64_t)p0 * ld, ld, pc, 1);
        }
        launch_sym_colscale(s, c->dKinv, ld, N, sc, c->dLooS, ld);
    }
    GemmArgs g{};
    g.C = c->dLinv;
    g.ldc = ld;
    g.A = c->dLooS;
    g.lda = ld;
    g.B = c->dLooS;
    g.ldb = ld;
    g.m = g.n = g.k = N;
    g.tri = 1;
    g.overwrite = 1;
    PhaseScope ps(c, GPE_PH_GRAD, gemm_flops(g));
    launch_gemm_sub(s, g);
    return GPE_OK;
}

// scratch of the pair-sum kernel (grad.hip) + the T outputs behind it
static int ensure_grad_partial(gpe_ctx* c, int n_grad)
{
    const int64_t need = grad_partial_size(c->N, n_grad) + GPE_MAX_THETA + 8;
    if (need > c->grad_partial_cap) {
        if (c->dGradPartial)
            hipFree(c->dGradPartial);
        c->dGradPartial = nullptr;
        c->grad_partial_cap = 0;
        HIPCHK(c, hipMalloc(&c->dGradPartial, sizeof(double) * (size_t)need));
        c->grad_partial_cap = need;
    }
    return GPE_OK;
}

int grad_enqueue(gpe_ctx* c, int n_grad, int optimize_noise, bool loo = false)
{
    if (c->host_K)
        return GPE_ERR_UNSUPPORTED;
    if (n_grad != c->n_theta + (optimize_noise ? 1 : 0))
        return GPE_ERR_ARG;
    int rc = ensure_inv(c);
    if (rc)
        return rc;
    if (loo) {
        rc = loo_weights(c);
        if (rc)
            return rc;
    }
    rc = ensure_grad_partial(c, n_grad);
    if (rc)
        return rc;
    const int64_t need = grad_partial_size(c->N, n_grad) + GPE_MAX_THETA + 8;
    double* dgrad = c->dGradPartial + (need - GPE_MAX_THETA - 8);
    {
        PhaseScope ps(c, GPE_PH_GRAD, 0.0);
        // log-likelihood: w = alpha alpha^T - K^-1;  leave-one-out: w = sym(u alpha^T) - K^-1 diag(c) K^-1, times 2
        launch_grad_loglik(c->stream, c->dXt, c->ld, c->N, c->kp, loo ? c->dLinv : c->dKinv, c->ld, c->dAl, c->ld,
                           loo ? c->dLooV : c->dAl, c->P, c->n_theta, optimize_noise, c->dGradPartial, dgrad);
        if (loo)
            launch_scale_vec(c->stream, dgrad, n_grad, 2.0);
    }
    c->dGrad = dgrad;
    return GPE_OK;
}

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
// ---- one data-flow launch at a time per device (dev.h: FlowGate) ----
namespace {
struct GateDev {
    std::recursive_mutex mu;
    int depth = 0, next = 0;
    hipEvent_t ring[64] = {};
    hipStream_t last_stream = nullptr; // where the device's last UNMASKED data-flow launch went
    // Round 5: two CU-masked stream pairs, half of every XCD's CUs each (mask bit i = XCD i % 8, CU i / 8 of it — measured,
    // profiles/r05_cumask_probe.log; a mask cannot leave an XCD empty, so "four XCDs each" is not to be had).  A data-flow
    // launch confined to its half always finds its lowest unfinished workgroup resident there — per XCD the dispatcher hands
    // a launch's workgroups out in order, and nothing else that WAITS can hold those CUs — so one chain per half runs
    // deadlock-free beside the other.  An unmasked data-flow launch can hold any CU: it waits for both halves to drain, and
    // the masked chains that follow wait for it.
    hipStream_t part[2] = {nullptr, nullptr}, part_aux[2] = {nullptr, nullptr};
    bool part_tried = false, part_dirty[2] = {false, false};
    unsigned rr = 0;
    unsigned char* d_owner = nullptr; // device: owner[xcd * 256 + place] = the half (0 / 1) that (XCD, CU) place belongs to, 255 unknown
    int* h_violation = nullptr;       // pinned: set by a workgroup of a masked chain that found itself in the OTHER half
    std::chrono::steady_clock::time_point last_busy{}; // when a chain last found another one in flight (ChainScope)
    bool ever_busy = false;
    // Round 6: other PROCESSES on the same GPU.  The gate above orders the data-flow launches of this process by stream events;
    // two processes have no events in common, and two data-flow launches resident together starve each other exactly as two
    // streams did (bounded polls, full re-runs: 1 evaluation/s).  Two files per GPU under /dev/shm, named by its PCI bus id:
    //   .users  every process that has a handle on the GPU write-locks ONE byte of it for its lifetime (POSIX record lock: the
    //           kernel drops it when the process ends, however it ends); F_GETLK over the whole range answers "is anybody
    //           else here?" in one system call (a process's own locks never conflict with itself);
    //   .lock   flock(LOCK_EX) around a data-flow launch (or an evaluation's whole chain) AND the host wait for it, taken only
    //           while somebody else is here: data-flow launches of different processes then never overlap on the device.
    // A process that is alone pays one fcntl per launch scope and never touches the lock.
    int xp_users = -1, xp_lock = -1, xp_byte = -1;
    bool xp_tried = false, xp_held = false;
    std::chrono::steady_clock::time_point xp_attach{};
    bool xp_crowded_at_attach = false;
};
GateDev g_gate[16];
std::atomic<int> g_live[16]; // live handles per physical device (gpe_create / gpe_destroy)
std::atomic<long long> g_xproc_waits{0}; // data-flow scopes that ran under the inter-process lock (gpe_xproc_waits)
bool gate_on()
{
    static const bool on = !(getenv("GPE_FLOW_GATE") && atoi(getenv("GPE_FLOW_GATE")) == 0);
    return on;
}
GateDev& gate_dev()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    return g_gate[dev & 15];
}
} // namespace
namespace {
bool xproc_on()
{
    static const bool on = !(getenv("GPE_XPROC_LOCK") && atoi(getenv("GPE_XPROC_LOCK")) == 0);
    return on;
}
// is another process holding a byte of the users file?
bool xproc_others(GateDev& g)
{
    if (g.xp_users < 0)
        return false;
    struct flock fl {};
    fl.l_type = F_WRLCK;
    fl.l_whence = SEEK_SET;
    fl.l_start = 0;
    fl.l_len = 4096;
    return fcntl(g.xp_users, F_GETLK, &fl) == 0 && fl.l_type != F_UNLCK;
}
// once per process and device (under g.mu): open the two files, take a byte of the users file
void xproc_attach(GateDev& g)
{
    if (g.xp_tried)
        return;
    g.xp_tried = true;
    if (!xproc_on())
        return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), dev) != hipSuccess || !bus[0])
        return;
    for (char* p = bus; *p; ++p)
        if (*p == ':' || *p == '/')
            *p = '_';
    const char* dirs[2] = {"/dev/shm", "/tmp"};
    for (const char* d : dirs) {
        const std::string base = std::string(d) + "/limbo_amd.gpu-" + bus;
        const mode_t um = umask(0);
        const int fu = open((base + ".users").c_str(), O_RDWR | O_CREAT | O_CLOEXEC, 0666);
        const int fl = open((base + ".lock").c_str(), O_RDWR | O_CREAT | O_CLOEXEC, 0666);
        umask(um);
        if (fu >= 0 && fl >= 0) {
            g.xp_users = fu;
            g.xp_lock = fl;
            return;
        }
        if (fu >= 0)
            close(fu);
        if (fl >= 0)
            close(fl);
        g.xp_users = -1;
    }
}
// the process's first live handle on the device appears / its last one goes (gpe_create, gpe_destroy; under g.mu): a byte of
// the users file is held exactly while the process can have work on the GPU
void xproc_show(GateDev& g)
{
    xproc_attach(g);
    if (g.xp_users < 0 || g.xp_byte >= 0)
        return;
    g.xp_crowded_at_attach = xproc_others(g); // (before this process shows up in the file itself)
    for (int k = 0; k < 4096 && g.xp_byte < 0; ++k) { // a byte of my own, starting from my pid's
        struct flock fk {};
        fk.l_type = F_WRLCK;
        fk.l_whence = SEEK_SET;
        fk.l_start = (getpid() + k) % 4096;
        fk.l_len = 1;
        if (fcntl(g.xp_users, F_SETLK, &fk) == 0)
            g.xp_byte = (int)fk.l_start;
    }
    g.xp_attach = std::chrono::steady_clock::now();
}
void xproc_hide(GateDev& g)
{
    if (g.xp_users < 0 || g.xp_byte < 0)
        return;
    struct flock fk {};
    fk.l_type = F_UNLCK;
    fk.l_whence = SEEK_SET;
    fk.l_start = g.xp_byte;
    fk.l_len = 1;
    (void)fcntl(g.xp_users, F_SETLK, &fk);
    g.xp_byte = -1;
}
// outermost data-flow scope opens (under g.mu): take the inter-process lock while anybody else is on the GPU
void xproc_enter(GateDev& g)
{
    if (g.xp_lock < 0 || g.xp_byte < 0 || !xproc_others(g))
        return;
    // somebody who was here before me may have a launch in flight that it started believing it was alone: not before 5 ms
    // after I showed up in the users file (an evaluation is ~1 ms; it sees me from its next launch on)
    if (g.xp_crowded_at_attach) {
        const auto ready = g.xp_attach + std::chrono::milliseconds(5);
        if (std::chrono::steady_clock::now() < ready)
            std::this_thread::sleep_until(ready);
        g.xp_crowded_at_attach = false;
    }
    while (flock(g.xp_lock, LOCK_EX) != 0 && errno == EINTR) {
    }
    g.xp_held = true;
    g_xproc_waits.fetch_add(1, std::memory_order_relaxed);
    static std::atomic<bool> said{false};
    if (!said.exchange(true))
        fprintf(stderr, "limbo_amd: another process is using this GPU: data-flow launches take turns through %s (GPE_XPROC_LOCK=0 to disable)\n",
                "/dev/shm/limbo_amd.gpu-*.lock");
}
// ... closes: what was enqueued must be THROUGH on the device before the next process may start its own
void xproc_leave(GateDev& g, hipStream_t s, hipStream_t s2 = nullptr)
{
    if (!g.xp_held)
        return;
    (void)hipStreamSynchronize(s);
    if (s2)
        (void)hipStreamSynchronize(s2);
    g.xp_held = false;
    (void)flock(g.xp_lock, LOCK_UN);
}
} // namespace
// a stream is about to be destroyed: nobody may record on it afterwards
void flow_gate_forget(hipStream_t s)
{
    GateDev& g = gate_dev();
    std::lock_guard<std::recursive_mutex> lk(g.mu);
    if (g.last_stream == s)
        g.last_stream = nullptr; // (gpe_destroy synchronises the stream first: its launches are through)
}
// `s` waits for whatever is on `behind` now (an event at that stream's current end)
static void gate_order(GateDev& g, hipStream_t s, hipStream_t behind)
{
    hipEvent_t& e = g.ring[g.next];
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
        e = nullptr;
    if (e && hipEventRecord(e, behind) == hipSuccess) {
        (void)hipStreamWaitEvent(s, e, 0);
        g.next = (g.next + 1) % 64;
    }
}
// an unmasked data-flow launch on s: behind the device's previous one on another stream, and behind both masked halves
static void gate_unmasked(GateDev& g, hipStream_t s)
{
    // the previous data-flow launch of the device went to another stream: an event at that stream's current end (behind
    // that launch; nothing is recorded per launch — a batch of 64 members steps through ~60 gated launches on one stream)
    if (g.last_stream && g.last_stream != s)
        gate_order(g, s, g.last_stream);
}
// An unmasked data-flow launch can hold any CU: both masked halves must have drained before it.  A HOST wait (not an event the
// stream waits for: ChainScope's destructor says why) — and NOT under the gate's mutex (one thread's query beside
// threads running masked chains used to stall every other thread's enqueue for a whole chain): called with g.mu held ONCE by
// this thread (depth as it was before this scope), returns with it held again and both halves clean.
static void gate_drain_halves(GateDev& g)
{
    for (;;) {
        hipStream_t w[2];
        int nw = 0;
        for (int i = 0; i < 2; ++i)
            if (g.part_dirty[i]) {
                if (hipStreamQuery(g.part[i]) != hipErrorNotReady)
                    g.part_dirty[i] = false;
                else
                    w[nw++] = g.part[i];
            }
        if (nw == 0)
            return;
        g.mu.unlock();
        for (int k = 0; k < nw; ++k)
            (void)hipStreamSynchronize(w[k]);
        g.mu.lock(); // (others may have dirtied a half again meanwhile: look again)
    }
}
void flow_gate_enter(hipStream_t s)
{
    if (!gate_on())
        return;
    GateDev& g = gate_dev();
    g.mu.lock(); // (held until flow_gate_leave: the launch in between is a few microseconds of host time)
    if (g.depth == 0)
        gate_drain_halves(g);
    if (g.depth++ == 0) {
        xproc_enter(g);
        gate_unmasked(g, s);
    }
}
static bool partitions_on()
{
    static const bool on = !(getenv("GPE_FLOW_PARTITIONS") && atoi(getenv("GPE_FLOW_PARTITIONS")) == 0);
    return on && !g_partitions_broken.load(std::memory_order_relaxed);
}
// where a workgroup runs: XCD and (shader engine, array, CU) of it
__global__ void k_partition_probe(unsigned* __restrict__ out, int spin)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x] = ((xcc & 15u) << 16) | (hw & 0xFF00u); // CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { // stay resident for a moment so that the launch spreads over every CU it may use
    }
}
// do the two masked streams really confine their launches to disjoint halves of every XCD, here, in this process?
// (owner, optional: the half every place seen belongs to)
static bool partition_masks_hold(hipStream_t a, hipStream_t b, unsigned char* owner = nullptr)
{
    constexpr int G = 2048;
    unsigned* d = nullptr;
    if (hipMalloc(&d, sizeof(unsigned) * 2 * G) != hipSuccess)
        return false;
    hipLaunchKernelGGL(k_partition_probe, dim3(G), dim3(64), 0, a, d, 300);
    hipLaunchKernelGGL(k_partition_probe, dim3(G), dim3(64), 0, b, d + G, 300);
    std::vector<unsigned> h(2 * G);
    bool ok = hipStreamSynchronize(a) == hipSuccess && hipStreamSynchronize(b) == hipSuccess
        && hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * G, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok)
        return false;
    std::vector<unsigned> pa(h.begin(), h.begin() + G), pb(h.begin() + G, h.end());
    std::sort(pa.begin(), pa.end());
    pa.erase(std::unique(pa.begin(), pa.end()), pa.end());
    std::sort(pb.begin(), pb.end());
    pb.erase(std::unique(pb.begin(), pb.end()), pb.end());
    std::vector<unsigned> both;
    std::set_intersection(pa.begin(), pa.end(), pb.begin(), pb.end(), std::back_inserter(both));
    unsigned xa = 0, xb = 0; // XCDs seen
    for (unsigned v : pa)
        xa |= 1u << (v >> 16);
    for (unsigned v : pb)
        xb |= 1u << (v >> 16);
    if (owner) {
        for (unsigned v : pa)
            owner[(v >> 16) * 256 + ((v >> 8) & 255)] = 0;
        for (unsigned v : pb)
            owner[(v >> 16) * 256 + ((v >> 8) & 255)] = 1;
    }
    return both.empty() && !pa.empty() && !pb.empty() && pa.size() <= 128 && pb.size() <= 128 && xa == 0xFFu && xb == 0xFFu;
}
// at the head of every masked chain: 64 single-wave workgroups look where they are; one that sits in the other half's CUs says so
__global__ void k_partition_check(const unsigned char* __restrict__ owner, int half, int* __restrict__ violation)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const unsigned char o = owner[(xcc & 15u) * 256 + ((hw >> 8) & 255u)];
        if (o != 255 && o != (unsigned char)half)
            *violation = 1;
    }
}
static bool partition_streams(GateDev& g)
{
    if (!g.part_tried) {
        g.part_tried = true;
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus == 256) {
            uint32_t lo[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0}; // CUs 0..15 of every XCD
            uint32_t hi[8] = {0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}; // CUs 16..31
            bool ok = hipExtStreamCreateWithCUMask(&g.part[0], 8, lo) == hipSuccess && hipExtStreamCreateWithCUMask(&g.part_aux[0], 8, lo) == hipSuccess
                && hipExtStreamCreateWithCUMask(&g.part[1], 8, hi) == hipSuccess && hipExtStreamCreateWithCUMask(&g.part_aux[1], 8, hi) == hipSuccess;
            auto drop_streams = [&g] { // (a partial or rejected set of masked streams is destroyed, not leaked)
                for (hipStream_t* st : {&g.part[0], &g.part_aux[0], &g.part[1], &g.part_aux[1]}) {
                    if (*st)
                        (void)hipStreamDestroy(*st);
                    *st = nullptr;
                }
            };
            if (!ok)
                drop_streams();
            else {
                // the runtime creates a stream's hardware queue at its FIRST launch (tens of milliseconds for a masked one):
                // here, not inside the first evaluation that meets another one
                void* word = nullptr;
                if (hipMalloc(&word, 64) == hipSuccess) {
                    for (hipStream_t st : {g.part[0], g.part_aux[0], g.part[1], g.part_aux[1]}) {
                        (void)hipMemsetAsync(word, 0, 64, st);
                        (void)hipStreamSynchronize(st);
                    }
                    (void)hipFree(word);
                }
                // ... and the assumption everything rests on is CHECKED, in this process, on these streams: launches on the two
                // halves land on disjoint sets of at most 128 (XCD, CU) places, all eight XCDs each.  If not: no partitions.
                std::vector<unsigned char> owner(16 * 256, 255);
                if (!partition_masks_hold(g.part[0], g.part[1], owner.data()) || !partition_masks_hold(g.part_aux[0], g.part_aux[1])
                    || hipMalloc(&g.d_owner, owner.size()) != hipSuccess
                    || hipMemcpy(g.d_owner, owner.data(), owner.size(), hipMemcpyHostToDevice) != hipSuccess
                    || hipHostMalloc(&g.h_violation, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
                    fprintf(stderr, "limbo_amd: the CU masks of the chain partitions are not honoured here: one chain at a time\n");
                    drop_streams();
                }
                else
                    *g.h_violation = 0;
            }
        }
    }
    return g.part[0] != nullptr;
}
ChainScope::ChainScope(gpe_ctx* c_, bool engage, bool may_partition) : c(c_), on(engage && gate_on())
{
    if (!on)
        return;
    GateDev& g = gate_dev();
    g.mu.lock(); // (held for the enqueue of the evaluation: ~50 us of host time)
    ++g.depth;   // the gates of the launches inside nest in this one
    if (g.depth == 1)
        xproc_enter(g); // (another PROCESS on the GPU: this chain runs under the inter-process lock, on the whole chip)
    if (g.h_violation && *g.h_violation) { // a masked chain saw one of its workgroups in the other half's CUs
        *g.h_violation = 0;
        g_masked_chains.fetch_add(1);
        partitions_give_up("a CU mask was not honoured");
    }
    if (g.depth == 1 && may_partition && !g.xp_held && partitions_on() && !c->prof && partition_streams(g)) {
        // is another chain in flight on the device?  (A query, not a guarantee: it picks the mode; ORDER comes from the
        // events below.)
        const bool full_busy = g.last_stream && g.last_stream != c->stream && hipStreamQuery(g.last_stream) == hipErrorNotReady;
        bool busy[2];
        for (int i = 0; i < 2; ++i)
            busy[i] = g.part_dirty[i] && hipStreamQuery(g.part[i]) == hipErrorNotReady;
        // Hysteresis: with R threads in flight a chain now and then finds the device idle for a moment (the others are between
        // evaluations on the host); were it to take the whole chip, both halves would have to drain for it and the next
        // masked chains to wait behind it — measured: 701 evaluations/s with four threads instead of 930.  So the device
        // stays in two halves for 3 ms after a chain last found another one in flight (a caller that alternates handles from
        // ONE thread never finds a chain in flight: always the whole chip; so does whoever comes 3 ms after the threads).
        const auto now = std::chrono::steady_clock::now();
        if (full_busy || busy[0] || busy[1]) {
            g.last_busy = now;
            g.ever_busy = true;
        }
        if (g.ever_busy && now - g.last_busy < std::chrono::milliseconds(3))
            part = !busy[0] ? 0 : (!busy[1] ? 1 : (int)(g.rr++ & 1));
    }
    if (part >= 0) {
        hipStream_t P = g.part[part];
        gate_order(g, P, c->stream); // behind the handle's own earlier work (uploads, the previous evaluation's readers)
        if (g.last_stream && g.last_stream != c->stream)
            gate_order(g, P, g.last_stream); // behind the device's last unmasked data-flow launch
        own = c->stream;
        own2 = c->stream2;
        c->stream = P;
        c->stream2 = g.part_aux[part];
        g.part_dirty[part] = true;
        g_masked_chains.fetch_add(1, std::memory_order_relaxed);
        static const bool fault = getenv("GPE_PARTITION_FAULT") && atoi(getenv("GPE_PARTITION_FAULT")) != 0; // (test hook: claims the other half)
        hipLaunchKernelGGL(k_partition_check, dim3(64), dim3(64), 0, P, g.d_owner, fault ? 1 - part : part, g.h_violation);
    }
    else if (g.depth == 1) {
        --g.depth; // (the mutex is released while the halves drain: the scope is not open yet)
        gate_drain_halves(g);
        ++g.depth;
        gate_unmasked(g, c->stream);
    }
}
ChainScope::~ChainScope()
{
    if (!on)
        return;
    GateDev& g = gate_dev();
    if (part >= 0) {
        hipStream_t P = c->stream;
        c->stream = own;
        c->stream2 = own2;
        // Whatever the handle does next comes behind the chain — by a HOST wait (wait_chain, in compute_finish), not by making
        // the handle's own stream wait for an event of the masked one: own streams are high-priority queues (create_main_stream),
        // masked ones cannot be (hipExtStreamCreateWithCUMask takes no priority), and a high-priority queue that sits on a
        // barrier packet keeps the scheduler from the normal-priority queue it is waiting for once the process has more
        // hardware queues than the chip maps at a time — measured with GPU_MAX_HW_QUEUES=8, torch in the process and eight
        // threads: masked chains got no service for seconds, their bounded polls fired (3 evaluations/s, re-runs); with the
        // own streams at the default priority, or with this, 950 evaluations/s.
        if (!c->chain_ev && hipEventCreateWithFlags(&c->chain_ev, hipEventDisableTiming) != hipSuccess)
            c->chain_ev = nullptr;
        if (c->chain_ev && hipEventRecord(c->chain_ev, P) == hipSuccess)
            c->chain_pending = true;
        else
            (void)hipStreamSynchronize(P);
    }
    else if (g.depth == 1)
        g.last_stream = c->stream;
    if (g.depth == 1)
        xproc_leave(g, c->stream, c->stream2);
    --g.depth;
    g.mu.unlock();
}
void flow_gate_leave(hipStream_t s)
{
    if (!gate_on())
        return;
    GateDev& g = gate_dev();
    if (--g.depth == 0) {
        g.last_stream = s;
        xproc_leave(g, s);
    }
    g.mu.unlock();
}

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

// The batched query (gp.hpp:613-632 for M points) with the POINTS along the contiguous axis.
//   Kst[m + i ldq] = k(x_i, v_m)  (the cross kernel, transposed);   Zt = Kst L^-T, panel by panel:
//     Zt[:, p]       = Acc[:, p] X_p^T                      X_p = inv(L_pp), all panels by one launch (inv.hip)
//     Acc[:, p+1 ..] -= Zt[:, p] L[p+1 .., p]^T             k = the panel width
//   var[m] = k(v_m, v_m) - sum_i Zt[m, i]^2,   kta[m, p] = sum_i Kst[m, i] alpha[i, p].
// In this layout EVERY product is C (-)= A B^T with both operands contiguous along their non-k index — the operand form
// of the direct-to-LDS matrix-core kernel (gemm.hip, k_gemm_glds), as K^-1's U = L^-T (ensure_inv).  The N x M layout
// (rounds 1-2, still the path for caller-supplied cross kernels) has the right-hand sides k-contiguous and ran its
// M N^2 flops through the register-staged kernel: 38 TFLOP/s at N = 16384 against the 53 of the factorisation's updates.
static int query_transposed(gpe_ctx* c, const double* Xq, int64_t M, double* kta, double* var)
{
    hipStream_t s = c->stream;
    const int64_t N = c->N, ld = c->ld, nbo = c->nbo;
    const int D = c->D, P = c->P;
    // chunk: two ldq x N buffers of <= 2 GiB each
    int64_t mc_max = std::max<int64_t>(64, (((int64_t)1 << 28) / std::max<int64_t>(N, 1)) / 64 * 64);
    mc_max = std::min<int64_t>(mc_max, round_up(M, 64));
    const int64_t ldq = mc_max + 16; // not a power of two (HBM channel camping on column strides), even, 16-byte rows
    const int64_t npan = (N + nbo - 1) / nbo;
    const int nseg = (int)std::max<int64_t>(1, std::min<int64_t>(32, N / 512));
    const size_t n_qrm = (size_t)(mc_max * std::max(D, 1)), n_qt = (size_t)(ldq * std::max(xt_rows(D), 1));
    const size_t n_mat = (size_t)(ldq * N), n_xp = (size_t)(npan * nbo * nbo), n_part = (size_t)nseg * std::max(P, 1) * (size_t)ldq;
    const size_t n_kta = (size_t)(mc_max * P);
    const size_t need = sizeof(double) * (n_qrm + n_qt + 2 * n_mat + n_xp + n_part + n_kta + 2 * (size_t)mc_max);
    if (need > c->query_bytes) {
        if (c->dQuery)
            hipFree(c->dQuery);
        c->dQuery = nullptr;
        c->query_bytes = 0;
        HIPCHK(c, hipMalloc(&c->dQuery, need));
        c->query_bytes = need;
    }
    double* dQrm = c->dQuery;
    double* dQt = dQrm + n_qrm;
    double* dKst = dQt + n_qt; // the cross kernel, then the running right-hand side Acc
    double* dZt = dKst + n_mat;
    double* dXp = dZt + n_mat;
    double* dPart = dXp + n_xp;
    double* dKta = dPart + n_part;
    double* dVar = dKta + n_kta;
    double* dKvv = dVar + mc_max;
    int rc = GPE_OK;
    // The tile of every product below is picked from N alone, never from the batch: the 128 x 128 and the 64 x 64 kernels round
    // differently in the last bit (measured, round 6: a point's variance moved by 1.5e-15 with the size of the batch it was asked
    // in, because launch_gemm_sub picks the tile from the live-tile count = from mc).  A point's answer must not depend on the
    // batch around it (tests/test_gpu_configs.py: the 100 000-point batch of configs[2] bitwise equal to chunks of 4096).
    const int qtile = N >= 1024 ? 128 : 64;
    if (var) {
        PhaseScope ps(c, GPE_PH_QUERY, 0.0);
        launch_inv_panels(s, c->dA, ld, N, (int)nbo, c->dXinv, dXp, 0, nullptr, 0); // X_p of every panel, compact
    }
    for (int64_t m0 = 0; m0 < M && rc == GPE_OK; m0 += mc_max) {
        const int64_t mc = std::min<int64_t>(mc_max, M - m0);
        hipMemcpyAsync(dQrm, Xq + m0 * D, sizeof(double) * (size_t)(mc * D), hipMemcpyHostToDevice, s);
        launch_transpose_x(s, dQrm, mc, D, dQt, ldq, 0);
        project_lambda(c, s, dQt, ldq, 0, mc);
        {
            // k is symmetric: the cross kernel with the roles of samples and points exchanged IS the transposed block
            PhaseScope ps(c, GPE_PH_QUERY, 0.0);
            launch_build_Ks(s, dQt, ldq, mc, c->dXt, ld, N, c->kp, dKst, ldq); // gp.hpp:626-632
        }
        if (kta) {
            PhaseScope ps(c, GPE_PH_QUERY, 2.0 * N * mc * P);
            launch_kta_t(s, dKst, ldq, N, mc, c->dAl, ld, P, dKta, mc_max, dPart, ldq, nseg); // gp.hpp:615
            for (int p = 0; p < P; ++p)
                hipMemcpyAsync(kta + m0 + (int64_t)p * M, dKta + (int64_t)p * mc_max, sizeof(double) * (size_t)mc,
                               hipMemcpyDeviceToHost, s);
        }
        if (var) {
            for (int64_t o0 = 0; o0 < N; o0 += nbo) { // gp.hpp:620, transposed
                const int64_t pw = std::min<int64_t>(nbo, N - o0), oe = o0 + pw;
                {
                    GemmArgs g{};
                    g.C = dZt + o0 * ldq;
                    g.ldc = ldq;
                    g.A = dKst + o0 * ldq;
                    g.lda = ldq;
                    g.B = dXp + (o0 / nbo) * (nbo * nbo);
                    g.ldb = nbo;
                    g.m = mc;
                    g.n = pw;
                    g.k = pw;
                    g.overwrite = 1;
                    g.tile = qtile;
                    PhaseScope ps(c, GPE_PH_QUERY, gemm_flops(g));
                    launch_gemm_sub(s, g);
                }
                if (oe < N) {
                    GemmArgs g{};
                    g.tile = qtile;
                    g.C = dKst + oe * ldq;
                    g.ldc = ldq;
                    g.A = dZt + o0 * ldq;
                    g.lda = ldq;
                    g.B = c->dA + oe + o0 * ld;
                    g.ldb = ld;
                    g.m = mc;
                    g.n = N - oe;
                    g.k = pw;
                    PhaseScope ps(c, GPE_PH_QUERY, gemm_flops(g));
                    launch_gemm_sub(s, g);
                }
            }
            PhaseScope ps(c, GPE_PH_QUERY, 2.0 * N * mc);
            launch_kvv(s, dQt, ldq, mc, c->kp, dKvv);
            launch_row_var_t(s, dZt, ldq, N, mc, dKvv, dVar, dPart, ldq, nseg); // gp.hpp:621
            hipMemcpyAsync(var + m0, dVar, sizeof(double) * (size_t)mc, hipMemcpyDeviceToHost, s);
        }
        if (hipStreamSynchronize(s) != hipSuccess) {
            c->err = "query_batch: stream sync failed";
            rc = GPE_ERR_HIP;
        }
    }
    drain_phases(c);
    if (c->query_bytes > ((size_t)64 << 20)) { // a large batch: give the memory back
        hipFree(c->dQuery);
        c->dQuery = nullptr;
        c->query_bytes = 0;
    }
    return rc;
}

// shared by gpe_query_batch (cross kernel built on the device from Xq) and gpe_query_batch_cross
// (cross kernel handed over by the caller): kta = Ks^T alpha, var = kvv - colsum((L^-1 Ks)^2)
static int query_impl(gpe_ctx* c, const double* Xq, const double* KsHost, int64_t M, double* kta, double* var)
{
    hipStream_t s = c->stream;
    digest_kernel(c);
    const int64_t N = c->N, ld = c->ld;
    const int D = c->D, P = c->P;
    if (c->small_path && Xq && N <= small_max_n() && M <= 8 && (int64_t)M * D <= 1024 && P <= GPE_MAX_P) {
        // the per-point query of an acquisition functor on a small GP: one launch (one workgroup per point), the
        // points read from and the results written to pinned host memory (small.hip)
        memcpy(c->hSmall + 256, Xq, sizeof(double) * (size_t)(M * D));
        SmallQueryArgs q{};
        q.L = c->dA;
        q.ld = ld;
        q.Xinv = c->dXinv;
        q.Xt = c->dXt;
        q.ldx = ld;
        q.Al = c->dAl;
        q.P = P;
        q.n = (int)N;
        q.M = (int)M;
        q.D = D;
        q.xq_host = c->hSmall + 256;
        q.kta_host = c->hSmall + 16;
        q.var_host = c->hSmall + 16 + 8 * GPE_MAX_P;
        q.seq = c->hSmallSeq;
        q.seq_val = ++c->small_seq;
        q.want_kta = kta ? 1 : 0;
        q.want_var = var ? 1 : 0;
        {
            PhaseScope ps(c, GPE_PH_QUERY, (double)N * N * M);
            launch_small_query(s, q, c->kp, lam_params(c));
        }
        ++c->small_calls;
        int rc = small_wait(c, (int)M, q.seq_val);
        drain_phases(c);
        if (rc)
            return rc;
        if (kta)
            memcpy(kta, q.kta_host, sizeof(double) * (size_t)(M * P));
        if (var)
            memcpy(var, q.var_host, sizeof(double) * (size_t)M);
        return GPE_OK;
    }
    // a handful of points (the per-point calls of an acquisition functor, gp.hpp:159-191): the forward
    // substitution runs as ONE data-flow launch (k_trsv_fwd_flow, <= GPE_MAX_P right-hand sides) instead of a
    // blocked matrix solve, whose dependent matrix-core launches are all launch floor there
    static const bool sweep_ok0 = !(getenv("GPE_QUERY_SWEEP") && atoi(getenv("GPE_QUERY_SWEEP")) == 0);
    static const bool transposed_ok = !(getenv("GPE_QUERY_T") && atoi(getenv("GPE_QUERY_T")) == 0);
    const bool few0 = sweep_ok0 && c->flow_solve && M <= GPE_MAX_P && (N + NB - 1) / NB <= 256;
    if (Xq && !few0 && transposed_ok && c->nbo % 128 == 0 && c->nbo <= 256 && N >= c->nbo)
        return query_transposed(c, Xq, M, kta, var);
    // chunk so that the N x mc cross matrix stays under ~2 GiB
    int64_t mc_max = std::max<int64_t>(64, ((int64_t)1 << 28) / std::max<int64_t>(ld, 1));
    mc_max = round_up(std::min<int64_t>(mc_max, round_up(M, 64)), 64);
    // a handful of points (the per-point calls of an acquisition functor, gp.hpp:159-191): the forward
    // substitution runs as ONE data-flow launch (k_trsv_fwd_flow, <= GPE_MAX_P right-hand sides) instead of the
    // blocked matrix solve, whose ~2 N/64 dependent matrix-core launches are all launch floor here
    static const bool sweep_ok = !(getenv("GPE_QUERY_SWEEP") && atoi(getenv("GPE_QUERY_SWEEP")) == 0);
    const bool few = sweep_ok && c->flow_solve && M <= GPE_MAX_P && (N + NB - 1) / NB <= 256;
    if (few)
        mc_max = GPE_MAX_P;
    const int64_t ldq = mc_max;
    // one allocation, carved up; kept across calls while small so that point queries do not malloc/free
    const size_t n_qrm = (size_t)(mc_max * std::max(D, 1)), n_qt = (size_t)(ldq * std::max(xt_rows(D), 1));
    const size_t n_ks = (size_t)(ld * mc_max), n_z = few ? n_ks : 0, n_kta = (size_t)(mc_max * P);
    const size_t need = sizeof(double) * (n_qrm + n_qt + n_ks + n_z + n_kta + 2 * (size_t)mc_max);
    if (need > c->query_bytes) {
        if (c->dQuery)
            hipFree(c->dQuery);
        c->dQuery = nullptr;
        c->query_bytes = 0;
        HIPCHK(c, hipMalloc(&c->dQuery, need));
        c->query_bytes = need;
    }
    double* dQrm = c->dQuery;
    double* dQt = dQrm + n_qrm;
    double* dKs = dQt + n_qt;
    double* dZ = dKs + n_ks;
    double* dKta = dZ + n_z;
    double* dVar = dKta + n_kta;
    double* dKvv = dVar + mc_max;
    int rc = GPE_OK;
    for (int64_t m0 = 0; m0 < M && rc == GPE_OK; m0 += mc_max) {
        const int64_t mc = std::min<int64_t>(mc_max, M - m0);
        if (Xq) {
            hipMemcpyAsync(dQrm, Xq + m0 * D, sizeof(double) * (size_t)(mc * D), hipMemcpyHostToDevice, s);
            launch_transpose_x(s, dQrm, mc, D, dQt, ldq, 0);
            project_lambda(c, s, dQt, ldq, 0, mc);
            PhaseScope ps(c, GPE_PH_QUERY, 0.0);
            launch_build_Ks(s, c->dXt, ld, N, dQt, ldq, mc, c->kp, dKs, ld); // gp.hpp:626-632
        }
        else {
            hipMemcpy2DAsync(dKs, sizeof(double) * ld, KsHost + m0 * N, sizeof(double) * N, sizeof(double) * N, mc,
                             hipMemcpyHostToDevice, s);
        }
        if (kta) {
            PhaseScope ps(c, GPE_PH_QUERY, 2.0 * N * mc * P);
            launch_kta(s, dKs, ld, N, mc, c->dAl, ld, P, dKta, mc_max); // gp.hpp:615
            for (int p = 0; p < P; ++p)
                hipMemcpyAsync(kta + m0 + (int64_t)p * M, dKta + (int64_t)p * mc_max, sizeof(double) * (size_t)mc,
                               hipMemcpyDeviceToHost, s);
        }
        if (var) {
            const double* Z = dKs;
            if (few) {
                PhaseScope ps(c, GPE_PH_QUERY, (double)N * N * mc);
                launch_trsv_fwd_flow(s, c->dA, ld, N, c->dXinv, dKs, ld, dZ, ld, (int)mc, c->dInfo + 1); // gp.hpp:620
                Z = dZ;
            }
            else
                trsm_left_blocked(c, c->dA, dKs, ld, N, mc, false, GPE_PH_QUERY); // gp.hpp:620
            PhaseScope ps(c, GPE_PH_QUERY, 2.0 * N * mc);
            if (Xq)
                launch_kvv(s, dQt, ldq, mc, c->kp, dKvv);
            else
                hipMemsetAsync(dKvv, 0, sizeof(double) * (size_t)mc, s);
            launch_col_var(s, Z, ld, N, mc, dKvv, dVar); // gp.hpp:621
            hipMemcpyAsync(var + m0, dVar, sizeof(double) * (size_t)mc, hipMemcpyDeviceToHost, s);
        }
        if (hipStreamSynchronize(s) != hipSuccess) {
            c->err = "query_batch: stream sync failed";
            rc = GPE_ERR_HIP;
        }
        else if (few && var && flow_failed(c)) {
            // the one-launch sweep gave up (never expected): the same chunk through the blocked solve, in place
            trsm_left_blocked(c, c->dA, dKs, ld, N, mc, false, GPE_PH_QUERY);
            launch_col_var(s, dKs, ld, N, mc, dKvv, dVar);
            hipMemcpyAsync(var + m0, dVar, sizeof(double) * (size_t)mc, hipMemcpyDeviceToHost, s);
            if (hipStreamSynchronize(s) != hipSuccess) {
                c->err = "query_batch: stream sync failed";
                rc = GPE_ERR_HIP;
            }
        }
    }
    drain_phases(c);
    if (c->query_bytes > ((size_t)64 << 20)) { // a large batch: give the memory back
        hipFree(c->dQuery);
        c->dQuery = nullptr;
        c->query_bytes = 0;
    }
    return rc;
}

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

// Can these two GPs be stepped by the same launches (dev.h, BatchTab)?  Same shape, same schedule, device code for K.
static bool batch_compatible(const gpe_ctx* a, const gpe_ctx* b)
{
    return a->device == b->device && a->N == b->N && a->D == b->D && a->P == b->P && a->cap == b->cap && a->ld == b->ld
        && a->nbo == b->nbo && a->fuse_panel == b->fuse_panel && a->flow_solve == b->flow_solve && !a->host_K && !b->host_K
        && a->kind != GPE_KERNEL_HOST_K && b->kind != GPE_KERNEL_HOST_K && a->n_theta == b->n_theta
        && ((a->kind == GPE_KERNEL_SE_ARD) == (b->kind == GPE_KERNEL_SE_ARD)) && a->dA && b->dA && !a->prof && !b->prof;
}

// gpe_compute on Gc <= GPE_BT_MAXG compatible handles as ONE launch sequence (gridDim.z = Gc): the chain of small
// latency-bound kernels of one factorisation does not fill the chip, Gc of them in lock-step do.  The handles' mutexes
// are held by the caller.
// Device copies of the batch tables: a pool per device behind a mutex.  (They were thread_local once: every host
// thread that ever batched — par::loop spawns fresh ones per call on a multi-GPU node — leaked 41 KB of device memory.)
static std::mutex g_tab_mu;
static std::vector<BatchTab*> g_tab_pool[16];
static BatchTab* acquire_tab(int device)
{
    {
        std::lock_guard<std::mutex> lk(g_tab_mu);
        auto& pool = g_tab_pool[device];
        if (!pool.empty()) {
            BatchTab* t = pool.back();
            pool.pop_back();
            return t;
        }
    }
    BatchTab* t = nullptr;
    return hipMalloc(&t, sizeof(BatchTab)) == hipSuccess ? t : nullptr;
}
static void release_tab(int device, BatchTab* t)
{
    if (!t)
        return;
    std::lock_guard<std::mutex> lk(g_tab_mu);
    g_tab_pool[device].push_back(t);
}

// what a batched evaluation is to produce besides compute(): K^-1 and d log-lik / d theta of every member
// (kernel_lf_opt.hpp:77-92 for G restarts at once)
struct BatchWant {
    bool grad = false;
    int n_grad = 0, optimize_noise = 0;
    double* grad_out = nullptr; // host, Gc x n_grad
};

static int batch_enqueue_fused(gpe_ctx** cs, int Gc, BatchTab** tab_out, const BatchWant* want = nullptr)
{
    gpe_ctx* c0 = cs[0];
    DevGuard g(c0);
    *tab_out = nullptr;
    if (want && want->grad) { // every member needs the same three buffers before the table is built
        for (int q = 0; q < Gc; ++q) {
            gpe_ctx* c = cs[q];
            const size_t mat = sizeof(double) * (size_t)(c->ld * c->cap);
            if (!c->dLinv) {
                HIPCHK(c, hipMalloc(&c->dLinv, mat));
                c->inv_pad_n = -1; // (fresh memory: the pads of the recursive K^-1 are to be zero-filled)
            }
            if (!c->dKinv)
                HIPCHK(c, hipMalloc(&c->dKinv, mat));
            const int bufs_needed = Gc >= 4 ? 1 : 1 + inv2_partials(); // (inv2_prepare's rule: a batch of >= 4 cuts no k range)
            if (c->dInvS && c->invS_bufs < bufs_needed) {
                HIPCHK(c, hipStreamSynchronize(c->stream));
                hipFree(c->dInvS);
                c->dInvS = nullptr;
            }
            if (!c->dInvS && inv2_supported(c->N)) { // the recursive K^-1's scratch (inv2.hip)
                HIPCHK(c, hipMalloc(&c->dInvS, mat * (size_t)bufs_needed));
                c->invS_bufs = bufs_needed;
                c->inv_pad_n = -1; // (fresh memory: the pads of the recursive K^-1 are to be zero-filled)
            }
            if (inv2_supported(c->N))
                inv2_zero_pads(c, c0->stream); // (every member's, on the stream the batch runs on)
            int rc = ensure_grad_partial(c, want->n_grad);
            if (rc)
                return rc;
        }
    }
    if (c0->device >= 16)
        return GPE_ERR_UNSUPPORTED;
    bool batch_has_tail = false, batch_has_tall = false; // which data-flow launches this batch's plan issues
    { // the data-flow launches' hand-over buffers: every member's with the capacities and the armed parity of member 0's
        g_batch.G = Gc; // (the plan depends on the launch being batched, not on the table)
        g_batch.bt = reinterpret_cast<const BatchTab*>(1);
        const TailPlan pl = tail_plan(c0, c0->N, c0->N + c0->P);
        g_batch = BatchLaunch{};
        batch_has_tail = pl.t0 >= 0;
        batch_has_tall = pl.t0 >= 0 && pl.e0 >= 0;
        if (pl.t0 >= 0) {
            if (!prepare_tail(c0, pl, c0->stream))
                return GPE_ERR_NOMEM;
            for (int q = 1; q < Gc; ++q)
                if (!prepare_tail(cs[q], pl, c0->stream, c0))
                    return GPE_ERR_NOMEM;
        }
    }
    BatchTab* dtab = acquire_tab(c0->device); // held until the batch has finished (batch_finish_fused's caller releases it)
    if (!dtab)
        return GPE_ERR_NOMEM;
    *tab_out = dtab;
    std::vector<BatchTab> tabv(1);
    BatchTab& t = tabv[0];
    memset(&t, 0, sizeof(t));
    t.G = Gc;
    t.ncls = 8;
    for (int q = 0; q < Gc; ++q) {
        gpe_ctx* c = cs[q];
        digest_kernel(c);
        c->hInfo[0] = c->hInfo[1] = 0;
        const char* b[GPE_BT_CLS] = {(const char*)c->dA, (const char*)c->dXt, (const char*)c->dOm, (const char*)c->dAl,
                                     (const char*)c->dXinv, (const char*)c->dHead, (const char*)c->hInfo, (const char*)c->hScal,
                                     (const char*)c->dLinv, (const char*)c->dKinv, (const char*)c->dGradPartial, (const char*)c->dTail,
                                     (const char*)c->dInvS};
        for (int k = 0; k < GPE_BT_CLS; ++k)
            t.base[k][q] = b[k];
        t.kp[q] = c->kp;
        hipStreamSynchronize(c->stream); // nothing of this handle may still be in flight on its own stream
    }
    const size_t dbl = sizeof(double);
    const unsigned long long sz[GPE_BT_CLS] = {(unsigned long long)(dbl * c0->ld * c0->cap), (unsigned long long)(dbl * c0->ld * xt_rows(c0->D)),
                                               (unsigned long long)(dbl * c0->ld * c0->P), (unsigned long long)(dbl * c0->ld * c0->P),
                                               (unsigned long long)(dbl * (c0->cap / NB) * NB * NB), (unsigned long long)(dbl * GPE_HEAD_TILES * NB * NB), 64, 8192,
                                               (unsigned long long)(c0->dLinv ? dbl * c0->ld * c0->cap : 0), (unsigned long long)(c0->dKinv ? dbl * c0->ld * c0->cap : 0),
                                               (unsigned long long)(c0->dGradPartial ? dbl * c0->grad_partial_cap : 0),
                                               (unsigned long long)(c0->dTail ? dbl * 2 * (c0->tail_cap + c0->tall_cap) : 0),
                                               (unsigned long long)(c0->dInvS ? dbl * c0->ld * c0->cap * (Gc >= 4 ? 1 : 1 + inv2_partials()) : 0)};
    for (int k = 0; k < GPE_BT_CLS; ++k) {
        t.base0[k] = t.base[k][0];
        t.size[k] = sz[k];
    }
    HIPCHK(c0, hipMemcpyAsync(dtab, &t, sizeof(BatchTab), hipMemcpyHostToDevice, c0->stream));
    HIPCHK(c0, hipStreamSynchronize(c0->stream)); // `t` is pageable: the copy must have left it before it goes out of scope
    const bool la = c0->lookahead;
    c0->lookahead = false; // the batch fills the chip: one stream, no look-ahead split
    g_batch.bt = dtab;
    g_batch.G = Gc;
    int e = compute_enqueue(c0);
    for (int q = 1; q < Gc; ++q) { // the members' hand-over buffers went through the same launches as member 0's — those
        if (batch_has_tail) {      // that were issued: a pair no launch of this batch touched keeps the member's own state
            cs[q]->tail_count = c0->tail_count;
            cs[q]->tail_lay = c0->tail_lay;
        }
        if (batch_has_tall) {
            cs[q]->tall_count = c0->tall_count;
            cs[q]->tall_lay = c0->tall_lay;
        }
    }
    if (e == GPE_OK && want && want->grad) {
        // K^-1 (gp.hpp:254-264) and the gradient pair sum (gp.hpp:285-311) of every member, same launch sequence
        e = grad_enqueue(c0, want->n_grad, want->optimize_noise);
        if (e == GPE_OK && want->grad_out) {
            const int64_t off = c0->dGrad - c0->dGradPartial;
            for (int q = 0; q < Gc && e == GPE_OK; ++q)
                if (hipMemcpyAsync(want->grad_out + (size_t)q * want->n_grad, cs[q]->dGradPartial + off, sizeof(double) * want->n_grad,
                                   hipMemcpyDeviceToHost, c0->stream) != hipSuccess)
                    e = GPE_ERR_HIP;
        }
    }
    g_batch = BatchLaunch{};
    c0->lookahead = la;
    return e;
}

static int batch_finish_fused(gpe_ctx** cs, int Gc, int* rc, const BatchWant* want = nullptr)
{
    gpe_ctx* c0 = cs[0];
    DevGuard g(c0);
    HIPCHK(c0, wait_stream(c0->stream));
    HIPCHK(c0, hipGetLastError());
    const int64_t nblk = (c0->N + NB - 1) / NB;
    for (int q = 0; q < Gc; ++q) {
        gpe_ctx* c = cs[q];
        c->have_L = true;
        c->inv_ok = false;
        c->al_prefilled = false;
        c->ll_partials = c0->flow_solve && nblk <= 256 ? (int)nblk : 0;
        // the usual finish on the handle's own (idle) stream: sums the per-block partials; a sweep that gave up
        // (never expected) is re-run block by block for that GP alone
        const int64_t retries = c->flow_retries;
        rc[q] = compute_finish(c);
        if (want && want->grad) {
            c->inv_ok = true; // gp.hpp:263
            if (c->flow_retries != retries && rc[q] >= 0 && want->grad_out) {
                // (never expected) this member's sweep or factorisation was re-run on its own after the batch: its
                // K^-1 / gradient came from the first attempt — once more, alone
                c->inv_ok = false;
                int e = grad_fetch(c, want->grad_out + (size_t)q * want->n_grad, want->n_grad, want->optimize_noise, false);
                if (e < 0)
                    rc[q] = e;
            }
        }
    }
    return GPE_OK;
}

static int batch_compute_impl(gpe_handle* hs, int G, int* status, const BatchWant* want)
{
    for (int g_ = 0; hs && g_ < G; ++g_)
        if (hs[g_])
            ++hs[g_]->epoch;
    if (!hs || G < 0)
        return GPE_ERR_ARG;
    std::vector<int> rc(G, 0);
    static const bool fused_ok = !(getenv("GPE_BATCH") && atoi(getenv("GPE_BATCH")) == 0);
    bool fused = fused_ok && G >= 2;
    for (int g = 0; g < G && fused; ++g) {
        gpe_ctx* c = hs[g];
        fused = c && c->N > 0 && batch_compatible(hs[0], c) && lam_columns(c->kind, c->n_theta, c->D) == 0
            && c->flow_solve && (c->N + NB - 1) / NB <= 256
            && (!(want && want->grad) || (c->nbo % 128 == 0 && c->nbo <= 256)); // (K^-1: the one-launch panel inverses)
        for (int q = 0; q < g && fused; ++q)
            fused = hs[q] != c; // the same handle twice cannot be stepped in parallel
    }
    if (fused) {
        // every handle's mutex, taken in one canonical order (by address) whatever order the caller listed them in — two
        // threads batching overlapping sets cannot deadlock — and released by RAII on every way out
        std::vector<gpe_ctx*> order(hs, hs + G);
        std::sort(order.begin(), order.end());
        std::vector<std::unique_lock<std::mutex>> locks;
        locks.reserve(G);
        for (gpe_ctx* c : order) {
            locks.emplace_back(c->mu);
            DevGuard dg(c);
        }
        int worst = GPE_OK;
        // Sub-batches of <= GPE_BT_MAXG GPs, up to four in flight on their own streams: while one sub-batch is in its
        // panel steps (latency-bound workgroups, one per CU) another one's matrix-core updates fill the chip.
        static const int nsub_env = getenv("GPE_BATCH_SPLIT") ? atoi(getenv("GPE_BATCH_SPLIT")) : 2;
        int nsub = std::max(1, std::min(4, nsub_env));
        if (G < 16)
            nsub = 1;
        const int per = std::min(GPE_BT_MAXG, (G + nsub - 1) / nsub);
        for (int g0 = 0; g0 < G;) {
            // one wave of sub-batches
            int starts[4], counts[4], nw = 0;
            for (; nw < nsub && g0 < G; ++nw) {
                starts[nw] = g0;
                counts[nw] = std::min(per, G - g0);
                g0 += counts[nw];
            }
            int en[4];
            BatchTab* tabs[4] = {nullptr, nullptr, nullptr, nullptr};
            BatchWant wsub[4];
            for (int w = 0; w < nw; ++w) {
                if (want) {
                    wsub[w] = *want;
                    if (want->grad_out)
                        wsub[w].grad_out = want->grad_out + (size_t)starts[w] * want->n_grad;
                }
                if (counts[w] >= 2)
                    en[w] = batch_enqueue_fused(hs + starts[w], counts[w], &tabs[w], want ? &wsub[w] : nullptr);
                else {
                    DevGuard dg(hs[starts[w]]);
                    en[w] = compute_enqueue(hs[starts[w]]);
                }
            }
            for (int w = 0; w < nw; ++w) {
                if (en[w] != GPE_OK) {
                    for (int q = 0; q < counts[w]; ++q)
                        rc[starts[w] + q] = en[w];
                    worst = en[w];
                    if (tabs[w]) { // whatever was enqueued before the failure may still read the table
                        DevGuard dg(hs[starts[w]]);
                        hipStreamSynchronize(hs[starts[w]]->stream);
                        release_tab(hs[starts[w]]->device, tabs[w]);
                    }
                    continue;
                }
                if (counts[w] >= 2) {
                    int e = batch_finish_fused(hs + starts[w], counts[w], rc.data() + starts[w], want ? &wsub[w] : nullptr);
                    if (e < 0) {
                        worst = e;
                        DevGuard dg(hs[starts[w]]);
                        hipStreamSynchronize(hs[starts[w]]->stream);
                    }
                    release_tab(hs[starts[w]]->device, tabs[w]); // the stream is idle: nothing reads the table any more
                }
                else {
                    gpe_ctx* c1 = hs[starts[w]];
                    DevGuard dg(c1);
                    rc[starts[w]] = compute_finish(c1);
                    if (want && want->grad && rc[starts[w]] >= 0 && want->grad_out) { // a sub-batch of one: on its own
                        int e = grad_fetch(c1, wsub[w].grad_out, want->n_grad, want->optimize_noise, false);
                        if (e < 0)
                            rc[starts[w]] = e;
                    }
                }
            }
        }
        for (int g = 0; g < G; ++g) {
            if (status)
                status[g] = rc[g];
            if (rc[g] < 0)
                worst = rc[g];
        }
        return worst;
    }
    // enqueue everything first (each GP on its own stream), then collect: kernels of different
    // GPs overlap on the device — the TBB par::loop of multi_gp.hpp:124-126, on one GPU.
    std::vector<gpe_ctx*> order;
    for (int g = 0; g < G; ++g)
        if (hs[g])
            order.push_back(hs[g]);
    std::sort(order.begin(), order.end());
    order.erase(std::unique(order.begin(), order.end()), order.end()); // a handle listed twice is locked once
    std::vector<std::unique_lock<std::mutex>> locks;
    locks.reserve(order.size());
    for (gpe_ctx* c : order) {
        locks.emplace_back(c->mu);
        DevGuard dg(c);
    }
    std::vector<char> first(G, 0); // first occurrence of a handle: the one that is enqueued (a second one would race it)
    for (int g = 0; g < G; ++g) {
        gpe_ctx* c = hs[g];
        if (!c) {
            rc[g] = GPE_ERR_ARG;
            continue;
        }
        first[g] = 1;
        for (int q = 0; q < g; ++q)
            if (hs[q] == c)
                first[g] = 0;
        if (!first[g])
            continue;
        hipSetDevice(c->device);
        rc[g] = compute_enqueue(c);
    }
    int worst = GPE_OK;
    for (int g = 0; g < G; ++g) {
        gpe_ctx* c = hs[g];
        if (!c)
            continue;
        if (first[g]) {
            hipSetDevice(c->device);
            if (rc[g] == GPE_OK)
                rc[g] = compute_finish(c);
            if (want && want->grad && rc[g] >= 0 && want->grad_out) {
                int e = grad_fetch(c, want->grad_out + (size_t)g * want->n_grad, want->n_grad, want->optimize_noise, false);
                if (e < 0)
                    rc[g] = e;
            }
        }
        else
            for (int q = 0; q < g; ++q)
                if (hs[q] == c) {
                    rc[g] = rc[q];
                    break;
                }
        if (status)
            status[g] = rc[g];
        if (rc[g] < 0)
            worst = rc[g];
    }
    return worst;
}

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
