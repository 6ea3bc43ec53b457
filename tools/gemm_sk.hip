// gemm_sk.hip — the trailing update  C[m x n] -= A[m x k] B[n x k]^T  (lower triangle) as ONE persistent launch whose
// workgroups each do the same number of matrix-core k-iterations ("stream-k"), for gfx950.
//
// Why: the tile-per-workgroup kernels of gemm.hip quantise.  A 128 x 128 x 256 tile takes ~40 us on a CU; the five
// largest updates of an N = 4096 factorisation have 253 .. 465 live tiles for 256 CUs, so every one of them costs "two
// tiles" (80-90 us) whatever its size, and the small ones cannot fill the chip at all (profiles/r02_pmc_bench_n4096.txt:
// 0.375 of the fp64 matrix-core peak over the 15 launches).  Here the unit of work is one k-iteration (BKT deep) of one
// tile:  G persistent workgroups (1 or 2 per CU), the live tiles dealt to the 8 XCDs in contiguous bands; inside a band
// every workgroup first takes floor(Tb / Gb) whole tiles ("data-parallel" part, no overhead), and the remaining
// Tb mod Gb tiles' k-iterations are cut into Gb equal spans.  A span covers at most two tiles: a workgroup that holds
// the LAST k-segment of a tile finishes it (adds the partial sums of the earlier segments, then the usual
// read-modify-write of C); one that holds an earlier segment writes its accumulators to a private slot of a workspace
// and raises a flag.
//
// Order inside a workgroup: contributed segment FIRST, whole tiles, finished segment LAST — so a finisher's
// contributors (the workgroups just before it in the same band: lower blockIdx.x, same XCD) published their partials a
// whole span earlier and nobody waits in practice; and whatever a workgroup does wait for belongs to a workgroup that
// was dispatched before it (the rule of dev.h: progress does not depend on residency).  Partial sums are added in a
// fixed order (own segment, then contributors in ascending order): results are bitwise reproducible from run to run.
// The exchange carries no fence: slots and flags are written with device-scope (write-through) stores and read with
// device-scope loads, as the head-tile hand-over of potrf.hip.  The poll is bounded; a wave that gives up raises *err and
// the host runs the evaluation again with the tile-per-workgroup kernels (engine.hip).
//
// The right-hand-side rows appended under the matrix (engine.hip: z = L^-1 obs_mean rides in the factorisation) used to
// cost a whole extra row of tiles per launch (30 of 495 workgroups in the first update, each with ONE live row).  Here
// they are not tiles: every workgroup updates its n / G columns of those rows with plain FMAs while its first operand
// stage is in flight.
//
// The inner loop is gemm.hip's k_gemm_glds: 128 x 128 tile, 8 waves (2 x 4), operand k-rows HBM/L2 -> LDS with
// global_load_lds_dwordx4, counted vmcnt, raw s_barrier, v_mfma_f64_4x4x4_4b, C traffic in the lane = row layout.
#include "dev.h"
#include "gemm_glds64.h"
#include <hip/hip_ext.h>
#include <atomic>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

#ifdef SK_TIMING
// in-kernel stamps (tools/updbench_t): 100 MHz wall clock, 16 per workgroup
__device__ long long g_sk_ts[512 * 16];
#define SKTS(i) do { if (threadIdx.x == 0) g_sk_ts[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define SKTS(i) do { } while (0)
#endif

namespace {

constexpr int SK_TM = 128, SK_TN = 128;

struct SkParams {
    GemmArgs g;         // C, A, B, m = main rows (without the right-hand-side rows), n, k, tri, grow0, gcol0
    int rhs_rows;       // rows m .. m + rhs_rows - 1 of C / A: right-hand-side rows
    int T;              // live tiles
    int KI;             // k / BKT
    int G;              // workgroups = gridDim.x
    int bands;          // 8 (G a multiple of 8) or 1
    int smin;           // shortest span worth a workgroup, in k-iterations (2 smin <= KI)
    int wA, wB;         // two workgroups per CU: the first-dispatched half of a band (the older waves: they win the matrix-core
                        // issue arbitration while both run, 3.0 against 4.2 us per k-iteration) takes spans wA / wB times longer
                        // than the second half, so that both reach their last k-iteration together (1, 1: equal spans)
    double* ws;         // G slots of 128 x 128 doubles
    gpe_epoch_t* flags; // G words, never reset: a slot is valid when its word holds this launch's epoch
    gpe_epoch_t epoch;      // what a contributor writes into its flag word
    gpe_epoch_t wait_epoch; // what a finisher waits for (= epoch; the fault-injection hook makes it a value nobody writes)
    int spin_limit;
    int* err;
};

// live tile t (dense enumeration, column by column, row tile fastest) -> (ti, tj)
static __device__ __forceinline__ void sk_tile_coords(const GemmArgs& g, int tiles_m, int t, int& ti, int& tj)
{
    if (!g.tri) {
        ti = t % tiles_m;
        tj = t / tiles_m;
        return;
    }
    int c = 0;
    if (g.grow0 == g.gcol0) { // first_live(tj) = tj: S(tj) = tj tiles_m - tj (tj - 1) / 2 tiles before column tj
        // the last column with S(c) <= t by bisection — integers only: everything here is wave-uniform and stays in SGPRs
        // (a closed form through sqrt() put doubles into VGPRs for the whole kernel)
        int lo = 0, hi = tiles_m < (int)((g.n + SK_TN - 1) / SK_TN) ? tiles_m : (int)((g.n + SK_TN - 1) / SK_TN);
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (mid * tiles_m - mid * (mid - 1) / 2 <= t)
                lo = mid;
            else
                hi = mid;
        }
        c = lo;
        t -= c * tiles_m - c * (c - 1) / 2;
    }
    const int tiles_n = (int)((g.n + SK_TN - 1) / SK_TN);
    for (; c < tiles_n; ++c) {
        const int f = first_live_tile<SK_TM, SK_TN>(g, c);
        const int cnt = tiles_m - f;
        if (t < cnt) {
            ti = f + t;
            tj = c;
            return;
        }
        t -= cnt;
    }
    ti = tiles_m - 1; // not reached: t < T by construction (the loop is bounded all the same)
    tj = tiles_n - 1;
}

template <int BKT, int NST, int EPC, int MINB>
__global__ __launch_bounds__(512, MINB) void k_gemm_sk(SkParams P)
{
    constexpr int TM = SK_TM, TN = SK_TN, WM = 2, WN = 4, NWV = 8;
    constexpr int SA = TM + 16, SB = TN + 16; // k-row strides (doubles), == 16 mod 32
    constexpr int STAGE = BKT * (SA + SB);
    constexpr int RA = TM / WM / 16, RB = TN / WN / 4;
    constexpr int LPW = 2 * BKT / NWV; // glds instructions per wave per stage
    __shared__ __attribute__((aligned(16))) double lds[NST * STAGE];
    const GemmArgs& g = P.g;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave % WM) * (TM / WM), wn = (wave / WM) * (TN / WN);
    const int arow = wm + (lane & 15), bcol = wn + (lane & 3), kq = lane >> 4;
    const int tiles_m = (int)((g.m + TM - 1) / TM);
    SKTS(0);
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15;

    // ---- this workgroup's share -------------------------------------------------------------------------------
    const int b = blockIdx.x;
    int band = 0, idx = b, Gb = P.G;
    if (P.bands == 8) {
        band = b & 7;
        idx = b >> 3;
        Gb = P.G >> 3;
    }
    const int tb0 = (int)((int64_t)band * P.T / P.bands), tb1 = (int)((int64_t)(band + 1) * P.T / P.bands);
    const int Tb = tb1 - tb0;
    const int whole = Tb / Gb, rem = Tb - whole * Gb;
    const int KI = P.KI;
    const int64_t Isk = (int64_t)rem * KI;
    int Gsk = Gb;
    if (Isk < (int64_t)Gb * P.smin) {
        Gsk = (int)(Isk / P.smin);
        Gsk = Gsk < 1 ? 1 : Gsk;
    }
    // spans: equal, or weighted wA : wB between the two halves of the band (only with every workgroup taking part, and only
    // while the longer span still fits within one tile's KI iterations: a span never covers more than two tiles)
    const int half = Gb >> 1;
    const bool weighted = P.wA != P.wB && Gsk == Gb && half > 0 && (int64_t)rem * 2 * P.wA <= (int64_t)Gb * (P.wA + P.wB);
    const int64_t wtot = (int64_t)half * P.wA + (int64_t)(Gb - half) * P.wB;
    auto span_start = [&](int i) -> int64_t {
        if (i >= Gsk)
            return Isk;
        if (!weighted)
            return (int64_t)i * Isk / Gsk;
        const int64_t cum = i <= half ? (int64_t)i * P.wA : (int64_t)half * P.wA + (int64_t)(i - half) * P.wB;
        return cum * Isk / wtot;
    };
    const int64_t s0 = span_start(idx), s1 = span_start(idx + 1);
    const int sk_tile0 = tb0 + whole * Gb;

    // the pieces of the span: at most two (2 smin <= KI and rem < Gb keep a span within KI iterations)
    int pa_t = -1, pa_k0 = 0, pa_k1 = 0, pb_t = -1, pb_k1 = 0;
    if (s1 > s0) {
        const int ta = (int)(s0 / KI);
        pa_t = ta;
        pa_k0 = (int)(s0 - (int64_t)ta * KI);
        const int64_t e = (int64_t)(ta + 1) * KI;
        pa_k1 = (int)((s1 < e ? s1 : e) - (int64_t)ta * KI);
        if (s1 > e) {
            pb_t = ta + 1;
            pb_k1 = (int)(s1 - e);
        }
    }

    bool rhs_pending = P.rhs_rows > 0; // done by the first piece, under the latency of its first operand stage
    // ---- one piece: k-iterations [k0, k1) of tile t -------------------------------------------------------------
    // contributor (k1 < KI): accumulators -> this workgroup's slot.  finisher (k1 == KI): + the slots of the workgroups
    // that hold [0, k0) (if k0 > 0), then C -= acc.
    auto run_piece = [&](auto mode_tag, int t, int k0, int k1, int64_t tile_first_iter) {
        constexpr int MODE = decltype(mode_tag)::value; // 0: contributed segment, 1: whole tile, 2: finished segment
        int ti, tj;
        sk_tile_coords(g, tiles_m, t, ti, tj);
        const int64_t row0 = (int64_t)ti * TM, col0 = (int64_t)tj * TN;
        const int64_t mrows = g.m - row0, ncols = g.n - col0;
        const int mr = (int)(mrows < TM ? mrows : TM), nc = (int)(ncols < TN ? ncols : TN);
        int ra = 2 * lane, rb = 2 * lane;
        {
            const int ma = (mr - 1) & ~1, mb = (nc - 1) & ~1;
            ra = ra < ma ? ra : ma;
            rb = rb < mb ? rb : mb;
        }
        const int64_t kbeg = (int64_t)k0 * BKT;
        const double* pa = g.A + row0 + ra + ((int64_t)wave + kbeg) * g.lda;
        const double* pb = g.B + col0 + rb + ((int64_t)wave + kbeg) * g.ldb;
        const int64_t astep = (int64_t)NWV * g.lda, bstep = (int64_t)NWV * g.ldb;
        auto issue = [&](int stage) {
            double* sa = lds + stage * STAGE + wave * SA;
            double* sb = lds + stage * STAGE + BKT * SA + wave * SB;
#pragma unroll
            for (int q = 0; q < BKT / NWV; ++q) {
                __builtin_amdgcn_global_load_lds(pa, (lds_void_t*)(sa + q * NWV * SA), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(pb, (lds_void_t*)(sb + q * NWV * SB), 16, 0, 0);
                pa += astep;
                pb += bstep;
            }
        };
        double acc[RA][RB];
#pragma unroll
        for (int a = 0; a < RA; ++a)
#pragma unroll
            for (int bb = 0; bb < RB; ++bb)
                acc[a][bb] = 0.0;

        const int nk = k1 - k0;
        SKTS(1 + 4 * MODE);
        issue(0);
        if (rhs_pending) { // (workgroup-uniform) in the LDS of stage NST - 1, which is requested only after this
            gemm_rhs_rows<512>(g, P.rhs_rows, P.G, lds + (NST - 1) * STAGE);
            rhs_pending = false;
        }
        for (int tt = 0; tt < nk; ++tt) {
            const int st = tt % NST;
            if (tt + 1 < nk) {
                issue((tt + 1) % NST);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
            }
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // every wave's pieces of stage st have landed
            const double* As = lds + st * STAGE;
            const double* Bs = As + BKT * SA;
#pragma unroll
            for (int ks = 0; ks < BKT; ks += 4) {
                double af[RA], bf[RB];
#pragma unroll
                for (int x = 0; x < RA; ++x)
                    af[x] = As[(ks + kq) * SA + arow + 16 * x];
#pragma unroll
                for (int x = 0; x < RB; ++x)
                    bf[x] = Bs[(ks + kq) * SB + bcol + 4 * x];
#pragma unroll
                for (int n = 0; n < RB; ++n)
#pragma unroll
                    for (int m = 0; m < RA; ++m)
                        acc[m][n] = mfma4(af[m], bf[n], acc[m][n]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // stage st may be refilled
        }
        SKTS(2 + 4 * MODE);

        if constexpr (MODE == 0) {
            // contributor: the accumulators, as they are (fragment layout: a slot is private to the workgroups that
            // exchange it), to this workgroup's slot — PLAIN stores: they are acknowledged by this XCD's L2, which is also
            // the finisher's (same band = same blockIdx.x mod 8 = same XCD: the dispatcher deals workgroups round-robin over
            // the XCDs).  Write-through stores + device-scope loads (the form of potrf.hip's hand-over, right for 32 KB
            // tiles between arbitrary XCDs) cost ~10 us per launch here: 128 KB per slot to the memory side and back.
            // The flag word carries the writer's XCC id; a finisher on another XCD (never seen) reports it and the host
            // re-runs the evaluation with the tile-per-workgroup kernels.
            double* slot = P.ws + (int64_t)b * (TM * TN);
#pragma unroll
            for (int m = 0; m < RA; ++m)
#pragma unroll
                for (int n = 0; n < RB; ++n)
                    slot[(m * RB + n) * 512 + tid] = acc[m][n];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's part is in L2
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(P.flags + b, (P.epoch << 4) | (gpe_epoch_t)xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            SKTS(3);
            return;
        }
        if (MODE == 2 && k0 > 0) {
            // finisher of a split tile: the workgroups before this one (same band) whose spans reach back to the tile's
            // first iteration, in ascending order
            int first = idx - 1;
            while (first > 0 && span_start(first) > tile_first_iter)
                --first;
            for (int i = first; i < idx; ++i) {
                const int bb = P.bands == 8 ? (i << 3) + band : i;
                int spins = 0;
                gpe_epoch_t w;
                while (((w = __hip_atomic_load(P.flags + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 4) != P.wait_epoch) {
                    if (++spins > P.spin_limit) {
                        if (lane == 0)
                            *P.err = 1;
                        w = (gpe_epoch_t)xcc;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if ((int)(w & 15) != xcc && lane == 0) // the slot sits in another XCD's L2: not visible from here
                    *P.err = 2;
                asm volatile("" ::: "memory"); // the slot's loads stay behind the poll
                const double* slot = P.ws + (int64_t)bb * (TM * TN);
                if constexpr (EPC > 0) { // (the two-per-CU shape has 128 VGPRs: a quarter of the slot at a time)
#pragma unroll
                    for (int m = 0; m < RA; ++m) {
                        double pv[RB];
#pragma unroll
                        for (int n = 0; n < RB; ++n)
                            pv[n] = slot[(m * RB + n) * 512 + tid];
#pragma unroll
                        for (int n = 0; n < RB; ++n)
                            acc[m][n] += pv[n];
                    }
                }
                else {
                    double pv[RA][RB];
#pragma unroll
                    for (int m = 0; m < RA; ++m)
#pragma unroll
                        for (int n = 0; n < RB; ++n)
                            pv[m][n] = slot[(m * RB + n) * 512 + tid];
#pragma unroll
                    for (int m = 0; m < RA; ++m)
#pragma unroll
                        for (int n = 0; n < RB; ++n)
                            acc[m][n] += pv[m][n];
                }
                asm volatile("" ::: "memory");
            }
        }
        SKTS(3 + 4 * MODE);
        {
            using WT = WaveTileC<RA, RB>;
            constexpr int SCR = WT::SCRATCH / (EPC > 0 ? EPC : 1);
            static_assert(NWV * SCR <= NST * STAGE, "transposition scratch must fit in the operand stages");
            double* Cw = g.C + (col0 + wn) * g.ldc + row0 + wm;
            const int rlim = mr - wm < WT::R ? mr - wm : WT::R, clim = nc - wn < WT::CN ? nc - wn : WT::CN;
            if (rlim > 0 && clim > 0) {
                if constexpr (EPC > 0)
                    WT::template rmw_chunked<EPC>(acc, lds + wave * SCR, Cw, g.ldc, rlim, clim, g.overwrite, lane);
                else {
                    double cv[WT::NIT];
                    if (g.overwrite != 1)
                        WT::load(cv, Cw, g.ldc, rlim, clim, lane);
                    WT::store(acc, cv, lds + wave * WT::SCRATCH, Cw, g.ldc, rlim, clim, g.overwrite, lane);
                }
            }
        }
        __syncthreads(); // the next piece's prologue overwrites the LDS stages
        SKTS(4 + 4 * MODE);
    };

    // contributed segment first, whole tiles, finished segment(s) last.  Three instantiations of the piece body (as ONE
    // loop over a step list the compiler needed 256 VGPRs and scratch).
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    const bool a_contrib = pa_t >= 0 && pa_k1 < KI, b_contrib = pb_t >= 0 && pb_k1 < KI;
    if (a_contrib || b_contrib) // (a contributed first piece has no second piece)
        run_piece(M0{}, sk_tile0 + (a_contrib ? pa_t : pb_t), a_contrib ? pa_k0 : 0, a_contrib ? pa_k1 : pb_k1, 0);
    for (int j = 0; j < whole; ++j)
        run_piece(M1{}, tb0 + idx * whole + j, 0, KI, 0);
    // finals: the second piece (a whole tile from its first iteration) before the first (which may wait)
    const bool a_final = pa_t >= 0 && !a_contrib, b_final = pb_t >= 0 && !b_contrib;
    for (int f = (b_final ? 0 : 1); f < (a_final ? 2 : 1); ++f) {
        const bool tb = f == 0;
        run_piece(M2{}, sk_tile0 + (tb ? pb_t : pa_t), tb ? 0 : pa_k0, tb ? pb_k1 : pa_k1, (int64_t)(tb ? pb_t : pa_t) * KI);
    }
    if (rhs_pending) // a workgroup without any piece still owns its columns of the right-hand-side rows
        gemm_rhs_rows<512>(g, P.rhs_rows, P.G, lds);
    SKTS(13);
}

int env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

} // namespace

// number of live 128 x 128 tiles of the update (rows = main rows only)
static int sk_live_tiles(const GemmArgs& g)
{
    const int tm = (int)((g.m + SK_TM - 1) / SK_TM), tn = (int)((g.n + SK_TN - 1) / SK_TN);
    if (!g.tri)
        return tm * tn;
    int cnt = 0;
    for (int tj = 0; tj < tn; ++tj)
        cnt += tm - first_live_tile<SK_TM, SK_TN>(g, tj);
    return cnt;
}

// Is this update one for the persistent kernel?  (both operands row-contiguous, k a multiple of 32, no batch)
bool gemm_sk_ok(const GemmArgs& g)
{
    static const int on = env_int("GPE_SK", 1);
    return on && !g_batch.bt && !g.a_kmajor && !g.b_kmajor && !g.ktri && !g.overwrite && g.k >= 64 && g.k % 32 == 0
        && (g.lda % 2) == 0 && (g.ldb % 2) == 0 && ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0 && g.m > 0 && g.n > 0;
}

size_t gemm_sk_workspace_bytes() { return (size_t)512 * SK_TM * SK_TN * sizeof(double) + 512 * sizeof(gpe_epoch_t); }

// g.m counts ALL rows; the last rhs_rows of them are right-hand-side rows.  ws: gemm_sk_workspace_bytes() of device
// memory, zeroed once at allocation (slots first, then the flag words); err: raised by a wave whose poll gave up.
// variant 1: BKT 32, one workgroup per CU (G = 256); 2: BKT 16, two per CU (G = 512).  G, smin: 0 = default.
void launch_gemm_sk_ex(hipStream_t s, const GemmArgs& g0, int rhs_rows, void* ws, int* err, int variant, int G_req, int smin_req,
                       int wA, int wB)
{
    static std::atomic<gpe_epoch_t> g_epoch{0};
    SkParams P{};
    P.g = g0;
    P.g.m = g0.m - rhs_rows;
    P.rhs_rows = rhs_rows;
    P.T = sk_live_tiles(P.g);
    const int bkt = variant == 1 ? 32 : 16;
    P.KI = (int)(g0.k / bkt);
    int G = G_req > 0 ? G_req : (variant == 1 ? 256 : 512);
    G = G > 512 ? 512 : G;
    if (g0.grid_limit > 0 && G > g0.grid_limit)
        G = g0.grid_limit;
    const int smin = smin_req > 0 ? smin_req : (variant == 1 ? 2 : 4);
    P.smin = 2 * smin <= P.KI ? smin : (P.KI / 2 > 0 ? P.KI / 2 : 1);
    if (G >= 8)
        G -= G % 8;
    P.G = G < 1 ? 1 : G;
    P.bands = P.G >= 8 ? 8 : 1;
    P.wA = P.wB = 1;
    if (variant != 1 && P.G == 512 && wA > 0 && wB > 0) { // two per CU: blockIdx.x < 256 are the older workgroups
        P.wA = wA;
        P.wB = wB;
    }
    P.ws = (double*)ws;
    P.flags = (gpe_epoch_t*)((char*)ws + (size_t)512 * SK_TM * SK_TN * sizeof(double));
    P.epoch = P.wait_epoch = ++g_epoch;
    static const bool fault = env_int("GPE_SK_FAULT", 0) != 0; // test hook: every finisher of a split tile gives up at once
    P.spin_limit = fault ? 0 : GPE_FLOW_SPIN_LIMIT;
    if (fault)
        P.wait_epoch = ~(gpe_epoch_t)0 >> 4;
    P.err = err;
    const dim3 grid((unsigned)P.G), block(512);
    auto go = [&](auto kern) {
        if (g0.stop_event)
            GPE_LAUNCH_STOP("k_gemm_sk", kern, grid, block, 0, s, (hipEvent_t)g0.stop_event, P);
        else
            GPE_LAUNCH_NAMED("k_gemm_sk", kern, grid, block, 0, s, P);
    };
    if (variant == 1)
        go(k_gemm_sk<32, 2, 0, 1>);
    else
        go(k_gemm_sk<16, 2, 4, 4>); // (HIP: the second launch bound is waves per SIMD — 4 = two workgroups of 8 waves per CU, <= 128 VGPRs)
}

void launch_gemm_sk(hipStream_t s, const GemmArgs& g0, int rhs_rows, void* ws, int* err)
{
    static const int variant = env_int("GPE_SK_VARIANT", 2);
    static const int smin_env = env_int("GPE_SK_SMIN", 0);
    static const int g_env = env_int("GPE_SK_G", 0);
    static const int wa = env_int("GPE_SK_WA", 4), wb = env_int("GPE_SK_WB", 3);
    launch_gemm_sk_ex(s, g0, rhs_rows, ws, err, variant, g_env, smin_env, wa, wb);
}

#ifdef SK_TIMING
void dump_sk_timing(int G, const char* what)
{
    static long long h[512 * 16];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sk_ts), sizeof(h));
    long long t0 = h[0];
    for (int b = 0; b < G; ++b)
        t0 = h[b * 16] && h[b * 16] < t0 ? h[b * 16] : t0;
    printf("%s: stamps in us from the first workgroup's start (0 = not taken): start | contributed piece: issue, k-loop done, slot+flag out | "
           "whole tiles (last): issue, k-loop, -, done | final piece (last): issue, k-loop, partials in, done | end\n", what);
    for (int b = 0; b < G; ++b) {
        if (!(b < 16 || b % 37 == 0 || b >= G - 8))
            continue;
        printf("  wg %3d:", b);
        for (int i : {0, 1, 2, 3, 5, 6, 8, 9, 10, 11, 12, 13})
            printf(" %7.2f", h[b * 16 + i] ? (h[b * 16 + i] - t0) * 0.01 : 0.0);
        printf("\n");
    }
}
void clear_sk_timing()
{
    static long long z[512 * 16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sk_ts), z, sizeof(z));
}
#endif
