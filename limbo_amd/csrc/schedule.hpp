// schedule.hpp — the launch schedule of ONE evaluation: where the data-flow launches begin (tail_plan), the blocked factorisation (potrf_blocked), the sweeps, compute_enqueue / compute_finish and their re-run paths.
// A part of engine.hip's translation unit (included there, once, at the place its contents used to stand: they share the
// file-local types and helpers of the engine — gpe_ctx, PhaseScope, DevGuard ...); split out in round 6 for readability.
#pragma once

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
                // (A ragged order: the columns of its last block, N64 .. N, stay out of this launch — they would be a 23rd column of
                // 128 x 128 tiles at N = 4100, a second round of the chip, 214 -> 337 us — and take the tall launch's columns together
                // with the closing launch's in their own update below.  The rows under N64 — the ragged rows and the right-hand sides —
                // are the launch's "right-hand-side rows": plain FMAs in front of the tiles where that saves the round, gemm.hip.)
                g.m = M - t0;
                g.n = N64 - t0;
                g.k = t0 - e0;
                g.tri = 1;
                g.grow0 = t0;
                g.gcol0 = t0;
                g.rhs_rows = (int)(M - N64);
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
                const int64_t k0 = e0 >= 0 ? e0 : t0; // (behind a tall launch: its columns too, see the update above)
                g.C = A + N64 + N64 * ld;
                g.ldc = ld;
                g.A = A + N64 + k0 * ld;
                g.lda = ld;
                g.B = A + N64 + k0 * ld;
                g.ldb = ld;
                g.m = M - N64;
                g.n = N - N64;
                g.k = N64 - k0;
                g.tri = 1;
                g.grow0 = N64;
                g.gcol0 = N64;
                PhaseScope ps(c, GPE_PH_POTRF_UPDATE, gemm_flops(g));
                // ONE tile with k up to 2816: dealt to up to 32 workgroups + an ordered fold (potrf.hip); its scratch is the pair of polled
                // buffers the closing launch has just used — dead until the next launch arms all of them again
                double* const used = c->dTail + ((c->tail_count - 1) & 1) * c->tail_cap;
                // ... and factored, inverted and its right-hand-side rows solved by the same two launches where that form serves
                // (GPE_RAGGED_FINISH=0: the update alone, then the panel code below)
                static const bool finish = !(getenv("GPE_RAGGED_FINISH") && atoi(getenv("GPE_RAGGED_FINISH")) == 0);
                // (one workgroup adds the slots up here: worth it while the block is narrow — N = 520 0.156 -> 0.149 ms, 1100 0.269 ->
                // 0.258; from ~40 columns on the sixteen workgroups of k_ragged_fold are quicker than the launches they cost)
                if (finish && !c->prof && N - N64 <= 40
                    && launch_ragged_finish(s, g.C, ld, g.A, ld, N - N64, M - N, g.k, used, pl.need_tail, c->dXinv + (N64 / NB) * (NB * NB),
                                            c->dInfo, N64))
                    break;
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
