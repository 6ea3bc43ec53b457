// inverse.hpp — K^-1 (the recursion on the factor and the panel form), the LOO weight matrix, the gradient objectives' enqueue.
// A part of engine.hip's translation unit (included there, once, at the place its contents used to stand: they share the
// file-local types and helpers of the engine — gpe_ctx, PhaseScope, DevGuard ...); split out in round 6 for readability.
#pragma once

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
