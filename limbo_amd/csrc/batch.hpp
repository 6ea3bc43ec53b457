// batch.hpp — G independent GPs stepped by ONE launch sequence (gridDim.z = GP): the pointer table, the fused enqueue / finish, sub-batches on their own streams.
// A part of engine.hip's translation unit (included there, once, at the place its contents used to stand: they share the
// file-local types and helpers of the engine — gpe_ctx, PhaseScope, DevGuard ...); split out in round 6 for readability.
#pragma once

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
