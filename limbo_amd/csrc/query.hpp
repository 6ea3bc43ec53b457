// query.hpp — the batched query (gp.hpp:613-632 for M points): transposed layout for device kernels, the N x M layout for caller-supplied cross kernels, the one-launch paths for a handful of points.
// A part of engine.hip's translation unit (included there, once, at the place its contents used to stand: they share the
// file-local types and helpers of the engine — gpe_ctx, PhaseScope, DevGuard ...); split out in round 6 for readability.
#pragma once

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
