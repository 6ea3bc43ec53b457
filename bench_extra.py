#!/usr/bin/env python
"""Secondary measurements of SURVEY.md §8(d) on one MI355X (not the driver's bench.py contract):
HP-objective evaluations/s with gradient (config 2), batched queries/s (config 3), add_sample/s
(config 5), G independent GPs (config 4 on one GPU).  Prints one JSON object."""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    import torch
    assert torch.cuda.is_available()
    from limbo_amd import _capi
    from limbo_amd import synth as O  # problem generator (pure numpy)
    eng = _capi.load_engine()
    out = {}

    # config 2: HP objective with gradient at N=4096, D=6
    X, Y = O.make_problem("c2", N=4096)
    om, mean = O.obs_mean_data(Y)
    h = _capi.Handle(eng, 0)
    h.set_data(X, om)
    th = np.zeros(7)
    h.set_kernel(O.SE_ARD, th, 0.01)
    h.compute(); h.log_lik_grad(False)
    n = 3 if args.quick else 10
    t0 = time.perf_counter()
    for i in range(n):
        h.set_kernel(O.SE_ARD, th + 1e-3 * i, 0.01)
        h.compute(); h.log_lik(); g = h.log_lik_grad(False)
    dt = time.perf_counter() - t0
    out["c2_hp_objective_with_grad_per_s"] = n / dt
    out["c2_hp_objective_ms"] = 1e3 * dt / n
    # leave-one-out objective with gradient (kernel_loo_opt.hpp:77-95): compute + LOO + its gradient
    h.log_loo_cv(); h.log_loo_cv_grad(False)
    t0 = time.perf_counter()
    for i in range(n):
        h.set_kernel(O.SE_ARD, th + 1e-3 * i, 0.01)
        h.compute(); h.log_loo_cv(); g = h.log_loo_cv_grad(False)
    dt = time.perf_counter() - t0
    out["c2_loo_objective_with_grad_per_s"] = n / dt
    # SparsifiedGP::_sparsify at N = 4096 -> 200 (sparsified_gp.hpp default max_points)
    _capi.sparsify(eng, X[:300], 100)
    t0 = time.perf_counter(); keep = _capi.sparsify(eng, X, 200); dt = time.perf_counter() - t0
    out["c2_sparsify_4096_to_200_s"] = dt
    # batched query at N=4096 (M = 20000)
    M = 5000 if args.quick else 20000
    Xq = np.random.default_rng(1).uniform(0, 1, size=(M, 6))
    h.query_batch(Xq[:256])
    dt = 1e30
    for _ in range(3):  # best of 3: one call is ~10 ms, a host-side hiccup shows
        t0 = time.perf_counter(); kta, var = h.query_batch(Xq); dt = min(dt, time.perf_counter() - t0)
    out["c2_query_batch_points_per_s_N4096"] = M / dt
    h.close()

    # config 3: N=16384, D=12, Matern5/2: compute + batched query
    N3 = 8192 if args.quick else 16384
    rng = np.random.default_rng(3)
    X3 = rng.uniform(0, 1, size=(N3, 12))
    Y3 = (np.cos(2 * X3).sum(axis=1) + 0.05 * rng.normal(size=N3))[:, None]
    om3, mean3 = O.obs_mean_data(Y3)
    h = _capi.Handle(eng, 0)
    h.set_data(X3, om3)
    h.set_kernel(O.MATERN52, np.zeros(2), 0.01)
    t0 = time.perf_counter(); info = h.compute(); ll = h.log_lik(); dt = time.perf_counter() - t0
    out["c3_N"] = N3
    out["c3_compute_loglik_s"] = dt
    out["c3_info"] = int(info)
    M3 = 10000 if args.quick else 100000
    Xq3 = rng.uniform(0, 1, size=(M3, 12))
    h.query_batch(Xq3[:256])
    dt = 1e30
    for _ in range(2):
        t0 = time.perf_counter(); kta, var = h.query_batch(Xq3); dt = min(dt, time.perf_counter() - t0)
    out["c3_query_points"] = M3
    out["c3_query_batch_s"] = dt
    out["c3_query_points_per_s"] = M3 / dt
    out["c3_query_tflops"] = 1.0 * M3 * N3 * N3 / dt / 1e12
    h.close()

    # config 4 on one GPU: 8 independent GPs N=2048 via batch_compute
    X4, Y4 = O.make_problem("c2", N=2048)
    om4, _ = O.obs_mean_data(Y4)
    hs = []
    for g_ in range(8):
        hh = _capi.Handle(eng, 0)
        hh.set_data(X4, om4)
        hh.set_kernel(O.SE_ARD, np.zeros(7) + 1e-2 * g_, 0.01)
        hs.append(hh)
    _capi.batch_compute(hs) if hasattr(_capi, "batch_compute") else [x.compute() for x in hs]
    reps = 2 if args.quick else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        if hasattr(_capi, "batch_compute"):
            _capi.batch_compute(hs)
        else:
            [x.compute() for x in hs]
        [x.log_lik() for x in hs]
    dt = time.perf_counter() - t0
    out["c4_8gps_N2048_evals_per_s"] = 8 * reps / dt
    for hh in hs:
        hh.close()

    # config 5: incremental add_sample, n = 10 -> 200, D = 6
    X5, Y5 = O.make_problem("c2", N=200)
    h = _capi.Handle(eng, 0)
    h.set_kernel(O.SE_ARD, np.zeros(7), 0.01)
    om5, _ = O.obs_mean_data(Y5[:10])
    h.set_data(X5[:10], om5)
    h.compute()
    t0 = time.perf_counter()
    for i in range(10, 200):
        omi, _ = O.obs_mean_data(Y5[: i + 1])
        h.add_sample(X5[i], omi)
    dt = time.perf_counter() - t0
    out["c5_add_sample_per_s"] = 190 / dt
    h.close()

    # per-point query latency (gp.hpp:159-191 as an acquisition functor calls it: one point, mu and sigma^2,
    # host to host), N = 200 (the end of a BO run), 1000, 4096
    lat, upd = {}, {}
    for n_ in (200, 1000, 4096):
        Xn, Yn = O.make_problem("c2", N=n_)
        omn, _ = O.obs_mean_data(Yn)
        hq = _capi.Handle(eng, 0)
        hq.set_kernel(O.SE_ARD, np.zeros(7), 0.01)
        hq.set_data(Xn, omn)
        hq.compute()
        pts = np.random.default_rng(3).uniform(0, 1, size=(220, 6))
        for i in range(20):
            hq.query_batch(pts[i:i + 1])
        best = 1e30
        for i0 in (20, 120):  # best of two blocks of 100
            t0 = time.perf_counter()
            for i in range(i0, i0 + 100):
                hq.query_batch(pts[i:i + 1])
            best = min(best, time.perf_counter() - t0)
        lat[str(n_)] = 1e6 * best / 100
        # recompute(false, false) + compute_log_lik: new observations, same factor (gp.hpp:241-252, :605-611)
        hq.update_alpha(omn)
        t0 = time.perf_counter()
        for i in range(50):
            hq.update_alpha(omn)
            hq.log_lik()
        upd[str(n_)] = 1e6 * (time.perf_counter() - t0) / 50
        hq.close()
    out["single_point_query_latency_us"] = lat
    out["update_alpha_loglik_latency_us"] = upd

    # CPU upper bound at config 2: numpy kernel build + LAPACK dpotrf/dpotrs on all
    # host threads — a reported side-by-side figure, not the oracle and not on any product path
    import os
    import scipy.linalg as sl
    Xc, Yc = O.make_problem("c2", N=4096)
    omc, _ = O.obs_mean_data(Yc)
    best = 1e30
    for _ in range(2):
        t0 = time.perf_counter()
        sq = (Xc * Xc).sum(axis=1)
        K = np.exp(-0.5 * np.maximum(sq[:, None] + sq[None, :] - 2.0 * (Xc @ Xc.T), 0.0))  # SE-ARD, ell = 1, sigma_f = 1
        K[np.diag_indices(len(Xc))] += 0.01 + 1e-8
        L = sl.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
        al = sl.cho_solve((L, True), omc, check_finite=False)
        ll = -0.5 * float((omc * al).sum()) - float(np.log(np.diag(L)).sum()) - 0.5 * len(Xc) * np.log(2 * np.pi)
        best = min(best, time.perf_counter() - t0)
    out["c2_cpu_lapack_evals_per_s"] = 1.0 / best
    out["cpu_threads"] = os.cpu_count()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
