#!/usr/bin/env python
"""Secondary measurements of SURVEY.md §8(d) on one MI355X (not the driver's bench.py contract):
HP-objective evaluations/s with gradient (config 2), config 3 (N=16384 factorisation + 100k batched queries),
config 4 (64 independent GPs on one GPU through one batched launch sequence), config 5 (add_sample / point-query
latency: the one-launch small path and the general path), each with its fraction of the fp64 matrix-core peak where
that is the bound, and a same-box CPU figure beside it (`cpu_*`: the C oracle — the checker, used here only as the
timed CPU baseline, as in bench.py — on one core / as 64 host tasks, and LAPACK on all cores).  Prints one JSON object."""
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
    PEAK = 78.6e12  # fp64 matrix-core peak (bench.py)
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
    out["c3_compute_tflops"] = (N3 ** 3 / 3.0 + 2.0 * N3 * N3) / dt / 1e12
    out["c3_compute_frac_of_fp64_peak"] = (N3 ** 3 / 3.0 + 2.0 * N3 * N3) / dt / PEAK
    # the trailing updates of this factorisation, each launch timed alone with HIP events on the handle's stream (as bench.py's
    # `roofline` does at N = 4096): 63 launches, 1.4e12 flop
    h.set_profiling(True)
    h.reset_phase_ms()
    h.compute()
    ph3 = h.get_phase_ms()["potrf_update"]
    h.set_profiling(False)
    out["c3_trailing_update_tflops"] = ph3["flops"] / (ph3["ms"] * 1e-3) / 1e12
    out["c3_trailing_update_frac_of_fp64_peak"] = ph3["flops"] / (ph3["ms"] * 1e-3) / PEAK
    out["c3_trailing_update_launches"] = ph3["launches"]
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
    out["c3_query_frac_of_fp64_peak"] = 1.0 * M3 * N3 * N3 / dt / PEAK
    if not args.quick:
        # same box, all host cores: LAPACK dpotrf on the same K (the factorisation only: the bounded sample)
        import os
        import scipy.linalg as sl
        from scipy.spatial.distance import cdist
        d3 = cdist(X3, X3)
        t1 = np.sqrt(5.0) * d3
        K3 = (1.0 + t1 + 5.0 * d3 * d3 / 3.0) * np.exp(-t1)
        del d3, t1
        K3[np.diag_indices(N3)] += 0.01 + 1e-8
        t0 = time.perf_counter(); L3 = sl.cholesky(K3, lower=True, overwrite_a=True, check_finite=False); dtc = time.perf_counter() - t0
        out["c3_cpu_lapack_dpotrf_s"] = dtc
        out["c3_cpu_lapack_cores"] = os.cpu_count()
        out["c3_cpu_note"] = "factorisation only (kernel build and solves excluded): scipy/OpenBLAS dpotrf of the same 16384 x 16384 K on all host cores"
        del K3, L3
    h.close()

    # config 4 on one GPU: 64 independent GPs, N=2048, D=6, one batched launch sequence (gpe_batch_compute)
    X4, Y4 = O.make_problem("c4", N=2048)
    G4 = 16 if args.quick else 64
    rng4 = np.random.default_rng(4)
    hs, oms4, ths4 = [], [], []
    for g_ in range(G4):
        om4, _ = O.obs_mean_data(Y4 * rng4.uniform(0.5, 1.5) + 0.1 * np.sin(3.0 * X4[:, g_ % 6: g_ % 6 + 1] + g_))
        th4 = rng4.uniform(-1e-2, 1e-2, size=7)
        hh = _capi.Handle(eng, 0)
        hh.set_data(X4, om4)
        hh.set_kernel(O.SE_ARD, th4, 0.01)
        hs.append(hh)
        oms4.append(om4)
        ths4.append(th4)
    _capi.batch_compute(hs)
    reps = 2 if args.quick else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        _capi.batch_compute(hs)
        ll4 = _capi.batch_log_lik(hs)
    dt = time.perf_counter() - t0
    fl4 = 2048 ** 3 / 3.0 + 2.0 * 2048 * 2048
    out["c4_gps"] = G4
    out["c4_evals_per_s"] = G4 * reps / dt
    out["c4_tflops"] = G4 * reps * fl4 / dt / 1e12
    out["c4_frac_of_fp64_peak"] = G4 * reps * fl4 / dt / PEAK
    for hh in hs:
        hh.close()
    if not args.quick:
        # same box: the reference's own strategy (multi_gp.hpp:124-126, one GP = one host task): the C oracle, one core per
        # GP, 64 tasks at once
        import os
        from concurrent.futures import ThreadPoolExecutor
        from oracle import binding as OB  # the checker, here only as the timed CPU baseline

        orc = OB.load_oracle()
        ohs = []
        for g_ in range(G4):
            oh = _capi.Handle(orc)
            oh.set_data(X4, oms4[g_])
            oh.set_kernel(O.SE_ARD, ths4[g_], 0.01)
            ohs.append(oh)

        def one(oh):
            oh.compute()
            return oh.log_lik()

        with ThreadPoolExecutor(max_workers=G4) as ex:
            t0 = time.perf_counter(); llo = list(ex.map(one, ohs)); dtc = time.perf_counter() - t0
        out["c4_cpu_64_host_tasks_evals_per_s"] = G4 / dtc
        out["c4_cpu_cores_used"] = min(G4, os.cpu_count())
        out["c4_cpu_vs_gpu_max_rel_diff_log_lik"] = float(np.max(np.abs(np.array(llo) - ll4) / np.abs(np.array(llo))))
        for oh in ohs:
            oh.close()

    # config 5: add_sample 10 -> 200 and the one-point query at N = 200: small path, general path, CPU oracle on one core
    import os
    import subprocess
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "small_bench.py")], capture_output=True, text=True, timeout=600)
    try:
        out["c5"] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        out["c5_error"] = (r.stdout + r.stderr)[-500:]

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
