#!/usr/bin/env python
"""bench.py — GP compute()+compute_log_lik() evaluations/sec at N=4096, D=6, SE-ARD, fp64.

One "step" = one pass of the hot path over one synthetic regression problem already resident
in HBM: kernel-matrix build -> blocked Cholesky -> alpha -> log marginal likelihood
(limbo::model::GP::compute + compute_log_lik, src/limbo/model/gp.hpp:88-116, :267-282), i.e.
BASELINE.json configs[1] ("N=4096 D=6 SquaredExpARD ... fp64, 1xMI355X").

--gpus N > 1 (launched by torch.distributed.run, one rank per GPU): every rank evaluates its
own independent GP (a hyper-parameter restart of the same data: opt/parallel_repeater.hpp:86-105),
no data-path collective; the only RCCL traffic is the final all-gather of (log_lik, theta) for
the arg-max (tools/parallel.hpp:169-191 `par::max`), issued inside the timed region.
value = (N ranks x K steps) / max-over-ranks time.  Scaling is weak.

Besides the headline (`value`: X resident in HBM when the timed region starts) the line carries
`value_incl_h2d` (the same step with gpe_set_data — 196 KB of X + 32 KB of obs_mean over PCIe — inside the timed
region: the reference's compute() includes that ingest, gp.hpp:88-116), `config4` (BASELINE configs[3] as written:
8 independent GPs of N=2048 per GPU through ONE gpe_batch_compute, aggregated over the ranks => 64 GPs on 8 GPUs),
`roofline`, and two CPU baselines timed on this box (single-core port; all-core numpy + LAPACK).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FP64_MFMA_PEAK_TF = 78.6  # 32 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz; v_mfma_f64_4x4x4_4b measures 76-77 (tools/ubench3)
N_C2, D_C2 = 4096, 6


def bo_inner_loop(lib, _capi, O, device):
    """The BO inner loop of src/benchmarks/limbo/bench.cpp:66-84 / bayes_opt/boptimizer.hpp:148-161 through the C-ABI of
    `lib` (the HIP engine; in the cpu_baseline leg the oracle, for the same-box figure)."""
    n0, n1 = 10, 200
    rng5 = np.random.default_rng(5)
    X5 = rng5.uniform(0, 1, size=(n1, D_C2))
    Y5 = O.hartmann6(X5)[:, None]
    oms = [np.asfortranarray(O.obs_mean_data(Y5[: k + 1])[0]) for k in range(n1)]
    xs = [np.ascontiguousarray(X5[k]) for k in range(n1)]
    best_add = 1e30
    h5 = None
    for _ in range(5):
        if h5 is not None:
            h5.close()
        h5 = _capi.Handle(lib, device)
        h5.set_kernel(O.SE_ARD, np.zeros(D_C2 + 1), 0.01)
        h5.set_data(X5[:n0], oms[n0 - 1])
        h5.compute()
        t0 = time.perf_counter()
        for k in range(n0, n1):
            h5.add_sample(xs[k], oms[k])
        best_add = min(best_add, time.perf_counter() - t0)
    pts = [np.ascontiguousarray(p_[None, :]) for p_ in rng5.uniform(0, 1, size=(350, D_C2))]
    for p_ in pts[:50]:
        h5.query_batch(p_)
    best_q = 1e30
    for i0 in (50, 150, 250):
        t0 = time.perf_counter()
        for p_ in pts[i0: i0 + 100]:
            h5.query_batch(p_)
        best_q = min(best_q, time.perf_counter() - t0)
    Xb = rng5.uniform(0, 1, size=(20000, D_C2))
    h5.query_batch(Xb[:256])
    t0 = time.perf_counter()
    h5.query_batch(Xb)
    dtb = time.perf_counter() - t0
    ll = h5.log_lik()
    h5.close()
    return {"add_sample_per_s": (n1 - n0) / best_add, "add_sample_us": 1e6 * best_add / (n1 - n0), "one_point_query_us": 1e6 * best_q / 100,
            "batched_query_points_per_s_n200": 20000 / dtb, "log_lik": ll}


def bo_iteration(lib, _capi, O, device, n_candidates=10000):
    """ONE iteration of the Bayesian-optimisation loop as the reference runs it (bayes_opt/boptimizer.hpp:148-161,
    src/benchmarks/limbo/bench.cpp:66-84): maximise the acquisition over the model — here acqui::UCB (acqui/ucb.hpp:83-90:
    mu + alpha sqrt(sigma^2)) over a BatchGridSearch-sized candidate set through ONE gpe_query_batch (the drop-in's
    acqui::UCB::batch + opt::BatchGridSearch) — then add_sample() of the winner.  Through the C-ABI of `lib`
    (the HIP engine; in the cpu_baseline leg the oracle on one core, same caller).  Mean of 3 consecutive iterations from
    n = 50 / 100 / 200 samples."""
    rng = np.random.default_rng(55)
    Xall = rng.uniform(0, 1, size=(260, D_C2))
    Yall = O.hartmann6(Xall)[:, None]
    Xc = rng.uniform(0, 1, size=(n_candidates, D_C2))
    alpha_ucb = 0.5  # defaults::acqui_ucb::alpha
    res = {}
    for n in (50, 100, 200):
        h = _capi.Handle(lib, device)
        h.set_kernel(O.SE_ARD, np.zeros(D_C2 + 1), 0.01)
        X, Y = list(Xall[:n]), list(Yall[:n, 0])
        om, mean = O.obs_mean_data(np.array(Y)[:, None])
        h.set_data(np.array(X), om)
        h.compute()
        h.query_batch(Xc[:256])
        t_q = t_add = 0.0
        iters = 3
        for _ in range(iters):
            t0 = time.perf_counter()
            kta, var = h.query_batch(Xc)
            mu, s2 = O.finish_query(kta, var, mean, 0.01)
            best = int(np.argmax(mu[:, 0] + alpha_ucb * np.sqrt(s2)))
            t1 = time.perf_counter()
            X.append(Xc[best])
            Y.append(float(O.hartmann6(Xc[best:best + 1])[0]))
            om, mean = O.obs_mean_data(np.array(Y)[:, None])  # mean::Data: every obs_mean moves with the new observation
            h.add_sample(np.ascontiguousarray(Xc[best]), np.asfortranarray(om))
            t2 = time.perf_counter()
            t_q += t1 - t0
            t_add += t2 - t1
        h.close()
        res[f"n{n}"] = {"iteration_ms": 1e3 * (t_q + t_add) / iters, "acquisition_ms": 1e3 * t_q / iters, "add_sample_ms": 1e3 * t_add / iters,
                        "candidates_per_s": n_candidates * iters / t_q}
    res["candidates"] = n_candidates
    return res


def box_probe(eng, _capi, O, local_rank):
    """How fast is THIS box on latency-bound work?  One box in about ten of the pool runs the data-flow launches at half
    speed while its matrix-core and HBM rates are normal (profiles/r04_slow_box_observation.log: 457 / 476 evaluations/s
    instead of 840, some CUs taking 22-30 us for a 64 x 64 block factorisation instead of 6-7).  The probe is the engine's own
    one-launch factorisation at N = 1024 (16 chain hops, nothing else): 0.245-0.25 ms on the other boxes of rounds 3 and 4,
    0.36-0.38 ms on the slow ones."""
    import numpy as np

    X, Y = O.make_problem("c2", N=1024)
    om, _ = O.obs_mean_data(Y)
    h = _capi.Handle(eng, local_rank)
    h.set_kernel(O.SE_ARD, np.zeros(D_C2 + 1), 0.01)
    h.set_data(X, om)
    for _ in range(3):
        h.compute()
        h.log_lik()
    per = []
    for _ in range(20):
        t0 = time.perf_counter()
        h.compute()
        h.log_lik()
        per.append(time.perf_counter() - t0)
    h.close()
    ms = 1e3 * float(np.median(per))
    return {"n1024_compute_loglik_ms": ms, "usual_ms": 0.22, "slow_box": bool(ms > 0.28),
            "note": "compute()+log_lik at N = 1024 (ONE data-flow launch + the sweep: pure chain latency; 0.215-0.222 ms on the usual "
                    "boxes since the chain work of round 5, 0.245 before); boxes that read > 0.28 ms here ran every latency-bound figure "
                    "of this line at about half speed in rounds 3-5 (one box in ten of the pool), with `roofline.trailing_update` unaffected"}


def extras(eng, _capi, O, local_rank, steps):
    """Secondary objects of the bench line, N=1 only, outside the headline's timed region: BASELINE configs[1] as written (the gradient objective of one KernelLFOpt iteration),
    configs[2] (N=16384 factorisation + 100k batched queries), configs[3] on one GPU (64 GPs), configs[4] (BO inner loop)."""
    import torch

    out = {}
    PEAK = FP64_MFMA_PEAK_TF * 1e12

    # ---- configs[1] as written: one KernelLFOptimization::operator() (kernel_lf_opt.hpp:77-92) = set theta ->
    # recompute -> log-lik -> K^-1 -> gradient, on resident buffers
    X, Y = O.make_problem("c2", N=N_C2)
    om, _ = O.obs_mean_data(Y)
    h = _capi.Handle(eng, local_rank)
    h.set_data(X, om)
    th = np.zeros(D_C2 + 1)
    h.hp_objective(O.SE_ARD, th, 0.01, optimize_noise=False, want_grad=True)
    n = max(5, steps // 2)
    per = []
    for i in range(n):
        t0 = time.perf_counter()
        ll, g, info = h.hp_objective(O.SE_ARD, th + 1e-3 * (i + 1), 0.01, optimize_noise=False, want_grad=True)
        per.append(time.perf_counter() - t0)
    assert info == 0 and np.all(np.isfinite(g))
    dt = float(np.sum(per))
    fl = float(N_C2) ** 3 / 3.0 + 2.0 * float(N_C2) ** 3 / 3.0  # factorisation + K^-1 = L^-T L^-1 (triangular products)
    h.set_profiling(True)
    h.reset_phase_ms()
    h.hp_objective(O.SE_ARD, th, 0.01, optimize_noise=False, want_grad=True)
    ph = h.get_phase_ms()
    h.set_profiling(False)
    out["hp_objective"] = {
        "workload": f"configs[1] as written: SquaredExpARD N={N_C2} D={D_C2} + one KernelLFOpt objective evaluation "
                    "(gpe_hp_objective: new theta -> K -> L -> alpha -> log-lik -> K^-1 -> d log-lik / d theta), resident buffers",
        "value": n / dt, "unit": "objective evaluations/s", "ms_per_evaluation": 1e3 * dt / n, "median_ms": 1e3 * float(np.median(per)),
        "roofline": {"bound": "mfma", "achieved": fl * n / dt / 1e12, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                     "frac": fl * n / dt / PEAK, "algorithmic_flops": fl,
                     "note": "N^3/3 (Cholesky) + 2 N^3/3 (K^-1 from L, triangular) per evaluation; the O(N^2 D) build and pair sums not counted"},
        "phases_ms_profiled": {k: v["ms"] for k, v in ph.items() if v["launches"]},
        "log_lik": ll, "grad_norm": float(np.linalg.norm(g)),
    }
    h.close()

    # ---- the same objective for G restarts in lock-step (opt/parallel_repeater.hpp:84-105 as the drop-in runs it:
    # opt/batched_rprop.hpp -> gpe_batch_hp_objective, one launch sequence with gridDim.z = restart), against the
    # restarts as host threads with one launch chain each (rounds 1-2)
    import threading

    def batched(G, N, reps):
        Xb, Yb = O.make_problem("c2" if N == N_C2 else "c4", N=N)
        omb, _ = O.obs_mean_data(Yb)
        rngb = np.random.default_rng(77)
        hs = []
        for _ in range(G):
            hb = _capi.Handle(eng, local_rank)
            hb.set_data(Xb, omb)
            hs.append(hb)
        th = rngb.uniform(-1e-2, 1e-2, size=(G, D_C2 + 1))
        _capi.batch_hp_objective(hs, O.SE_ARD, th, 0.01, want_grad=True)
        t0 = time.perf_counter()
        for r in range(reps):
            lk, gr, st = _capi.batch_hp_objective(hs, O.SE_ARD, th + 1e-3 * (r + 1), 0.01, want_grad=True)
        dtb = time.perf_counter() - t0
        assert all(s_ == 0 for s_ in st) and np.all(np.isfinite(gr))
        # the same restarts as host threads (one handle, one launch chain each)
        def worker(hb, row):
            for r in range(reps):
                hb.hp_objective(O.SE_ARD, th[row] + 1e-3 * (r + 1), 0.01, want_grad=True)
        ths = [threading.Thread(target=worker, args=(hb, i)) for i, hb in enumerate(hs)]
        t0 = time.perf_counter()
        for t_ in ths:
            t_.start()
        for t_ in ths:
            t_.join()
        dtt = time.perf_counter() - t0
        for hb in hs:
            hb.close()
        flb = float(N) ** 3
        return {"restarts": G, "N": N, "value": G * reps / dtb, "unit": "objective evaluations/s", "ms_per_batch": 1e3 * dtb / reps,
                "tflops": G * reps * flb / dtb / 1e12, "frac_of_fp64_peak": G * reps * flb / dtb / PEAK,
                "threads_one_chain_each_per_s": G * reps / dtt}

    out["batched_hp_objective"] = {
        "workload": "G hyper-parameter restarts of configs[1] / configs[3] in lock-step: every member's K -> L -> alpha -> log-lik -> "
                    "K^-1 -> gradient by ONE launch sequence (gpe_batch_hp_objective); flops = N^3 per member and evaluation",
        "c2_10_restarts": batched(10, N_C2, 4), "c4_64_restarts": batched(64, 2048, 3)}
    torch.cuda.empty_cache()

    # ---- configs[1] end to end: a KernelLFOpt fit by the reference's benchmark protocol — Rprop, 50 iterations
    # (waf_tools/benchmark_template.cpp:67, regression_benchmarks.json) x 10 restarts (opt/parallel_repeater.hpp:66,84-105),
    # the restarts in lock-step, every iteration ONE gpe_batch_hp_objective (limbo_amd/hpfit.py = the drop-in's
    # opt/batched_rprop.hpp)
    from limbo_amd import hpfit

    R_fit, it_fit = 10, 50
    hs = []
    for _ in range(R_fit):
        hb = _capi.Handle(eng, local_rank)
        hb.set_data(X, om)
        hs.append(hb)
    inits = np.random.default_rng(66).uniform(-1e-2, 1e-2, size=(R_fit, D_C2 + 1))  # parallel_repeater.hpp:88
    hpfit.kernel_lf_opt_lockstep(hs, O.SE_ARD, inits, iterations=2)
    tr = []
    t0 = time.perf_counter()
    bp, bl = hpfit.kernel_lf_opt_lockstep(hs, O.SE_ARD, inits, noise=0.01, optimize_noise=False, iterations=it_fit, eps_stop=0.0, trace=tr)
    dt_fit = time.perf_counter() - t0
    out["c2_fit"] = {"workload": f"configs[1] end to end: KernelLFOpt<Rprop> fit of the N={N_C2} D={D_C2} SquaredExpARD GP, {it_fit} iterations x "
                                 f"{R_fit} restarts (the reference's benchmark protocol), restarts in lock-step through gpe_batch_hp_objective",
                     "wall_s": dt_fit, "iterations": len(tr), "restarts": R_fit, "objective_evaluations_per_s": R_fit * len(tr) / dt_fit,
                     "ms_per_iteration": 1e3 * dt_fit / max(len(tr), 1), "frac_of_fp64_peak": R_fit * len(tr) * float(N_C2) ** 3 / dt_fit / PEAK,
                     "log_lik_start": float(tr[0][1].max()), "log_lik_best": float(bl.max()),
                     "reruns": int(sum(hb.flow_retries() for hb in hs))}
    for hb in hs:
        hb.close()
    torch.cuda.empty_cache()

    # ---- configs[2]: N=16384, D=12, Matern-5/2: compute()+log_lik, its trailing updates alone, 100k batched queries
    N3, D3, M3 = 16384, 12, 100000
    X3, Y3 = O.make_problem("c3")
    om3, _ = O.obs_mean_data(Y3)
    h = _capi.Handle(eng, local_rank)
    h.set_data(X3, om3)
    h.set_kernel(O.MATERN52, np.zeros(2), 0.01)
    h.compute()
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        info3 = h.compute()
        ll3 = h.log_lik()
        best = min(best, time.perf_counter() - t0)
    fl3 = float(N3) ** 3 / 3.0 + 2.0 * float(N3) * N3
    h.set_profiling(True)
    h.reset_phase_ms()
    h.compute()
    ph3 = h.get_phase_ms()["potrf_update"]
    h.set_profiling(False)
    Xq3 = np.random.default_rng(3).uniform(0, 1, size=(M3, D3))
    h.query_batch(Xq3[:512])
    bq = 1e30
    for _ in range(3):  # (the first full-size call sizes the query buffers: 0.75 s against 0.50)
        t0 = time.perf_counter()
        kta3, var3 = h.query_batch(Xq3)
        bq = min(bq, time.perf_counter() - t0)
    assert info3 == 0 and np.isfinite(ll3) and np.all(np.isfinite(var3))
    out["config3"] = {
        "workload": f"configs[2]: Matern5/2 GP, N={N3}, D={D3}, fp64: compute()+log_lik (best of 3), then mu/sigma^2 for {M3} query "
                    "points through gpe_query_batch (best of 3, host to host incl. the PCIe copies of the points and results)",
        "compute_loglik_ms": 1e3 * best, "factorisation_tflops": fl3 / best / 1e12, "factorisation_frac_of_fp64_peak": fl3 / best / PEAK,
        "trailing_update": {"tflops": ph3["flops"] / (ph3["ms"] * 1e-3) / 1e12, "frac": ph3["flops"] / (ph3["ms"] * 1e-3) / PEAK,
                            "launches": ph3["launches"], "note": "every launch alone, HIP events on the handle's stream (as `roofline`)"},
        "query_points_per_s": M3 / bq, "query_s": bq, "query_tflops": 1.0 * M3 * N3 * N3 / bq / 1e12,
        "query_frac_of_fp64_peak": 1.0 * M3 * N3 * N3 / bq / PEAK, "log_lik": ll3,
    }
    h.close()
    torch.cuda.empty_cache()

    # ---- configs[3] on ONE GPU: all 64 GPs of N=2048 through one batched launch sequence
    G4, N4 = 64, 2048
    X4, Y4 = O.make_problem("c4", N=N4)
    rng4 = np.random.default_rng(4)
    hs = []
    for g_ in range(G4):
        om4, _ = O.obs_mean_data(Y4 * rng4.uniform(0.5, 1.5) + 0.1 * np.sin(3.0 * X4[:, g_ % 6: g_ % 6 + 1] + g_))
        hh = _capi.Handle(eng, local_rank)
        hh.set_data(X4, om4)
        hh.set_kernel(O.SE_ARD, rng4.uniform(-1e-2, 1e-2, size=D_C2 + 1), 0.01)
        hs.append(hh)
    _capi.batch_compute(hs)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        st4 = _capi.batch_compute(hs)
        ll4 = _capi.batch_log_lik(hs)
    dt4 = time.perf_counter() - t0
    assert all(s_ == 0 for s_ in st4) and np.all(np.isfinite(ll4))
    fl4 = float(N4) ** 3 / 3.0 + 2.0 * float(N4) * N4
    out["config4_g64"] = {"workload": f"configs[3] on one GPU: {G4} independent SquaredExpARD GPs, N={N4}, D={D_C2}, compute()+log_lik each, "
                                      "one gpe_batch_compute", "value": G4 * reps / dt4, "unit": "evaluations/s",
                          "ms_per_batch": 1e3 * dt4 / reps, "tflops": G4 * reps * fl4 / dt4 / 1e12, "frac_of_fp64_peak": G4 * reps * fl4 / dt4 / PEAK}
    for hh in hs:
        hh.close()

    # ---- configs[4]: the BO inner loop (benchmarks/limbo/bench.cpp:66-84): add_sample 10 -> 200, one-point queries at n = 200
    out["config5"] = dict(bo_inner_loop(eng, _capi, O, local_rank), bo_iteration=bo_iteration(eng, _capi, O, local_rank),
                          workload="configs[4] regime: SquaredExpARD D=6, noise 0.01, add_sample() n = 10 -> 200 (best of 5 loops), one-point "
                                   "mu+sigma^2 at n = 200 (best of 3 blocks of 100), host to host through ctypes; fp64 throughout (bf16: DESIGN §10)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=N_C2, help="override N (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="profiling runs: only the N=4096 evaluation loop (no H2D variant, no config 4)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary objects (hp_objective, config3, config4_g64, config5)")
    ap.add_argument("--c3-n", type=int, default=16384, help="config3_sharded: samples (debug / tests: a reduced N)")
    ap.add_argument("--c3-m", type=int, default=100000, help="config3_sharded: query points in all")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group (and run barrier / all_gather / all_reduce) even with ONE rank: the RCCL "
                         "branch rehearsed on a one-GPU box (tests/test_gpu_configs.py::test_gpu_bench_one_rank_rccl)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for --gpus > 1: nccl (= RCCL over xGMI, the default) or gloo (CPU tensors for the two "
                         "collectives; lets several ranks share one visible GPU: tests/test_gpu_configs.py)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch

    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # ranks may outnumber the visible GPUs only with --dist-backend gloo (the one-GPU rehearsal of the multi-rank path)
    n_vis = torch.cuda.device_count()
    if local_rank >= n_vis:
        assert args.dist_backend == "gloo", f"rank {rank}: LOCAL_RANK {local_rank} but {n_vis} visible GPU(s)"
        local_rank %= n_vis
    torch.cuda.set_device(local_rank)
    dist = None
    coll_dev = f"cuda:{local_rank}" if args.dist_backend == "nccl" else "cpu"  # where the collectives' tensors live
    executed = []  # the collectives this run actually issued
    if world > 1 or args.force_dist:
        import torch.distributed as dist_

        dist = dist_
        if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:  # --force-dist outside torch.distributed.run
            import socket

            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from limbo_amd import _capi
    from limbo_amd import synth as O  # synthetic-problem generator (pure numpy)

    eng = _capi.load_engine()
    N = args.n
    X, Y = O.make_problem("c2", N=N)
    om, mean = O.obs_mean_data(Y)
    # this rank's restart: theta0 + U[-eps, eps] (parallel_repeater.hpp:88, epsilon = 1e-2)
    rng = np.random.default_rng(1000 + rank)
    theta = np.zeros(D_C2 + 1) + (rng.uniform(-1e-2, 1e-2, size=D_C2 + 1) if world > 1 else 0.0)
    h = _capi.Handle(eng, local_rank)
    h.set_kernel(O.SE_ARD, theta, 0.01)
    h.set_data(X, om)  # X and obs_mean are resident in HBM from here on

    def step():
        info = h.compute()
        ll = h.log_lik()
        return info, ll

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
            if "barrier" not in executed:
                executed.append("barrier")

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        tmax = torch.tensor([seconds], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        if "all_reduce" not in executed:
            executed.append("all_reduce")
        return float(tmax.item())

    def argmax(values, thetas):
        res = PAR.argmax_over_ranks(values, thetas, dist, device=coll_dev, force=args.force_dist)
        if dist is not None and "all_gather" not in executed:
            executed.append("all_gather")
        return res

    from limbo_amd import parallel as PAR

    for _ in range(args.warmup):
        step()
    if dist is not None:
        # the collectives of the timed region once outside it: the first all_gather / all_reduce of a process group loads RCCL's
        # kernels and sets up its channels (milliseconds — 10 % of a 20-step region, profiles/r06_bench_ranks.log), which is
        # warm-up, not arg-max
        argmax([0.0], [theta])
        max_over_ranks(0.0)
    sync()
    t0 = time.perf_counter()
    ll = 0.0
    stamps = [t0]
    for _ in range(args.steps):
        info, ll = step()  # returns with the result on the host: every step ends synchronised
        stamps.append(time.perf_counter())
    # arg-max over the restarts: all-gather of (log_lik, theta) -- the only collective
    best_ll, best_theta, best_rank = argmax([ll], [theta])
    sync()
    dt = max_over_ranks(time.perf_counter() - t0)
    assert info == 0 and np.isfinite(ll), (info, ll)
    evals = world * args.steps
    value = evals / dt
    # spread of the K timed steps (this rank's host clock; each step ends with its result on the host): a scheduler
    # hiccup inside the timed region shows as max >> median instead of silently moving `value`
    per = np.diff(np.asarray(stamps)) * 1e3
    blocks = [float(np.sum(b)) / len(b) for b in np.array_split(per, min(5, len(per))) if len(b)]
    step_stats = {"median_ms": float(np.median(per)), "min_ms": float(per.min()), "max_ms": float(per.max()),
                  "p90_ms": float(np.percentile(per, 90)), "block_ms_per_step": blocks,
                  "value_from_median_step": world * 1e3 / float(np.median(per))}

    value_incl_h2d, config4 = None, None
    if not args.headline_only:
        # the same step with the host -> device ingest of (X, obs_mean) inside: gp.hpp:88-116 as the reference times it
        n_h2d = max(5, args.steps // 2)
        sync()
        t0 = time.perf_counter()
        for _ in range(n_h2d):
            h.set_data(X, om)
            step()
        sync()
        dt_h2d = max_over_ranks(time.perf_counter() - t0)
        value_incl_h2d = world * n_h2d / dt_h2d

        # BASELINE configs[3] as written: 64 independent GPs of N=2048, D=6 sharded 8 per GPU (multi_gp.hpp:124-126 /
        # parallel_repeater.hpp:86-105); each rank steps its 8 through ONE launch sequence (gpe_batch_compute), the final
        # arg-max over all of them is the same all-gather.  Weak scaling: 8 GPs per GPU at every N.
        G4, N4 = 8, 2048
        X4, Y4 = O.make_problem("c4", N=N4)
        rng4 = np.random.default_rng(4000 + rank)
        hs4, th4 = [], []
        for g4 in range(G4):
            om4, _ = O.obs_mean_data(Y4 * rng4.uniform(0.5, 1.5) + 0.1 * np.sin(3.0 * X4[:, g4 % 6: g4 % 6 + 1] + g4 + 8 * rank))
            t4 = rng4.uniform(-1e-2, 1e-2, size=D_C2 + 1)
            h4 = _capi.Handle(eng, local_rank)
            h4.set_kernel(O.SE_ARD, t4, 0.01)
            h4.set_data(X4, om4)
            hs4.append(h4)
            th4.append(t4)
        _capi.batch_compute(hs4)
        reps4 = max(3, args.steps // 5)
        sync()
        t0 = time.perf_counter()
        for _ in range(reps4):
            st4 = _capi.batch_compute(hs4)
            ll4 = _capi.batch_log_lik(hs4)
        best4 = argmax(list(ll4), th4)
        sync()
        dt4 = max_over_ranks(time.perf_counter() - t0)
        assert all(s4 == 0 for s4 in st4) and np.all(np.isfinite(ll4))
        for h4 in hs4:
            h4.close()
        config4 = {"workload": f"configs[3]: {G4 * world} independent SquaredExpARD GPs, N={N4}, D={D_C2}, {G4} per GPU, one batched launch "
                               "sequence per GPU (gpe_batch_compute), compute()+log_lik each, arg-max over all by all-gather",
                   "value": world * G4 * reps4 / dt4, "unit": "evaluations/s", "gps_total": G4 * world, "ms_per_batch": 1e3 * dt4 / reps4,
                   "tflops": world * G4 * reps4 / dt4 * (N4 ** 3 / 3.0 + 2.0 * N4 * N4) / 1e12, "best_log_lik": float(best4[0])}

    config3_sharded = None
    if not args.headline_only and dist is not None:
        # BASELINE configs[2] across ranks — SURVEY §8(e)'s second shard axis (multi_gp.hpp:191-195, tools/parallel.hpp:138-201: the
        # reference's parallel query): every rank factors its OWN replica of the N = 16384 Matern-5/2 GP (31 ms — cheaper than
        # broadcasting a 2 GiB factor over xGMI), answers row_slice(M, rank, world) of the 100 000 points with one
        # gpe_query_batch, ONE all-gather of (P + 1) doubles per point reassembles mu / sigma^2 on every rank.  Strong scaling:
        # the M points are fixed, the per-rank share shrinks.  Rank 0 then asks 512 points of the LAST rank's slice itself:
        # replicas and batch-invariant queries make the gathered answer bitwise its own.
        N3, M3 = args.c3_n, args.c3_m
        X3, Y3 = O.make_problem("c3", N=N3)
        om3, mean3 = O.obs_mean_data(Y3)
        h3 = _capi.Handle(eng, local_rank)
        h3.set_kernel(O.MATERN52, np.zeros(2), 0.01)
        h3.set_data(X3, om3)
        assert h3.compute() == 0  # warm (allocations, hand-over buffers)
        Xq3 = np.random.default_rng(20260928).uniform(0.0, 1.0, size=(M3, X3.shape[1]))  # the same points on every rank
        h3.query_batch(Xq3[: min(M3, 4096)])
        sync()
        t0 = time.perf_counter()
        info3 = h3.compute()
        ll3 = h3.log_lik()
        sync()
        dt3_fact = max_over_ranks(time.perf_counter() - t0)
        st3 = {}
        t0 = time.perf_counter()
        kta3, var3 = PAR.query_sharded(h3.query_batch, Xq3, dist, device=coll_dev, force=args.force_dist, stats=st3)
        sync()
        dt3_q = max_over_ranks(time.perf_counter() - t0)
        gather3 = max_over_ranks(st3["gather_s"])
        local3 = max_over_ranks(st3["local_s"])
        assert info3 == 0 and np.isfinite(ll3) and kta3.shape == (M3, 1) and var3.shape == (M3,) and np.all(np.isfinite(var3))
        sl_last = PAR.row_slice(M3, world - 1, world)
        chk = slice(sl_last.start, min(sl_last.stop, sl_last.start + 512))
        k_own, v_own = h3.query_batch(Xq3[chk])
        same = bool(np.array_equal(np.asarray(k_own).reshape(-1), kta3[chk, 0]) and np.array_equal(np.asarray(v_own).reshape(-1), var3[chk]))
        h3.close()
        config3_sharded = {
            "workload": f"configs[2] sharded over the query points: Matern5/2 GP, N={N3}, D={X3.shape[1]}, fp64, a replica per rank (each rank "
                        f"factors its own), {M3} query points dealt in contiguous slices over {world} rank(s) (row_slice), one all-gather of "
                        "(mu, sigma^2); strong scaling in M",
            "value": M3 / dt3_q, "unit": "query points/s (whole job, incl. the all-gather)", "points": M3, "points_per_rank": int(st3["points_local"]),
            "query_s": dt3_q, "local_query_s_max_over_ranks": local3, "gather_s_max_over_ranks": gather3,
            "replica_compute_loglik_ms": 1e3 * dt3_fact, "log_lik": float(ll3),
            "gathered_equals_own_answer_bitwise": same,
            "query_tflops": M3 / dt3_q * (float(N3) * N3 + 2.0 * N3) / 1e12,
        }
        assert same, "config3_sharded: the gathered answer of another rank's slice differs from this rank's own answer"

    out = {
        "metric": "GP compute()+log_lik evaluations/sec at N=4096 D=6 fp64",
        "value": value,
        "unit": "evaluations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (Hartmann6 + noise on U[0,1]^6, numpy default_rng(20260925 + config index = 20260927 — SURVEY §8(d) names mt19937_64 with the same seed rule; numpy's generator is what the committed goldens use), theta0 = 0",
        "config": {"workload": f"configs[1]: SquaredExpARD GP, N={N}, D={D_C2}, P=1, fp64, theta0=0, noise=0.01; "
                               "step = compute()+compute_log_lik(), X resident in HBM",
                   "parallelism": f"{world} independent GP restart(s), 1 per GPU, final RCCL all-gather arg-max"},
        "log_lik": ll,
        "argmax": {"best_log_lik": best_ll, "owner_rank": best_rank},
        "step_time_spread": step_stats,
        "value_incl_h2d": value_incl_h2d,
        "value_incl_h2d_note": "same step with gpe_set_data (X: 196 KB, obs_mean: 32 KB, host -> HBM) inside the timed region",
        "config4": config4,
        "config3_sharded": config3_sharded,
        "collectives": {"backend": args.dist_backend if dist is not None else None, "world": world,
                        "tensors_on": coll_dev if dist is not None else None, "executed": executed,
                        "note": "barrier + max-over-ranks all_reduce around every timed region, all_gather arg-max of (log_lik, theta) "
                                "inside it (tools/parallel.hpp:169-191); --force-dist runs them in a group of one rank"},
        # evaluations that had to be run a second time because a data-flow hand-over timed out (never expected: a silent 2x
        # slowdown on some future runtime would show here) / sweeps re-run block by block
        "handover_reruns": h.handover_reruns(), "flow_retries": h.flow_retries(),
    }

    if rank == 0 and not args.no_roofline:
        # dominant kernel: the Cholesky trailing update (k_gemm_glds / k_gemm_glds64 / k_gemm4, fp64 MFMA).  HIP events on
        # the handle's own stream around every launch of it (profiling mode: every phase alone on the chip), outside the
        # timed region.  Round 4: ALL of the factorisation is accounted for — the update launch(es), the two data-flow
        # launches (whose flops are 64^3 products inside a latency chain) and the whole-factorisation rate over the step.
        import ctypes as C

        def profiled(handle_, reps_=5):
            handle_.compute()
            handle_.set_profiling(True)
            handle_.reset_phase_ms()
            for _ in range(reps_):
                handle_.compute()
                handle_.log_lik()
            ph_ = handle_.get_phase_ms()
            handle_.set_profiling(False)
            return {k: {"us": 1e3 * v["ms"] / reps_, "launches": v["launches"] / reps_, "flops": v["flops"] / reps_} for k, v in ph_.items() if v["launches"]}

        def rate(rec):
            tf_ = rec["flops"] / (rec["us"] * 1e-6) / 1e12 if rec["us"] > 0 else 0.0
            return {"us": rec["us"], "launches": rec["launches"], "algorithmic_flops": rec["flops"], "tflops": tf_, "frac": tf_ / FP64_MFMA_PEAK_TF}

        ph = profiled(h)
        upd = ph.get("potrf_update", {"us": 0.0, "launches": 0, "flops": 0.0})
        fl_fact = float(N) ** 3 / 3.0 + 2.0 * float(N) * N  # N^3/3 + 2 N^2 P, P = 1
        tf = rate(upd)["tflops"]
        pk = C.c_double()
        eng.fn("mfma_f64_peak")(local_rank, C.byref(pk))
        # the round-3 accounting beside it: 256-column panels to the end (GPE_TALL=0, GPE_TAIL_MAX=0), all 15 trailing updates
        os.environ["GPE_TALL"], os.environ["GPE_TAIL_MAX"] = "0", "0"
        h15 = _capi.Handle(eng, local_rank)
        os.environ.pop("GPE_TALL"), os.environ.pop("GPE_TAIL_MAX")
        h15.set_kernel(O.SE_ARD, theta, 0.01)
        h15.set_data(X, om)
        ph15 = profiled(h15, 3)
        h15.close()
        # HBM traffic from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH doubled per
        # MI355X_MICROARCH.md): a file under profiles/, not measured in this run (a PMC pass cannot run inside it) — null when
        # the file is absent
        pmc, pmc_rec = None, {}
        for cand in ("r06_pmc_bench.json", "r05_pmc_bench.json", "r04_pmc_trailing_update.json"):
            if (ROOT / "profiles" / cand).exists() and N == N_C2:
                pmc = ROOT / "profiles" / cand
                pmc_rec = json.loads(pmc.read_text())
                break
        upd_roof = {
            "kernel": "k_gemm_glds (Cholesky trailing update A22 -= L21 L21^T, v_mfma_f64_4x4x4_4b): every trailing-update launch of one "
                      "factorisation, each alone, HIP events on the handle's stream.  At N = 4096: ONE update with k = 1280 between the "
                      "tall data-flow launch (columns 0..1279, all rows) and the closing one (columns 1280..4095)",
            "achieved": tf, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TF,
            "traffic": pmc_rec.get("hbm_bytes_per_launch_corrected"),
            "traffic_note": f"bytes of the update launch, PMC (profiles/{pmc.name if pmc else '-'}: committed, not measured in this run); "
                            "algorithmic bytes of that launch = 2 x 31.7 MB C tiles (lower triangle of 2816^2) + 28.8 MB panel (2816 x 1280)",
            "mfma_busy_percent": pmc_rec.get("mfma_util_percent"),
            "launches_per_step": upd["launches"], "avg_launch_us": upd["us"] / max(upd["launches"], 1),
            "algorithmic_flops_per_step": upd["flops"],
            "share_of_factorisation_flops": upd["flops"] / (float(N) ** 3 / 3.0),
        }
        total_us = sum(v["us"] for v in ph.values())
        if "potrf_tall" in ph and "potrf_tail" in ph:
            # Round 5: the line's `roofline` is the DOMINANT kernel — k_tail, the two data-flow launches, 70 %
            # of the GPU time of a step — not the update launch, which follows as `trailing_update`
            dfl = {"us": ph["potrf_tall"]["us"] + ph["potrf_tail"]["us"], "launches": ph["potrf_tall"]["launches"] + ph["potrf_tail"]["launches"],
                   "flops": ph["potrf_tall"]["flops"] + ph["potrf_tail"]["flops"]}
            kt = pmc_rec.get("k_tail") or {}
            out["roofline"] = {
                "bound": "mfma",
                "kernel": "k_tail (potrf.hip): the two data-flow launches of the factorisation — tall (columns 0..1279, every row strip below) "
                          "and closing (the last 2816 columns); a workgroup per 64 x 64 tile, 64^3 matrix-core products inside a latency "
                          "chain of 64 diagonal blocks (10.5-11.5 us each), operands polled between workgroups; each launch alone, HIP events on the handle's "
                          "stream.  flops = the factorisation flops of the columns each launch covers",
                "achieved": rate(dfl)["tflops"], "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": rate(dfl)["frac"],
                "traffic": kt.get("hbm_bytes_per_step_corrected"),
                "traffic_note": f"bytes of the two launches together, PMC (profiles/{pmc.name if pmc else '-'}: committed, not measured in this "
                                "run; FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md); algorithmic bytes: read + write of the lower triangle "
                                "each launch covers = 2 x 8 B x (4096^2 - 2816^2) / 2 + 2 x 8 B x 2816^2 / 2 = 134 MB (+ the hand-over slots)",
                "mfma_busy_percent": {"tall": (kt.get("tall") or {}).get("mfma_util_percent"), "closing": (kt.get("closing") or {}).get("mfma_util_percent")},
                "launches_per_step": dfl["launches"], "avg_launch_us": dfl["us"] / max(dfl["launches"], 1),
                "algorithmic_flops_per_step": dfl["flops"], "share_of_gpu_time": dfl["us"] / total_us if total_us > 0 else None,
                "share_of_factorisation_flops": dfl["flops"] / (float(N) ** 3 / 3.0),
                "trailing_update": upd_roof,
            }
        else:
            out["roofline"] = dict(upd_roof, bound="mfma", share_of_gpu_time=upd["us"] / total_us if total_us > 0 else None)
        out["roofline"].update({
            "data_flow_launches": {"tall": rate(ph["potrf_tall"]) if "potrf_tall" in ph else None,
                                   "closing": rate(ph["potrf_tail"]) if "potrf_tail" in ph else None,
                                   "note": "k_tail: a workgroup per 64 x 64 tile, operands polled between workgroups; flops = the "
                                           "factorisation flops of the columns the launch covers (64^3 products inside a latency chain)"},
            "factorisation": {"algorithmic_flops": fl_fact, "tflops_over_the_step": fl_fact / (dt / args.steps) / 1e12,
                              "frac_over_the_step": fl_fact / (dt / args.steps) / 1e12 / FP64_MFMA_PEAK_TF,
                              "note": "N^3/3 + 2 N^2 P over the whole timed step (kernel-matrix build, factorisation, both sweeps, host round trip)"},
            "all_updates_like_for_like": dict(rate(ph15["potrf_update"]), share_of_factorisation_flops=ph15["potrf_update"]["flops"] / (float(N) ** 3 / 3.0),
                                              note="GPE_TALL=0 GPE_TAIL_MAX=0: 256-column panels to the end, every one of the 15 trailing updates "
                                                   "(k = 256) alone between two events — rounds 1-3's accounting"),
            "measured_mfma_f64_4x4x4_peak_tflops": pk.value,
        })
        # VERDICT r5 (weak 8): the kernel-matrix build as north_star asks for it — HBM GB/s of the launch (each alone, HIP events),
        # against the 8 TB/s of the spec sheet and against the write stream this chip sustains (gpe_hbm_stream_peak: a plain
        # 16-byte-per-lane store kernel over 1 GiB, measured here).  Algorithmic bytes: the lower triangle out, the samples in.
        kb = ph.get("kernel_build")
        if kb and kb["us"] > 0:
            ws = C.c_double()
            eng.fn("hbm_stream_peak")(local_rank, C.byref(ws))
            kb_bytes = 8.0 * (N * (N + 1) / 2.0 + N * D_C2)
            gbps = kb_bytes / (kb["us"] * 1e-6) / 1e9
            out["kernel_build"] = {"kernel": "k_build_wide (kbuild.hip): lower triangle of K + obs_mean's rows under it, fp64 exp per pair",
                                   "us": kb["us"], "launches": kb["launches"], "bytes": kb_bytes, "GBps": gbps, "frac_of_hbm_spec": gbps / 8000.0,
                                   "measured_write_stream_GBps": ws.value, "frac_of_measured_write_stream": gbps / ws.value if ws.value > 0 else None,
                                   "note": "8.4 M pairs x (3 D + ~25 fp64 operations of the branch-free exp); floors: write stream 14 us, fp64 VALU "
                                           "~10 us.  Round 6 (profiles/r06_kernel_build.log): 32-34 -> 23-24 us (the profiled launch carries events: "
                                           "+2) by an unconditional pair loop over padded dimensions, a mask-free loop with 16-byte stores below the "
                                           "diagonal and 32-column workgroups; 2 % of the step"}
        sw = ph.get("solve")
        if sw and sw["us"] > 0:
            nblk_ = (N + 63) // 64
            out["backward_sweep"] = {"kernel": "k_trsv_bwd_m (sweep2.hip, round 6): a_j = g_j - M2_j a_{j+2} - M_j a_{j+1}, one data-flow launch",
                                     "us": sw["us"], "hops": nblk_, "us_per_hop": sw["us"] / nblk_, "bytes": 8.0 * N * (N + 1) / 2.0,
                                     "GBps": 8.0 * N * (N + 1) / 2.0 / (sw["us"] * 1e-6) / 1e9,
                                     "note": "bound by the latency of a chain of N / 64 hand-overs between workgroups, not by HBM (L is read once)"}
        out["factorisation_frac"] = out["roofline"]["factorisation"]["frac_over_the_step"]
        out["box_probe"] = box_probe(eng, _capi, O, local_rank)
        out["phases_us_per_step_profiled"] = {k: v["us"] for k, v in ph.items()}

    if rank == 0 and world == 1 and not args.no_roofline:
        # R independent evaluations in flight from R host threads (the reference runs hyper-parameter restarts concurrently:
        # opt/parallel_repeater.hpp:86-105 under tools::par::max).  Round 4 ordered the data-flow launches of different handles
        # chain behind chain (two at once can starve each other's lowest unfinished workgroup); round 5: while another chain
        # is in flight an evaluation runs on one of two CU-masked streams, half of every XCD's CUs each (csrc/engine.hip:
        # ChainScope) — two chains side by side.  Restarts in lock-step (one batched launch sequence) are `batched_hp_objective`.
        import threading

        conc = {}
        for R in (4, 8):
            hs = []
            for r in range(R):
                hr = _capi.Handle(eng, local_rank)
                hr.set_kernel(O.SE_ARD, theta + 1e-3 * r, 0.01)
                hr.set_data(X, om)
                hs.append(hr)
            per = max(4, args.steps // 2)

            def worker(hr):
                for _ in range(per):
                    hr.compute()
                    hr.log_lik()

            for hr in hs:  # warm-up: every handle alone ...
                hr.compute()

            def warm(hr):
                for _ in range(3):
                    hr.compute()

            ths = [threading.Thread(target=warm, args=(hr,)) for hr in hs]  # ... and three evaluations each TOGETHER, untimed (the
            for t in ths:                                                   # first chains a process runs side by side pay one-off
                t.start()                                                   # set-up, as the first steps of the headline do)
            for t in ths:
                t.join()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ths = [threading.Thread(target=worker, args=(hr,)) for hr in hs]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize()
            dtc = time.perf_counter() - t0
            conc[f"in_flight_{R}"] = R * per / dtc
            for hr in hs:
                hr.close()
        out["concurrent_evaluations_per_s"] = conc

    if rank == 0 and world == 1 and not args.no_extras and not args.headline_only and N == N_C2:
        out.update(extras(eng, _capi, O, local_rank, args.steps))

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU side-by-side: the oracle (C restatement of the reference's path, 1 thread = the
        # reference's default: one GP::compute is single-threaded) on a bounded sample.
        from oracle import binding as OB  # the checker, cpu_baseline leg only

        orc = OB.load_oracle()
        ho = _capi.Handle(orc)
        ho.set_kernel(O.SE_ARD, theta, 0.01)
        ho.set_data(X, om)
        n_cpu = 8  # ~11 s of single-core work
        t0 = time.perf_counter()
        for _ in range(n_cpu):
            ho.compute()
            llo = ho.log_lik()
        tc = time.perf_counter() - t0
        ho.close()
        # the reference ITSELF: limbo::model::GP compiled from the unmodified headers (oracle/_ref/libref.so, built in the
        # container where /root/reference is; the dense loops under its semantics are the Eigen stand-in's, one thread — the
        # reference's default: no TBB call inside gp.hpp, Eigen's LLT is serial without OpenMP/MKL).  ONE evaluation (~25 s).
        cpu_ref = None
        if OB.REF_SO.exists() and N == N_C2:
            rg = OB.RefGP(O.SE_ARD, D_C2, 1, noise=0.01)
            rg.set_h_params(theta)
            t0 = time.perf_counter()
            rg.compute(X, Y)
            llr = rg.log_lik()
            tr_ = time.perf_counter() - t0
            rg.close()
            cpu_ref = {"value": 1.0 / tr_, "unit": "evaluations/s", "cores": 1, "kind": "reference",
                       "sample": f"1 compute()+compute_log_lik() at N={N}, D={D_C2} through limbo's own gp.hpp (unmodified /root/reference/src "
                                 "headers compiled against the Eigen/Boost stand-ins of oracle/ref_build: oracle/_ref/libref.so)",
                       "log_lik": llr, "rel_diff_vs_gpu": abs(llr - ll) / abs(llr)}
        out["cpu_baseline_reference"] = cpu_ref
        # LAPACK on the host cores: an upper bound for a CPU, not the reference's code path.  The parts
        # separately — round 3's single figure was dominated by a single-threaded numpy exp over N^2 entries
        import scipy.linalg as sl

        try:
            from threadpoolctl import threadpool_info

            blas = [{"api": i.get("internal_api"), "threads": i.get("num_threads")} for i in threadpool_info() if i.get("user_api") == "blas"]
        except Exception:  # noqa: BLE001
            blas = None
        t_build = t_potrf = t_solve = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            sq = (X * X).sum(axis=1)
            Kc = np.exp(-0.5 * np.maximum(sq[:, None] + sq[None, :] - 2.0 * (X @ X.T), 0.0))
            Kc[np.diag_indices(N)] += 0.01 + 1e-8
            t1 = time.perf_counter()
            Lc = sl.cholesky(Kc, lower=True, overwrite_a=True, check_finite=False)
            t2 = time.perf_counter()
            ac = sl.cho_solve((Lc, True), om, check_finite=False)
            ll_all = -0.5 * float((om * ac).sum()) - float(np.log(np.diag(Lc)).sum()) - 0.5 * N * np.log(2 * np.pi)
            t3 = time.perf_counter()
            t_build, t_potrf, t_solve = min(t_build, t1 - t0), min(t_potrf, t2 - t1), min(t_solve, t3 - t2)
        out["cpu_lapack"] = {"dpotrf_s": t_potrf, "dpotrf_gflops": float(N) ** 3 / 3.0 / t_potrf / 1e9, "dpotrs_and_loglik_s": t_solve,
                             "numpy_kernel_build_s": t_build, "blas_threads": blas, "host_cores": os.cpu_count(), "log_lik": ll_all,
                             "evaluations_per_s_factor_and_solve_only": 1.0 / (t_potrf + t_solve),
                             "note": f"best of 3 at N={N}: scipy/OpenBLAS dpotrf + dpotrs with the BLAS thread count shown; the kernel-matrix "
                                     "build is vectorised numpy (one thread: exp over N^2 entries) and listed apart — not the reference's path"}
        if "config5" in out:  # the same BO inner loop / BO iteration through the oracle on one core of this box
            out["config5"]["cpu_oracle_1_core"] = dict(bo_inner_loop(orc, _capi, O, 0), bo_iteration=bo_iteration(orc, _capi, O, 0))
        if "config4_g64" in out:
            # B4: configs[3] by the reference's own strategy — one GP = one host task (tools::par::loop,
            # multi_gp.hpp:124-126), 64 tasks of the single-threaded port on the host cores
            from concurrent.futures import ThreadPoolExecutor

            N4 = 2048
            X4, Y4 = O.make_problem("c4", N=N4)
            rng4 = np.random.default_rng(4)
            hos = []
            for g_ in range(64):
                om4, _ = O.obs_mean_data(Y4 * rng4.uniform(0.5, 1.5) + 0.1 * np.sin(3.0 * X4[:, g_ % 6: g_ % 6 + 1] + g_))
                ho4 = _capi.Handle(orc)
                ho4.set_kernel(O.SE_ARD, rng4.uniform(-1e-2, 1e-2, size=D_C2 + 1), 0.01)
                ho4.set_data(X4, om4)
                hos.append(ho4)
            nthr = min(64, os.cpu_count() or 1)

            def task(ho4):
                ho4.compute()
                return ho4.log_lik()

            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=nthr) as ex:
                lls = list(ex.map(task, hos))
            t64 = time.perf_counter() - t0
            for ho4 in hos:
                ho4.close()
            out["config4_g64"]["cpu_oracle_64_host_tasks"] = {"value": 64 / t64, "unit": "evaluations/s", "threads": nthr, "host_cores": os.cpu_count(),
                                                              "wall_s": t64, "kind": "port", "log_lik_0": lls[0]}
        out["cpu_baseline"] = {"value": n_cpu / tc, "unit": "evaluations/s", "cores": 1, "kind": "port",
                               "sample": f"{n_cpu} full compute()+log_lik at N={N}, D={D_C2} (oracle/gp_oracle.c, "
                                         f"gcc -O3, {os.cpu_count()} host cores present, 1 used)",
                               "log_lik": llo, "rel_diff_vs_gpu": abs(llo - ll) / abs(llo)}
    h.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
