#!/usr/bin/env python
"""K^-1 by recursion (csrc/inv2.hip) against the panel form it replaces and against LAPACK.

  python tools/inv2_try.py check            K^-1 and the gradient at N = 1024, 2048, 4096 vs LAPACK (both paths)
  python tools/inv2_try.py time [N]         gpe_hp_objective (gradient) wall time + the `inv` phase, both paths
The path is chosen per PROCESS (GPE_INV2, GPE_INV2_BINS, GPE_INV2_LOAD are read once): this script re-runs itself.
"""
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def child(mode, N):
    import scipy.linalg as sla

    from limbo_amd import _capi
    from limbo_amd import synth as O
    from oracle import np_oracle as NO

    eng = _capi.load_engine()
    X, Y = O.make_problem("c2", N=N)
    om, _ = O.obs_mean_data(Y)
    h = _capi.Handle(eng, 0)
    h.set_data(X, om)
    rng = np.random.default_rng(N)
    th = rng.uniform(-0.3, 0.3, size=7)
    h.set_kernel(O.SE_ARD, th, 0.01)
    tag = f"GPE_INV2={os.environ.get('GPE_INV2', '1')} BINS={os.environ.get('GPE_INV2_BINS', '-')} LOAD={os.environ.get('GPE_INV2_LOAD', '-')}"
    if mode == "check":
        assert h.compute() == 0
        g = h.log_lik_grad(False)
        Kinv = h.get_Kinv()
        K = NO.kernel_matrix(NO.SE_ARD, X, th, 0.01)
        L = sla.cholesky(K, lower=True)
        Kr = sla.cho_solve((L, True), np.eye(N))
        e = np.linalg.norm(Kinv - Kr) / np.linalg.norm(Kr)
        g2 = h.log_lik_grad(False)
        # a second evaluation on the same handle (resident plan) at another theta
        th2 = th + 0.1
        ll2, gg2, info = h.hp_objective(O.SE_ARD, th2, 0.01, optimize_noise=False, want_grad=True)
        K2 = NO.kernel_matrix(NO.SE_ARD, X, th2, 0.01)
        L2 = sla.cholesky(K2, lower=True)
        Kr2 = sla.cho_solve((L2, True), np.eye(N))
        Kinv2 = h.get_Kinv()
        e2 = np.linalg.norm(Kinv2 - Kr2) / np.linalg.norm(Kr2)
        # ... and a smaller, ragged sample set on the same handle (the pads of the recursion's buffers are zero-filled again)
        Ns = N - 37
        h.set_data(X[:Ns], om[:Ns])
        h.hp_objective(O.SE_ARD, th2, 0.01, optimize_noise=False, want_grad=True)
        Kr3 = sla.cho_solve((sla.cholesky(K2[:Ns, :Ns], lower=True), True), np.eye(Ns))
        e3 = np.linalg.norm(h.get_Kinv() - Kr3) / np.linalg.norm(Kr3)
        print(f"{tag} N={N}: K^-1 rel err {e:.2e} (second theta {e2:.2e}; shrunk to {Ns}: {e3:.2e}), symmetric {np.max(np.abs(Kinv - Kinv.T)) == 0.0}, "
              f"grad repeat bitwise {np.array_equal(g, g2)}, nan {np.isnan(Kinv).any()}", flush=True)
    elif mode == "batch":
        G = 10
        hs = [h]
        for _ in range(G - 1):
            q = _capi.Handle(eng, 0)
            q.set_data(X, om)
            hs.append(q)
        ths = th[None, :] + 1e-2 * rng.uniform(-1, 1, size=(G, 7))
        nz = np.full(G, 0.01)
        for _ in range(2):
            lik, grad, st = _capi.batch_hp_objective(hs, O.SE_ARD, ths, nz, optimize_noise=False, want_grad=True)
        t0 = time.perf_counter()
        n = 5
        for i in range(n):
            lik, grad, st = _capi.batch_hp_objective(hs, O.SE_ARD, ths + 1e-4 * i, nz, optimize_noise=False, want_grad=True)
        wall = (time.perf_counter() - t0) / n
        lik, grad, st = _capi.batch_hp_objective(hs, O.SE_ARD, ths, nz, optimize_noise=False, want_grad=True)
        one = _capi.Handle(eng, 0)
        one.set_data(X, om)
        worst = 0.0
        for q in (0, 3, 9):
            l1, g1, _ = one.hp_objective(O.SE_ARD, ths[q], 0.01, optimize_noise=False, want_grad=True)
            worst = max(worst, abs(l1 - lik[q]) / abs(l1), float(np.linalg.norm(g1 - grad[q]) / np.linalg.norm(g1)))
        print(f"{tag} N={N}: batch of {G}: {1e3 * wall:.2f} ms = {G / wall:.0f} objective evaluations/s; members vs single handle {worst:.1e}; status {st}", flush=True)
    else:
        for _ in range(3):
            h.hp_objective(O.SE_ARD, th, 0.01, optimize_noise=False, want_grad=True)
        t0 = time.perf_counter()
        n = 20
        for i in range(n):
            h.hp_objective(O.SE_ARD, th + 1e-3 * i, 0.01, optimize_noise=False, want_grad=True)
        wall = (time.perf_counter() - t0) / n
        h.set_profiling(True)
        h.reset_phase_ms()
        for i in range(3):
            h.hp_objective(O.SE_ARD, th + 1e-3 * i, 0.01, optimize_noise=False, want_grad=True)
        ph = h.get_phase_ms()
        h.set_profiling(False)
        inv = ph.get("inv", {"ms": 0, "flops": 0, "launches": 0})
        fl = 2.0 * N ** 3 / 3.0
        print(f"{tag} N={N}: hp_objective {1e3 * wall:.3f} ms; inv phase (serialised by profiling) {inv['ms'] / 3:.3f} ms "
              f"= {fl / (inv['ms'] / 3 * 1e-3) / 1e12 / 78.6:.3f} of peak for 2N^3/3", flush=True)
        if os.environ.get("INV2_TRACE"):
            eng.fn("trace")(1)
            h.hp_objective(O.SE_ARD, th, 0.01, optimize_noise=False, want_grad=True)
            eng.fn("trace_dump")(os.environ["INV2_TRACE"].encode())
            eng.fn("trace")(0)
    h.close()


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
        return
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    sizes = [int(a) for a in sys.argv[2:]] or ([1024, 2048, 4096] if mode == "check" else [4096])
    variants = [{"GPE_INV2": "0"}, {"GPE_INV2": "1"}]
    if mode == "time":
        variants += [{"GPE_INV2": "1", "GPE_HP_FUSED": "0"}]
    for N in sizes:
        for v in variants:
            env = dict(os.environ, **v)
            if mode == "time" and v == {"GPE_INV2": "1"}:
                env["INV2_TRACE"] = str(ROOT / "gpurun_out" / f"r05_inv2_trace_n{N}.txt")
            r = subprocess.run([sys.executable, __file__, "--child", mode, str(N)], env=env, capture_output=True, text=True, timeout=600)
            print(r.stdout.strip() or f"{v} N={N}: no output; rc {r.returncode}; {r.stderr[-800:]}", flush=True)


if __name__ == "__main__":
    main()
