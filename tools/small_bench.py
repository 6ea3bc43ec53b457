#!/usr/bin/env python
"""Config 5 regime (src/benchmarks/limbo/bench.cpp:66-84): add_sample() 10 -> 200 samples and the one-point query of an
acquisition functor at N = 200, host to host through the C-ABI — the one-launch small path (csrc/small.hip), the
general multi-launch path (GPE_SMALL=0) and, beside them, the CPU oracle on one core of the same box."""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from limbo_amd import _capi, synth  # noqa: E402
from oracle import binding as OB  # noqa: E402  (the CPU side-by-side figure)


def run(lib, noise=0.01, reps=5):
    n0, n1, D = 10, 200, 6
    rng = np.random.default_rng(5)
    X = rng.uniform(0, 1, size=(n1, D))
    Y = synth.hartmann6(X)[:, None]
    oms = [np.asfortranarray(synth.obs_mean_data(Y[: n + 1])[0]) for n in range(n1)]
    xs = [np.ascontiguousarray(X[n]) for n in range(n1)]
    best_add = 1e30
    for _ in range(reps):
        h = _capi.Handle(lib)
        h.set_kernel(0, np.zeros(D + 1), noise)
        h.set_data(X[:n0], oms[n0 - 1])
        h.compute()
        t0 = time.perf_counter()
        for n in range(n0, n1):
            h.add_sample(xs[n], oms[n])
        best_add = min(best_add, time.perf_counter() - t0)
        if _ + 1 < reps:
            h.close()
    pts = [np.ascontiguousarray(p[None, :]) for p in rng.uniform(0, 1, size=(400, D))]
    for p in pts[:50]:
        h.query_batch(p)
    best_q = 1e30
    for i0 in (50, 150, 250):
        t0 = time.perf_counter()
        for p in pts[i0 : i0 + 100]:
            h.query_batch(p)
        best_q = min(best_q, time.perf_counter() - t0)
    ll = h.log_lik()
    small = h.small_calls() if lib.prefix == "gpe_" else None
    h.close()
    return {"add_sample_per_s": (n1 - n0) / best_add, "add_sample_us": 1e6 * best_add / (n1 - n0),
            "one_point_query_us": 1e6 * best_q / 100, "log_lik": ll, "small_calls": small}


def main():
    eng = _capi.load_engine()
    out = {}
    out["gpu_small_path"] = run(eng)
    os.environ["GPE_SMALL"] = "0"  # read when a handle is created
    out["gpu_general_path"] = run(eng)
    del os.environ["GPE_SMALL"]
    out["cpu_oracle_1_core"] = run(OB.load_oracle())
    out["note"] = ("n = 10 -> 200, D = 6, P = 1, SE-ARD, noise 0.01; Python ctypes caller (its per-call overhead, ~2-3 us, "
                   "is inside every figure); best of 5 loops / best of 3 blocks of 100 queries")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
