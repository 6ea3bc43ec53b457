#!/usr/bin/env python
"""Per-point query latency (one point, mu and sigma^2, host to host) at a few N.  GPE_QUERY_SWEEP=0 selects the
blocked matrix solve for single points too (A/B of engine.hip:query_impl's small-batch path)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from limbo_amd import _capi  # noqa: E402
from limbo_amd import synth as O  # problem generator (pure numpy)

eng = _capi.load_engine()
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
    t0 = time.perf_counter()
    for i in range(20, 220):
        hq.query_batch(pts[i:i + 1])
    t1 = time.perf_counter()
    k8, v8 = hq.query_batch(pts[:8])
    t2 = time.perf_counter()
    for i in range(20):
        hq.query_batch(pts[:8])
    t3 = time.perf_counter()
    hq.update_alpha(omn)
    t4 = time.perf_counter()
    for i in range(20):
        hq.update_alpha(omn)
        hq.log_lik()
    t5 = time.perf_counter()
    print(f"N={n_}: update_alpha+log_lik {1e6 * (t5 - t4) / 20:.1f} us", flush=True)
    print(f"N={n_}: single point {1e6 * (t1 - t0) / 200:.1f} us, 8 points {1e6 * (t3 - t2) / 20:.1f} us", flush=True)
    hq.close()
