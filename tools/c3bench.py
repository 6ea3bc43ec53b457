#!/usr/bin/env python
"""Config 3 factorisation (N=16384, D=12, Matern-5/2): compute()+log_lik time for the switches given in the environment
(e.g. GPE_BULK_WGS = workgroups of a look-ahead bulk update; 0 = unlimited).  usage: c3bench.py [N]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from limbo_amd import _capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
eng = _capi.load_engine()
X, Y = synth.make_problem("c3", N=N)
om, _ = synth.obs_mean_data(Y)
h = _capi.Handle(eng)
h.set_data(X, om)
h.set_kernel(synth.MATERN52, np.zeros(2), 0.01)
h.compute()
best = 1e30
for _ in range(3):
    t0 = time.perf_counter()
    info = h.compute()
    ll = h.log_lik()
    best = min(best, time.perf_counter() - t0)
fl = N ** 3 / 3.0 + 2.0 * N * N
print(f"N={N}: compute+log_lik {1e3 * best:.2f} ms = {fl / best / 1e12:.1f} TFLOP/s = {fl / best / 78.6e12:.3f} of fp64 peak (info {info}, log_lik {ll:.6f})")
