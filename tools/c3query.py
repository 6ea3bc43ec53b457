#!/usr/bin/env python
"""Config 3 queries (N=16384, D=12, Matern-5/2): 100 k points through gpe_query_batch, five times, host to host; the phases of the
last call.  usage: c3query.py [M]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from limbo_amd import _capi, synth  # noqa: E402
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
eng = _capi.load_engine()
X, Y = synth.make_problem("c3", N=16384)
om, _ = synth.obs_mean_data(Y)
h = _capi.Handle(eng); h.set_data(X, om); h.set_kernel(synth.MATERN52, np.zeros(2), 0.01); h.compute()
Xq = np.random.default_rng(5).uniform(0, 1, size=(M, 12))
h.query_batch(Xq[:4096])
for r in range(5):
    t0 = time.perf_counter(); mu, var = h.query_batch(Xq); dt = time.perf_counter() - t0
    print(f"call {r}: {dt:.3f} s = {M / dt / 1e3:.0f} k points/s = {M * 16384.0 ** 2 * 2 / dt / 78.6e12:.3f} of peak", flush=True)
