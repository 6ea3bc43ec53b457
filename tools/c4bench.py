#!/usr/bin/env python
"""Config 4 on one GPU: G independent GPs of N = 2048, D = 6 through gpe_batch_compute (+ log_lik each).
Round 1 (every GP its own launch chain on its own streams): G = 8: 1900 evaluations/s, G = 64: 1880 whatever the
host threading / queue count.  Round 2 (one launch sequence for all G, gridDim.z = GP): G = 8: 4100, G = 64: 7200
(GPE_BATCH=0 restores the per-GP chains).  usage: c4bench.py [G ...]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from limbo_amd import _capi  # noqa: E402
from limbo_amd import synth as O  # problem generator (pure numpy)

eng = _capi.load_engine()
X4, Y4 = O.make_problem("c2", N=2048)
om4, _ = O.obs_mean_data(Y4)
for G in ([int(a) for a in sys.argv[1:]] or [8, 64]):
    hs = []
    for g_ in range(G):
        hh = _capi.Handle(eng, 0)
        hh.set_data(X4, om4)
        hh.set_kernel(O.SE_ARD, np.zeros(7) + 1e-2 * g_, 0.01)
        hs.append(hh)
    _capi.batch_compute(hs)
    reps = 5 if G == 8 else 2
    t0 = time.perf_counter()
    for _ in range(reps):
        _capi.batch_compute(hs)
        lls = [x.log_lik() for x in hs]
    dt = time.perf_counter() - t0
    print(f"G={G}: {G * reps / dt:.0f} evaluations/s  (log_lik[0] = {lls[0]:.9f}, [-1] = {lls[-1]:.9f})", flush=True)
    for hh in hs:
        hh.close()
