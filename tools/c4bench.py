#!/usr/bin/env python
"""Config 4 on one GPU: G independent GPs of N = 2048, D = 6 through gpe_batch_compute (+ log_lik each).
Measured on MI355X: G = 8: 1900 evaluations/s, G = 64: 1880; enqueueing from 8 host threads instead of one
changed nothing (1910 vs 1897: not host-launch bound); GPU_MAX_HW_QUEUES=16 (default 4 hardware queues for
the 2 x G streams): 1963 / 2006."""
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
for G in (8, 64):
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
