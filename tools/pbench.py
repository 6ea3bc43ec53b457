#!/usr/bin/env python
"""compute()+log_lik and update_alpha+log_lik at N = 4096, D = 6 for P = 1, 2, 4, 8 outputs (one GP, obs N x P)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from limbo_amd import _capi  # noqa: E402
from limbo_amd import synth as O  # problem generator (pure numpy)

eng = _capi.load_engine()
X, Y = O.make_problem("c2", N=4096)
for P in (1, 2, 4, 8):
    Yp = np.concatenate([Y * (p + 1) for p in range(P)], axis=1)
    om, _ = O.obs_mean_data(Yp)
    h = _capi.Handle(eng, 0)
    h.set_kernel(O.SE_ARD, np.zeros(7), 0.01)
    h.set_data(X, om)
    for _ in range(3):
        h.compute(); h.log_lik()
    t0 = time.perf_counter()
    for _ in range(20):
        h.compute(); h.log_lik()
    t1 = time.perf_counter()
    for _ in range(20):
        h.update_alpha(om); h.log_lik()
    t2 = time.perf_counter()
    print(f"P={P}: compute+log_lik {1e3 * (t1 - t0) / 20:.3f} ms, update_alpha+log_lik {1e6 * (t2 - t1) / 20:.0f} us", flush=True)
    h.close()
