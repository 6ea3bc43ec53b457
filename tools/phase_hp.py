#!/usr/bin/env python
"""Phase breakdown (HIP events on the handle's stream, profiling mode) of the HP objectives at N=4096, D=6:
compute + log_lik + grad (kernel_lf_opt.hpp:77-92) and compute + LOO + its gradient (kernel_loo_opt.hpp:77-95)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from limbo_amd import _capi  # noqa: E402
from limbo_amd import synth as O  # problem generator (pure numpy)

eng = _capi.load_engine()
X, Y = O.make_problem("c2", N=4096)
om, _ = O.obs_mean_data(Y)
h = _capi.Handle(eng, 0)
h.set_kernel(O.SE_ARD, np.zeros(7), 0.01)
h.set_data(X, om)
for name, fn in (("lf", lambda: (h.compute(), h.log_lik(), h.log_lik_grad(False))),
                 ("loo", lambda: (h.compute(), h.log_loo_cv(), h.log_loo_cv_grad(False)))):
    fn()
    t0 = time.perf_counter()
    for i in range(10):
        h.set_kernel(O.SE_ARD, np.zeros(7) + 1e-3 * i, 0.01)
        fn()
    wall = (time.perf_counter() - t0) / 10
    h.set_profiling(True)
    h.reset_phase_ms()
    for i in range(3):
        h.set_kernel(O.SE_ARD, np.zeros(7) + 1e-3 * i, 0.01)
        fn()
    ph = h.get_phase_ms()
    h.set_profiling(False)
    print(f"{name}: {1e3 * wall:.2f} ms wall per objective;  phases (ms, launches, TFLOP/s) per objective, each launch timed alone:")
    for k, v in ph.items():
        if v["launches"]:
            tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
            print(f"   {k:16s} {v['ms'] / 3:8.3f} {v['launches'] / 3:6.0f} {tf:7.1f}")
h.close()
