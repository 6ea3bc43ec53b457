#!/usr/bin/env python
"""rocprofv3 --kernel-trace helper for the K^-1 segment (engine.hip:ensure_inv).
   run:   rocprofv3 --kernel-trace -d /tmp/p_inv -o p -- python tools/inv_trace.py run
   dump:  python tools/inv_trace.py dump <rocpd .db>     (kernels of the last inversion in time order)"""
import sys
from pathlib import Path

if sys.argv[1] == "run":
    import numpy as np
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from limbo_amd import _capi
    from limbo_amd import synth as O  # problem generator (pure numpy)
    eng = _capi.load_engine()
    X, Y = O.make_problem("c2", N=4096)
    om, _ = O.obs_mean_data(Y)
    h = _capi.Handle(eng, 0)
    h.set_kernel(O.SE_ARD, np.zeros(7), 0.01)
    h.set_data(X, om)
    for i in range(3):
        h.set_kernel(O.SE_ARD, np.zeros(7) + 1e-3 * i, 0.01)
        h.compute(); h.log_lik(); h.log_lik_grad(False)
    h.close()
else:
    import sqlite3
    c = sqlite3.connect(sys.argv[2])
    rows = c.execute("select name, grid_x, start, end from kernels order by start").fetchall()
    last = max(i for i, r in enumerate(rows) if "k_inv_panels" in r[0] or "k_set_identity" in r[0])
    t0 = rows[last][2]
    print(f"{'kernel':70s} {'grid_x':>8s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s}")
    prev_end = t0
    for r in rows[last:]:
        print(f"{r[0][:70]:70s} {r[1]:8d} {(r[2] - t0) / 1e3:9.1f} {(r[3] - r[2]) / 1e3:8.1f} {(r[2] - prev_end) / 1e3:7.1f}")
        prev_end = r[3]
