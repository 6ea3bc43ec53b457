"""compute()+log_lik by size under GPE_TAIL_MAX = 2560 / 2816 (read per handle)."""
import os, sys
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tools")
import numpy as np
sys.argv = [sys.argv[0], "none"]
import r4_ab as R
from limbo_amd import synth as O
for N in (2048, 2624, 2816, 3072, 3584, 4096, 4160, 5000, 8192):
    X, Y = O.make_problem("c2", N=N)
    om, _ = O.obs_mean_data(Y)
    row = []
    for tm in (2560, 2816):
        h = R.handle(X, om, O.SE_ARD, np.zeros(7), None, tm)
        med, mn, ll = R.timed(h, steps=14, warm=3)
        row.append((med, ll)); h.close()
    print(f"N {N:5d}: tail_max 2560 {row[0][0]:.3f} ms | 2816 {row[1][0]:.3f} ms   rel diff {abs(row[0][1]-row[1][1])/abs(row[0][1]):.1e}", flush=True)
