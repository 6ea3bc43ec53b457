"""R handles evaluated from R host threads at once (bench.py's concurrent_evaluations_per_s): time, re-runs."""
import os, sys, time, threading
sys.path.insert(0, "/root/repo")
import numpy as np
from limbo_amd import _capi, synth as O
eng = _capi.load_engine()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X, Y = O.make_problem("c2", N=N)
om, _ = O.obs_mean_data(Y)
for R in (2, 4, 8):
    hs = []
    for r in range(R):
        h = _capi.Handle(eng, 0); h.set_kernel(O.SE_ARD, np.zeros(7) + 1e-3 * r, 0.01); h.set_data(X, om); hs.append(h)
    for h in hs:
        h.compute()
    per = 8
    def worker(h):
        for _ in range(per):
            h.compute(); h.log_lik()
    ths = [threading.Thread(target=worker, args=(h,)) for h in hs]
    t0 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]; dt = time.perf_counter() - t0
    print(f"N {N} in flight {R}: {R * per / dt:8.1f} evaluations/s  reruns {[h.handover_reruns() for h in hs]} retries {[h.flow_retries() for h in hs]}", flush=True)
    for h in hs:
        h.close()
