import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from limbo_amd import _capi, synth
eng = _capi.load_engine()
for N in (1100, 4096, 2048+64):
    X, Y = synth.make_problem("c2", N=N)
    om, _ = synth.obs_mean_data(Y)
    h = _capi.Handle(eng); h.set_data(X, om); h.set_kernel(0, np.zeros(7), 0.01)
    rc = h.compute(); ll = h.log_lik()
    t0 = time.perf_counter()
    for _ in range(20):
        h.compute(); h.log_lik()
    dt = (time.perf_counter() - t0) / 20
    print(N, rc, "%.17g" % ll, "retries", h.flow_retries(), "ms %.3f" % (dt * 1e3), flush=True)
    h.close()
