import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from limbo_amd import _capi, synth
eng = _capi.load_engine()
sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 1088, 2048, 4096, 1100]
for N in sizes:
    X, Y = synth.make_problem("c2", N=N)
    om, _ = synth.obs_mean_data(Y)
    h = _capi.Handle(eng); h.set_data(X, om); h.set_kernel(0, np.zeros(7), 0.01)
    rc = h.compute(); ll = h.log_lik()
    n = 20 if N <= 4096 else 5
    t0 = time.perf_counter()
    for _ in range(n):
        h.compute(); h.log_lik()
    dt = (time.perf_counter() - t0) / n
    print(N, rc, "%.17g" % ll, "retries", h.flow_retries(), "ms %.3f" % (dt * 1e3), flush=True)
    h.close()
