import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from limbo_amd import _capi, synth
eng = _capi.load_engine()
for N in (2048, 4096):
    X, Y = synth.make_problem("c2", N=N)
    om, _ = synth.obs_mean_data(Y)
    h = _capi.Handle(eng); h.set_data(X, om); h.set_kernel(0, np.zeros(7), 0.01)
    rc = h.compute(); ll = h.log_lik()
    L = np.tril(h.get_L())
    print(N, rc, "%.17g" % ll, "sum|L| %.17g" % np.abs(L).sum(), "L[-1,-1] %.17g" % L[-1, -1], "L[-1,-70] %.17g" % L[-1, -70])
    h.close()
