import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from limbo_amd import _capi, synth
eng = _capi.load_engine()
X, Y = synth.make_problem("c2", N=4096)
om, _ = synth.obs_mean_data(Y)
h = _capi.Handle(eng); h.set_data(X, om)
th = np.zeros(7)
for _ in range(3):
    h.hp_objective(0, th, 0.01, optimize_noise=False, want_grad=True)
n = 20
t0 = time.perf_counter()
for i in range(n):
    r = h.hp_objective(0, th + 1e-3 * i, 0.01, optimize_noise=False, want_grad=True); l = r[0]
dt = (time.perf_counter() - t0) / n
print("hp_objective %.3f ms  lik %.12g" % (dt * 1e3, l))
t0 = time.perf_counter()
for i in range(n):
    h.set_kernel(0, th + 1e-3 * i, 0.01); h.compute(); h.log_lik()
print("compute %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
