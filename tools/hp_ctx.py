"""hp_objective at N = 4096 on a fresh handle, before and after the allocation churn bench.py does in front of its `extras`
(eight handles of N = 4096 created, used and closed): does the placement of the buffers move the figure?"""
import sys, time, threading
sys.path.insert(0, "/root/repo")
import numpy as np
from limbo_amd import _capi, synth
eng = _capi.load_engine()
X, Y = synth.make_problem("c2", N=4096)
om, _ = synth.obs_mean_data(Y)
th = np.zeros(7)


def hp(tag):
    h = _capi.Handle(eng); h.set_data(X, om)
    h.hp_objective(0, th, 0.01, optimize_noise=False, want_grad=True)
    per = []
    for i in range(25):
        t0 = time.perf_counter()
        h.hp_objective(0, th + 1e-3 * (i + 1), 0.01, optimize_noise=False, want_grad=True)
        per.append(time.perf_counter() - t0)
    print(f"{tag}: hp_objective mean {1e3 * np.mean(per):.3f} ms  median {1e3 * np.median(per):.3f} ms")
    h.close()


hp("fresh process")
hs = []
for r in range(8):
    h = _capi.Handle(eng); h.set_kernel(0, th + 1e-3 * r, 0.01); h.set_data(X, om); hs.append(h)
ts = [threading.Thread(target=lambda hh=hh: [hh.compute() for _ in range(5)]) for hh in hs]
[t.start() for t in ts]; [t.join() for t in ts]
for h in hs:
    h.close()
hp("after 8 handles were created, used and closed")
hp("once more")
