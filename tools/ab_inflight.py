#!/usr/bin/env python
"""A/B of two builds of libgpengine.so on the same box: one evaluation at a time and R evaluations in flight
(usage: ab_inflight.py <old.so> ; the in-tree library is B).  Only the symbols both builds export are used."""
import ctypes as C
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from limbo_amd import synth as O  # noqa: E402

dp = C.POINTER(C.c_double)


def bench(path):
    lib = C.CDLL(str(Path(path).resolve()))
    X, Y = O.make_problem("c2", N=4096)
    om, _ = O.obs_mean_data(Y)
    X = np.ascontiguousarray(X)
    om = np.asfortranarray(om)

    def mk(i):
        h = C.c_void_p()
        assert lib.gpe_create(0, C.byref(h)) == 0
        th = np.zeros(7) + 1e-3 * i
        lib.gpe_set_kernel(h, 0, th.ctypes.data_as(dp), 7, C.c_double(0.01))
        lib.gpe_set_data(h, X.ctypes.data_as(dp), C.c_int64(4096), 6, om.ctypes.data_as(dp), 1)
        return h

    def step(h):
        lib.gpe_compute(h)
        ll = C.c_double()
        lib.gpe_log_lik(h, C.byref(ll))
        return ll.value

    h = mk(0)
    for _ in range(5):
        step(h)
    t0 = time.perf_counter()
    for _ in range(50):
        step(h)
    one = 50 / (time.perf_counter() - t0)
    res = {"one": one}
    for R in (4, 8):
        hs = [mk(r) for r in range(R)]
        for q in hs:
            step(q)
        per = 25
        ths = [threading.Thread(target=lambda q=q: [step(q) for _ in range(per)]) for q in hs]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        res[f"in_flight_{R}"] = R * per / (time.perf_counter() - t0)
        for q in hs:
            lib.gpe_destroy(q)
    lib.gpe_destroy(h)
    return res


if __name__ == "__main__":
    root = Path(__file__).resolve().parent.parent
    for name, p in (("A " + sys.argv[1], sys.argv[1]), ("B in-tree", str(root / "limbo_amd" / "libgpengine.so")), ("A again", sys.argv[1])):
        print(name, {k: round(v, 1) for k, v in bench(p).items()}, flush=True)
