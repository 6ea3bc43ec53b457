"""Handles that go through a batched launch sequence (tall + update + closing) and then compute() one by one / from threads."""
import os, sys, time, threading
sys.path.insert(0, "/root/repo")
import numpy as np
from limbo_amd import _capi, synth as O
eng = _capi.load_engine()
G, N = 8, 2048
X, Y = O.make_problem("c4", N=N)
rng = np.random.default_rng(4)
hs = []
for g in range(G):
    om, _ = O.obs_mean_data(Y * rng.uniform(0.5, 1.5))
    h = _capi.Handle(eng, 0); h.set_data(X, om); h.set_kernel(O.SE_ARD, rng.uniform(-1e-2, 1e-2, size=7), 0.01); hs.append(h)
for r in range(2):
    t0 = time.perf_counter(); st = _capi.batch_compute(hs); ll = _capi.batch_log_lik(hs); dt = time.perf_counter() - t0
    print(f"batch {r}: {1e3*dt:.3f} ms  ll[0] {ll[0]:.12g} ll[7] {ll[7]:.12g} reruns {[h.handover_reruns() for h in hs]}", flush=True)
for r in range(2):
    for g, h in enumerate(hs):
        t0 = time.perf_counter(); info = h.compute(); l = h.log_lik(); dt = time.perf_counter() - t0
        print(f"single pass {r} handle {g}: {1e3*dt:8.3f} ms info {info} ll {l:.12g} (batch {ll[g]:.12g}) reruns {h.handover_reruns()} retries {h.flow_retries()}", flush=True)
def worker(h):
    for _ in range(6):
        h.compute(); h.log_lik()
ths = [threading.Thread(target=worker, args=(h,)) for h in hs]
t0 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]; dt = (time.perf_counter() - t0) / 6
print(f"8 threads: {1e3*dt:.3f} ms per round  reruns {[h.handover_reruns() for h in hs]} retries {[h.flow_retries() for h in hs]}", flush=True)
t0 = time.perf_counter(); st = _capi.batch_compute(hs); ll2 = _capi.batch_log_lik(hs); dt = time.perf_counter() - t0
print(f"batch again: {1e3*dt:.3f} ms  max rel diff {np.max(np.abs((ll2-ll)/ll)):.2e} reruns {[h.handover_reruns() for h in hs]}", flush=True)
