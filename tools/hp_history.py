"""hp_objective after nine stream-creation histories (profiles/r04_stream_queue_mapping.log): with both streams of a handle at the default priority the runtime put them on one hardware queue in two of them (3.22 ms instead of 2.86)"""
import os, sys, time, threading
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
torch.cuda.set_device(0)
from limbo_amd import _capi, synth as O
eng = _capi.load_engine()
X, Y = O.make_problem("c2", N=4096)
om, _ = O.obs_mean_data(Y)
th = np.zeros(7)

def hp(tag):
    h = _capi.Handle(eng, 0); h.set_data(X, om)
    h.hp_objective(O.SE_ARD, th, 0.01, optimize_noise=False, want_grad=True)
    per = []
    for i in range(25):
        t0 = time.perf_counter()
        h.hp_objective(O.SE_ARD, th + 1e-3 * (i + 1), 0.01, optimize_noise=False, want_grad=True)
        per.append(time.perf_counter() - t0)
    h.set_profiling(True); h.reset_phase_ms()
    h.hp_objective(O.SE_ARD, th, 0.01, optimize_noise=False, want_grad=True)
    ph = h.get_phase_ms(); h.set_profiling(False)
    print(f"{tag}: hp_objective mean {1e3 * np.mean(per):.3f} ms median {1e3 * np.median(per):.3f}  phases sum {sum(v['ms'] for v in ph.values()):.3f}", flush=True)
    h.close()

hp("A fresh")
hm = _capi.Handle(eng, 0); hm.set_kernel(O.SE_ARD, th, 0.01); hm.set_data(X, om)
for _ in range(55):
    hm.compute(); hm.log_lik()
hp("B main handle alive, 55 steps")
for _ in range(10):
    hm.set_data(X, om); hm.compute(); hm.log_lik()
hp("B2 + set_data steps")
hs4 = []
X4, Y4 = O.make_problem("c4", N=2048); om4, _ = O.obs_mean_data(Y4)
for g in range(8):
    h4 = _capi.Handle(eng, 0); h4.set_kernel(O.SE_ARD, th, 0.01); h4.set_data(X4, om4); hs4.append(h4)
for _ in range(10):
    _capi.batch_compute(hs4); _capi.batch_log_lik(hs4)
for h4 in hs4:
    h4.close()
hp("C + config4 batch")
hm.compute(); hm.set_profiling(True); hm.reset_phase_ms()
for _ in range(5):
    hm.compute(); hm.log_lik()
hm.get_phase_ms(); hm.set_profiling(False)
hp("D + profiled main")
os.environ["GPE_TALL"], os.environ["GPE_TAIL_MAX"] = "0", "0"
h15 = _capi.Handle(eng, 0)
os.environ.pop("GPE_TALL"), os.environ.pop("GPE_TAIL_MAX")
h15.set_kernel(O.SE_ARD, th, 0.01); h15.set_data(X, om)
h15.compute(); h15.set_profiling(True)
for _ in range(3):
    h15.compute(); h15.log_lik()
h15.set_profiling(False); h15.close()
hp("E + h15 panels handle")
for R in (4, 8):
    hs = []
    for r in range(R):
        hr = _capi.Handle(eng, 0); hr.set_kernel(O.SE_ARD, th + 1e-3 * r, 0.01); hr.set_data(X, om); hs.append(hr)
    def worker(hr):
        for _ in range(25):
            hr.compute(); hr.log_lik()
    for hr in hs:
        hr.compute()
    ths = [threading.Thread(target=worker, args=(hr,)) for hr in hs]
    [t.start() for t in ths]; [t.join() for t in ths]
    for hr in hs:
        hr.close()
    hp(f"F + {R} threads in flight")
hm.close()
hp("G main handle closed")
