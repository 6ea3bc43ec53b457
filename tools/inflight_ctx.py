"""bench.py's concurrency section read 12 evaluations/s where tools/inflight_try.py reads 1300: which part of its context does it?
    python tools/inflight_ctx.py <letters>   t: torch imported and synchronised, m: a main handle alive (20 evaluations),
                                             p: profiled evaluations on it, h: a second handle under GPE_TALL=0 GPE_TAIL_MAX=0"""
import os, sys, time, threading
sys.path.insert(0, "/root/repo")
import numpy as np
V = sys.argv[1]
if "t" in V:
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
from limbo_amd import _capi, synth as O
eng = _capi.load_engine()
X, Y = O.make_problem("c2", N=4096)
om, _ = O.obs_mean_data(Y)
th = np.zeros(7)
if "m" in V:
    hm = _capi.Handle(eng, 0); hm.set_kernel(O.SE_ARD, th, 0.01); hm.set_data(X, om)
    for _ in range(20):
        hm.compute(); hm.log_lik()
if "p" in V:
    hm.compute(); hm.set_profiling(True); hm.reset_phase_ms()
    for _ in range(5):
        hm.compute(); hm.log_lik()
    hm.get_phase_ms(); hm.set_profiling(False)
if "h" in V:
    os.environ["GPE_TALL"], os.environ["GPE_TAIL_MAX"] = "0", "0"
    h15 = _capi.Handle(eng, 0)
    os.environ.pop("GPE_TALL"), os.environ.pop("GPE_TAIL_MAX")
    h15.set_kernel(O.SE_ARD, th, 0.01); h15.set_data(X, om); h15.compute()
    h15.set_profiling(True)
    for _ in range(3):
        h15.compute(); h15.log_lik()
    h15.set_profiling(False); h15.close()
for R in (4,):
    hs = []
    for r in range(R):
        h = _capi.Handle(eng, 0); h.set_kernel(O.SE_ARD, th + 1e-3 * r, 0.01); h.set_data(X, om); hs.append(h)
    for h in hs:
        h.compute()
    if "t" in V:
        torch.cuda.synchronize()
    per = 6
    def worker(h):
        for _ in range(per):
            h.compute(); h.log_lik()
    ths = [threading.Thread(target=worker, args=(h,)) for h in hs]
    t0 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]; dt = time.perf_counter() - t0
    print(f"variant {V:5s} in flight {R}: {R * per / dt:8.1f} evaluations/s  reruns {[h.handover_reruns() for h in hs]} retries {[h.flow_retries() for h in hs]}", flush=True)
