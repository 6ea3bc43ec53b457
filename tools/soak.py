"""Soak: eight host threads on eight handles of mixed sizes (64 .. 16384 samples: small path, one-launch sweeps, restricted and
unrestricted look-ahead updates all in flight together), every result compared bitwise with the same handle's lone run.
usage: soak.py [multiplier]"""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
from limbo_amd import _capi, synth
eng = _capi.load_engine()
sizes = [8192, 64, 8192, 700, 4096, 1500, 16384, 3000]
rng = np.random.default_rng(1)
probs = []
for N in sizes:
    X = rng.uniform(0, 1, size=(N, 6)); Y = synth.hartmann6(X)[:, None] + 0.05 * rng.normal(size=(N, 1))
    om, _ = synth.obs_mean_data(Y)
    probs.append((X, om, rng.uniform(-0.2, 0.2, size=7), rng.uniform(0, 1, size=(8, 6))))
hs = []
ref = []
for (X, om, th, Xq) in probs:
    h = _capi.Handle(eng); h.set_data(X, om); h.set_kernel(0, th, 0.01); assert h.compute() == 0
    ref.append((h.log_lik(), h.query_batch(Xq[:1])[1][0]))
    hs.append(h)
bad = []
def work(i, iters):
    X, om, th, Xq = probs[i]
    h = hs[i]
    for it in range(iters):
        assert h.compute() == 0
        ll = h.log_lik(); v = h.query_batch(Xq[:1])[1][0]
        h.update_alpha(om)
        if ll != ref[i][0] or v != ref[i][1]:
            bad.append((i, it, ll, ref[i][0]))
t0 = time.time()
mult = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ths = [threading.Thread(target=work, args=(i, mult * max(3, int(60000 // sizes[i]) if sizes[i] > 4096 else 40))) for i in range(len(sizes))]
[t.start() for t in ths]; [t.join() for t in ths]
print("soak done in %.1f s, mismatches: %d, retries: %d" % (time.time() - t0, len(bad), sum(h.flow_retries() for h in hs)), bad[:3])
