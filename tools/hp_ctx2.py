"""Does torch's HIP context (its streams / hardware queues) change hp_objective?  bench.py reads 3.2 ms where tools/hp_try.py reads 2.87."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
if os.environ.get("WITH_TORCH", "1") == "1":
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
    if os.environ.get("TORCH_WORK", "0") == "1":
        a = torch.ones(1024, device="cuda"); (a + 1).sum().item()
from limbo_amd import _capi, synth
eng = _capi.load_engine()
X, Y = synth.make_problem("c2", N=4096)
om, _ = synth.obs_mean_data(Y)
th = np.zeros(7)
for rep in range(2):
    h = _capi.Handle(eng); h.set_data(X, om)
    h.hp_objective(0, th, 0.01, optimize_noise=False, want_grad=True)
    per = []
    for i in range(25):
        t0 = time.perf_counter()
        h.hp_objective(0, th + 1e-3 * (i + 1), 0.01, optimize_noise=False, want_grad=True)
        per.append(time.perf_counter() - t0)
    print(f"WITH_TORCH={os.environ.get('WITH_TORCH','1')} TORCH_WORK={os.environ.get('TORCH_WORK','0')} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES','-')}: hp_objective mean {1e3 * np.mean(per):.3f} ms median {1e3 * np.median(per):.3f} ms")
    h.close()
