#!/usr/bin/env python
"""R handles evaluated from R host threads (compute + log_lik, N = 4096): evaluations/s with the CU-masked chain partitions of
round 5 (csrc/engine.hip: ChainScope) and without (GPE_FLOW_PARTITIONS=0: chain behind chain, round 4), re-runs, and every
log-likelihood against the handle's own sequential one.   python tools/conc_try.py [per-thread evaluations]"""
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def child(per):
    from limbo_amd import _capi
    from limbo_amd import synth as O

    if os.environ.get("CONC_TORCH"):
        import torch

        torch.zeros(4, device="cuda:0")
        torch.cuda.synchronize()
    eng = _capi.load_engine()
    X, Y = O.make_problem("c2", N=4096)
    om, _ = O.obs_mean_data(Y)
    tag = f"GPE_FLOW_PARTITIONS={os.environ.get('GPE_FLOW_PARTITIONS', '1')}"
    one = _capi.Handle(eng, 0)
    one.set_kernel(O.SE_ARD, np.zeros(7), 0.01)
    one.set_data(X, om)
    for _ in range(3):
        one.compute()
    t0 = time.perf_counter()
    for _ in range(per):
        one.compute()
        one.log_lik()
    single = per / (time.perf_counter() - t0)
    one.close()
    line = [f"{tag}: one handle {single:.0f}/s"]
    for R in [int(v) for v in os.environ.get("CONC_R", "2,4,8").split(",")]:
        hs, ref = [], []
        for r in range(R):
            h = _capi.Handle(eng, 0)
            h.set_kernel(O.SE_ARD, np.zeros(7) + 1e-3 * r, 0.01)
            h.set_data(X, om)
            assert h.compute() == 0
            ref.append(h.log_lik())
            hs.append(h)
        bad = [0]

        def worker(i):
            for _ in range(per):
                if hs[i].compute() != 0 or hs[i].log_lik() != ref[i]:
                    bad[0] += 1

        ths = [threading.Thread(target=worker, args=(i,)) for i in range(R)]
        t0 = time.perf_counter()
        [t.start() for t in ths]
        [t.join() for t in ths]
        dt = time.perf_counter() - t0
        reruns = sum(h.handover_reruns() + h.flow_retries() for h in hs)
        line.append(f"{R} in flight {R * per / dt:.0f}/s (wrong {bad[0]}, re-runs {reruns})")
        for h in hs:
            h.close()
    print(" | ".join(line), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        per = sys.argv[1] if len(sys.argv) > 1 else "100"
        for v in ("0", "1"):
            r = subprocess.run([sys.executable, __file__, "--child", per], env=dict(os.environ, GPE_FLOW_PARTITIONS=v), capture_output=True,
                               text=True, timeout=300)
            print(r.stdout.strip() or f"GPE_FLOW_PARTITIONS={v}: rc {r.returncode} {r.stderr[-1500:]}", flush=True)
