#!/usr/bin/env python
"""Launch trace of ONE gpe_batch_compute of G GPs of order N (config 4: G = 8, N = 2048) — the engine's own start/stop events of
every launch (gpe_trace), as tools/trace_eval.py.   python tools/trace_c4.py [G] [N]"""
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from limbo_amd import _capi, synth  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
eng = _capi.load_engine()
X, Y = synth.make_problem("c2", N=N)
om, _ = synth.obs_mean_data(Y)
hs = []
for g in range(G):
    h = _capi.Handle(eng, 0)
    h.set_data(X, om)
    h.set_kernel(synth.SE_ARD, np.zeros(7) + 1e-2 * g, 0.01)
    hs.append(h)
for _ in range(3):
    _capi.batch_compute(hs)
    [h.log_lik() for h in hs]
eng.fn("trace")(1)
_capi.batch_compute(hs)
lls = [h.log_lik() for h in hs]
path = tempfile.mktemp(suffix=".trace")
eng.fn("trace_dump")(path.encode())
eng.fn("trace")(0)
rows = []
for ln in open(path):
    a, b, sid, rest = ln.split(None, 3)
    name, grid, block = rest.rsplit(None, 2)
    rows.append((float(a), float(b), int(sid), name.strip("()"), grid, block))
rows.sort()
print(f"# tools/trace_c4.py {G} {N}: one gpe_batch_compute of {G} GPs of order {N}; {len(rows)} launches, first start -> last end "
      f"{max(r[1] for r in rows):.1f} us; log_lik[0] = {lls[0]:.9f}")
print("# start_us    end_us  dur_us stream kernel grid block")
for a, b, sid, name, grid, block in rows:
    print(f"{a:10.1f} {b:9.1f} {b - a:7.1f} {sid} {name} {grid} {block}")
