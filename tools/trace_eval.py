#!/usr/bin/env python
"""Launch trace of ONE evaluation in the PRODUCTION schedule (look-ahead on two streams, fused next-panel update + diagonal
block, head-tile hand-over): every launch carries its own start/stop events (gpe_trace, include/gpe.h) — no profiler, no
marker packets.  rocprofv3's kernel trace cannot show this schedule (it delays dispatches that carry a completion event).
    python tools/trace_eval.py [compute|hp] [N] > profiles/rNN_production_timeline.txt"""
import ctypes as C
import sys
import tempfile
from collections import defaultdict
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from limbo_amd import _capi, synth  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "compute"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    eng = _capi.load_engine()
    X, Y = synth.make_problem("c2", N=N)
    om, _ = synth.obs_mean_data(Y)
    h = _capi.Handle(eng)
    h.set_kernel(synth.SE_ARD, np.zeros(7), 0.01)
    h.set_data(X, om)

    def step(i):
        if what == "hp":
            h.hp_objective(synth.SE_ARD, np.zeros(7) + 1e-3 * i, 0.01, want_grad=True)
        else:
            h.compute()
            h.log_lik()

    for i in range(4):
        step(i)
    eng.fn("trace")(1)
    step(5)
    path = tempfile.mktemp(suffix=".trace")
    eng.fn("trace_dump")(path.encode())
    eng.fn("trace")(0)
    rows = []
    for ln in open(path):
        a, b, sid, rest = ln.split(None, 3)
        name, grid, block = rest.rsplit(None, 2)
        rows.append((float(a), float(b), int(sid), name.strip("()"), grid, block))
    rows.sort()
    t_end = max(r[1] for r in rows)
    print(f"# tools/trace_eval.py {what} {N}: one {'gpe_hp_objective (gradient)' if what == 'hp' else 'compute()+log_lik'} at N={N}, D=6, SE-ARD in the "
          f"production schedule; {len(rows)} launches, first start -> last end {t_end:.1f} us (device timestamps of the dispatches themselves)")
    busy = defaultdict(lambda: [0, 0.0])
    for a, b, sid, name, grid, block in rows:
        k = name.split("<")[0] + ("<" + name.split("<", 1)[1] if "<" in name else "")
        busy[(k, sid)][0] += 1
        busy[(k, sid)][1] += b - a
    print("# per kernel and stream: launches, busy us")
    for (k, sid), (n, t) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
        print(f"#   stream {sid}  n={n:3d}  busy={t:8.1f} us  {k}")
    # critical stream: the gaps between consecutive launches
    main_sid = rows[0][2]
    mrows = [r for r in rows if r[2] == main_sid]
    gaps = [mrows[i + 1][0] - mrows[i][1] for i in range(len(mrows) - 1)]
    print(f"# stream {main_sid} (the panel chain): {len(mrows)} launches, busy {sum(r[1] - r[0] for r in mrows):.1f} us, gaps between "
          f"consecutive launches: total {sum(gaps):.1f} us, mean {np.mean(gaps):.2f}, max {max(gaps):.1f}")
    print("# start_us    end_us  dur_us stream kernel grid block")
    for a, b, sid, name, grid, block in rows:
        print(f"{a:10.1f} {b:9.1f} {b - a:7.1f} {sid} {name} {grid} {block}")
    h.close()


if __name__ == "__main__":
    main()
