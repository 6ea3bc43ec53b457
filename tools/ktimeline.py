#!/usr/bin/env python
"""Timeline summary of the last N kernels of a rocprofv3 rocpd database: busy time, gaps between consecutive kernels."""
import sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# find the last evaluation: from the last k_build (kernel-matrix build) to the end
idx = [i for i, r in enumerate(rows) if r[0].startswith("void k_build") or r[0].startswith("k_build")]
lo = idx[-2] if len(idx) >= 2 else 0
hi = idx[-1]
ev = rows[lo:hi]
span = (ev[-1][2] - ev[0][1]) / 1e3
busy = sum(r[2] - r[1] for r in ev) / 1e3
gaps = [(ev[i + 1][1] - ev[i][2]) / 1e3 for i in range(len(ev) - 1)]
print(f"kernels {len(ev)}  span {span:.1f} us  busy {busy:.1f} us  gaps total {sum(gaps):.1f} us  mean gap {sum(gaps)/len(gaps):.2f} us  max gap {max(gaps):.2f}")
from collections import defaultdict
d = defaultdict(lambda: [0, 0.0, 0.0])
for i, r in enumerate(ev):
    k = r[0].split("(")[0][:60]
    d[k][0] += 1
    d[k][1] += (r[2] - r[1]) / 1e3
    if i + 1 < len(ev):
        d[k][2] += gaps[i]
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:60s} n={v[0]:4d} busy={v[1]:8.1f} us  gap_after={v[2]:7.1f} us")
