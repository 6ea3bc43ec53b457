#!/usr/bin/env python
"""Timeline of the last evaluation in a rocprofv3 kernel-trace db: prints kernels with start offset, duration, stream."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, stream_id, grid_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_build" in r[0]]
ev = rows[idx[-2]:idx[-1]]
t0 = ev[0][1]
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for r in ev[:lim]:
    print(f"{(r[1]-t0)/1e3:9.1f} +{(r[2]-r[1])/1e3:7.1f} us  s{r[3]} g={r[4]:7d} {r[0].split('(')[0][:50]}")
