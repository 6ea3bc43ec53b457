#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 rocpd database (rocprofv3 --kernel-trace): name, calls, total, avg, min, max."""
import sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
import os
by_grid = os.environ.get("KSTATS_GRID")
sel = "name || ' g=' || grid_x" if by_grid else "name"
rows = c.execute(f"select {sel}, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by {sel} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':80s} {'calls':>7s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for r in rows:
    print(f"{r[0][:80]:80s} {r[1]:7d} {r[2]:11.1f} {r[3]:9.2f} {r[4]:9.2f} {r[5]:9.2f} {100*r[2]/tot:6.1f}")
