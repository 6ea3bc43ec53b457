#!/usr/bin/env python
"""Per-kernel PMC summary of rocprofv3 --pmc rocpd databases: mean counter value per dispatch and
mean duration, grouped by kernel (and grid with KSTATS_GRID=1)."""
import os, sqlite3, sys
from collections import defaultdict
by_grid = os.environ.get("KSTATS_GRID")
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    grid = {r[0]: r[1] for r in c.execute("select dispatch_id, grid_x from kernels")} if by_grid else {}
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))
    for name, did, dur, cn, cv in c.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events"):
        key = name.split("(")[0][:70] + (f" g={grid.get(did)}" if by_grid else "")
        a = acc[key][cn]
        a[0] += 1; a[1] += cv; a[2] += dur
    print(f"== {db}")
    for k in sorted(acc, key=lambda k: -max(v[2] for v in acc[k].values())):
        parts = [f"{cn}: n={v[0]} mean={v[1]/v[0]:.4g} (dur {v[2]/v[0]/1e3:.1f} us)" for cn, v in acc[k].items()]
        print(f"  {k:72s} " + " | ".join(parts))
