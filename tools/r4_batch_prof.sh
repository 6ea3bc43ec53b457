#!/bin/bash
# kernel trace of the batched launch sequence (8 x N = 2048) under two splits
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for tm in 2560 1536; do
  rm -rf /tmp/pb_$tm
  R4_BATCH8=1 GPE_BATCH_TAIL_MAX=$tm GPE_STOP_EVENT=0 rocprofv3 --kernel-trace -d /tmp/pb_$tm -o p -- python $root/tools/r4_ab.py batch > /tmp/pb_$tm.out 2>&1
  db=$(find /tmp/pb_$tm -name '*.db' | head -1)
  echo "##### GPE_BATCH_TAIL_MAX=$tm"; grep batch_compute /tmp/pb_$tm.out
  python $root/tools/kstats.py $db | head -14
  python - "$db" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_build" in r[0]]
lo, hi = idx[-2], idx[-1]
t0 = rows[lo][1]
print("last batch, kernel by kernel: start us | duration us | gap before | name grid")
prev = None
for r in rows[lo:hi]:
    gap = (r[1] - prev) / 1e3 if prev else 0.0
    print(f"  {(r[1]-t0)/1e3:8.1f} {(r[2]-r[1])/1e3:8.1f} {gap:7.1f}  {r[0].split('(')[0][:50]} g={r[3]}")
    prev = r[2]
PY
done
