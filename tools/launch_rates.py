#!/usr/bin/env python3
"""tools/launch_rates.py <rocprofv3 results.db> <LPMP_LAUNCH_LOG csv> [passes to average] — algorithmic GB/s of every launch
of a pass: the durations of a kernel trace laid beside the engine's per-launch byte counts (same order)."""
import csv, sqlite3, sys
import numpy as np
db, log = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(log)))
n = len(rows)
c = sqlite3.connect(db)
k = c.execute("select name, duration, grid_x from kernels where name like '%sweep_%' or name like '%chain_%' order by start").fetchall()
d = np.array([r[1] for r in k], float) / 1e3
passes = len(d) // n
use = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, passes - 4)
seg = d[(passes - use) * n - (len(d) - passes * n) * 0: (passes - use) * n + use * n] if len(d) == passes * n else d[-use * n:]
seg = seg.reshape(use, n).mean(0)
tot_b = tot_t = 0.0
print("launch level class records recv send  MB      us    GB/s  ns/record")
for i, r in enumerate(rows):
    b = float(r["bytes"]); t = seg[i]
    tot_b += b; tot_t += t
    print(f"{i:4d} {int(r['level']):4d} {int(r['kclass']):4d} {int(r['records']):8d} {int(r['receives']):9d} {int(r['sends']):9d} {b/1e6:8.1f} {t:7.1f} {b/t/1e3:7.0f} {t*1e3/max(1,int(r['records'])):8.2f}")
print(f"pass: {tot_b/1e9:.2f} GB in {tot_t/1e3:.3f} ms of kernel time = {tot_b/tot_t/1e3:.0f} GB/s")
