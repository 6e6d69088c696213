#!/usr/bin/env python3
"""One steady-state optimizer step as a timeline (start offset, duration, gap to the previous kernel's end, name) from a
rocprofv3 rocpd database of tools/train_probe.py: the step that ends with the LAST td_clip_adam_ema_kernel but one.
usage: python3 tools/rocpd_timeline.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if "td_clip_adam_ema_kernel" in r[0]]
a, b = ends[-3] + 1, ends[-2]
step = rows[a:b + 1]
t0, busy, last_end = step[0][1], 0, step[0][1]
print(f"{len(step)} kernels, wall {(step[-1][2] - t0) / 1e3:.1f} us")
for name, s, e in step:
    gap = s - last_end
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  {name.split('(')[0][:70]}")
    busy += e - s
    last_end = max(last_end, e)
print(f"sum of durations {busy / 1e3:.1f} us")
