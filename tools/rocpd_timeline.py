#!/usr/bin/env python3
"""One steady-state optimizer step as a timeline (start offset, duration, gap to the previous kernel's end, name) from a
rocprofv3 rocpd database of tools/train_probe.py: the step that ends with the LAST optimizer kernel but one (one of the
probe's launch-by-launch steps), or with the one `back` launches from the end (back >= 8: a replayed graph of the timed loop).
usage: python3 tools/rocpd_timeline.py <results.db> [back]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if "td_clip_adam_ema" in r[0]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = ends[-back - 1] + 1, ends[-back]
step = rows[a:b + 1]
t0, busy, last_end = step[0][1], 0, step[0][1]
print(f"{len(step)} kernels, wall {(step[-1][2] - t0) / 1e3:.1f} us")
for name, s, e in step:
    gap = s - last_end
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  {name.split('(')[0][:70]}")
    busy += e - s
    last_end = max(last_end, e)
print(f"sum of durations {busy / 1e3:.1f} us")
