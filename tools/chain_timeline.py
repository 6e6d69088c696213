#!/usr/bin/env python3
"""One steady-state reverse step of the sampler as a timeline (start offset, duration, gap to the previous kernel's end,
name) from a rocprofv3 rocpd database of tools/chain_probe.py: the step that ends with the tenth-from-last reverse update.
usage: python3 tools/chain_timeline.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if "td_reverse_step_kernel" in r[0]]
a, b = ends[-11] + 1, ends[-10]
step = rows[a:b + 1]
t0, busy, last_end = rows[ends[-11]][2], 0, rows[ends[-11]][2]
print(f"{len(step)} kernels, wall {(step[-1][2] - t0) / 1e3:.1f} us (from the previous update's end)")
for name, s, e in step:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - last_end) / 1e3:7.1f}  {name.split('(')[0][:80]}")
    busy += e - s
    last_end = max(last_end, e)
print(f"sum of durations {busy / 1e3:.1f} us")
