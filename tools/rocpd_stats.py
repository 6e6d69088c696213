#!/usr/bin/env python3
"""Per-kernel statistics (calls, total/avg/min/max duration) from a rocprofv3 rocpd SQLite database.
usage: python tools/rocpd_stats.py <results.db> [> profiles/summary.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  "from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for name, n, tot, avg, mn, mx in rows:
    print(f'"{name}",{n},{tot},{avg:.0f},{mn},{mx},{100.0 * tot / total:.3f}')
