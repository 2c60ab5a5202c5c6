#!/usr/bin/env python3
"""Per-kernel durations out of a rocprofv3 --kernel-trace database (rocpd .db): python3 tools/prof_kernels.py <results.db> [substr ...]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
subs = sys.argv[2:]
rows = db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc").fetchall()
for n, c, a, mn, mx, tot in rows:
    if not subs or any(k in n for k in subs):
        print("%-72s n=%5d avg %8.1f us  min %7.1f  max %8.1f  total %9.1f us" % (n[:72], c, a / 1e3, mn / 1e3, mx / 1e3, tot / 1e3))
