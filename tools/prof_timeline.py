#!/usr/bin/env python3
"""tools/prof_timeline.py <rocprofv3 results.db> [n_last] — per-dispatch timeline (duration, gap to the previous
dispatch, grid) of the last dispatches, and per-kernel totals, from a rocprofv3 --kernel-trace database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
print("== totals")
tot = {}
for name, s, e, g, w in rows:
    t = tot.setdefault(name, [0, 0])
    t[0] += 1
    t[1] += e - s
for name, (c, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print("%-70s calls %5d  total %9.1f us  avg %8.1f us" % (short, c, ns / 1e3, ns / 1e3 / c))
window = rows[-n_last:]
if len(sys.argv) > 3:
    # [n_last] [substring]: the window ends n_last // 3 dispatches after the last dispatch whose name contains it
    idx = [i for i, r in enumerate(rows) if sys.argv[3] in r[0]]
    if idx:
        hi = min(len(rows), idx[-1] + 1 + n_last // 3)
        window = rows[max(0, hi - n_last):hi]
print("== %d dispatches" % len(window))
prev = None
for name, s, e, g, w in window:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    print("%-44s dur %8.1f us  gap %6.1f us  wgs %5d x %4d" % (short, (e - s) / 1e3, ((s - prev) / 1e3 if prev else 0.0), g // max(w, 1), w))
    prev = e
