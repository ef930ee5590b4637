#!/usr/bin/env python3
"""tools/gap_report.py <rocpd .db> [min_gap_us] — idle gaps between consecutive kernel dispatches (any queue) of a
rocprofv3 --kernel-trace database, largest first, with the kernels either side."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rows = list(db.execute("select name, start, end from kernels order by start"))
gaps = []
busy_end = rows[0][2]
for i in range(1, len(rows)):
    g = (rows[i][1] - busy_end) / 1e3
    if g > thr:
        gaps.append((g, i))
    busy_end = max(busy_end, rows[i][2])
print("dispatches %d, span %.1f ms, gaps > %.0f us: %d totalling %.1f ms" % (
    len(rows), (rows[-1][2] - rows[0][1]) / 1e6, thr, len(gaps), sum(g for g, _ in gaps) / 1e3))
for g, i in sorted(gaps, reverse=True)[:25]:
    a = rows[i - 1][0].replace("(anonymous namespace)::", "")[:40]
    b = rows[i][0].replace("(anonymous namespace)::", "")[:40]
    print("%9.1f us at %9.1f ms  after %-40s before %s" % (g, (rows[i][1] - rows[0][1]) / 1e6, a, b))
