#!/usr/bin/env python
"""Kernels and memory copies of the LAST cloud-filter call in a rocprofv3 rocpd .db, in time order: start offset, duration, gap to the previous end."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = [(s, e, n) for n, s, e in cur.execute("select %s, start, end from kernels order by start" % name)]
try:
    mc = [r[1] for r in cur.execute("pragma table_info(memory_copies)")]
    if mc:
        nm = "name" if "name" in mc else None
        for r in cur.execute("select start, end%s from memory_copies" % (", " + nm if nm else "")):
            rows.append((r[0], r[1], "[copy] " + (r[2] if nm else "")))
except Exception as ex:
    pass
rows.sort()
first = max(i for i, r in enumerate(rows) if r[2].startswith("k_bbox"))
t0 = rows[first][0]; prev = t0
for s, e, n in rows[first:]:
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n[:70]))
    prev = max(prev, e)
print("total %.1f us" % ((prev - t0) / 1e3))
