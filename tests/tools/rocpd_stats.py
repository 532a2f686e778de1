#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 rocpd .db  ->  CSV on stdout.
Used to turn gpurun_out/prof_*/..._results.db into the summaries committed under profiles/."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(
        "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s "
        "order by 3 desc" % (name, name)))
    tot = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
    for n, c, t, a, mn, mx in rows:
        print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (n, c, t, a, 100.0 * t / tot, mn, mx))


if __name__ == "__main__":
    main(sys.argv[1])
