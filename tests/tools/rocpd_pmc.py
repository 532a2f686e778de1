#!/usr/bin/env python
"""Per-kernel PMC summary from rocprofv3 rocpd .db files: mean counter value per dispatch, grouped by kernel."""
import sqlite3
import sys
from collections import defaultdict


def main(paths, filt=None):
    agg = defaultdict(lambda: defaultdict(list))
    for path in paths:
        db = sqlite3.connect(path)
        cur = db.cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        # typical columns: dispatch_id, kernel name, counter_name, value (+ start/end)
        name_col = [c for c in cols if c in ("kernel_name", "name")][0]
        rows = cur.execute("select %s, counter_name, value, dispatch_id from counters_collection" % name_col)
        per = defaultdict(float)
        for kn, cn, val, did in rows:
            per[(kn, cn, did)] += val  # sum over dimensions (XCD / instances)
        for (kn, cn, did), v in per.items():
            agg[kn][cn].append(v)
    print("kernel,counter,dispatches,mean,total")
    for kn in sorted(agg, key=lambda k: -sum(sum(v) for v in agg[k].values())):
        if filt and filt not in kn:
            continue
        for cn, vals in sorted(agg[kn].items()):
            print('"%s",%s,%d,%.1f,%.1f' % (kn[:60], cn, len(vals), sum(vals) / len(vals), sum(vals)))


if __name__ == "__main__":
    args = sys.argv[1:]
    filt = None
    if args and args[0].startswith("--filter="):
        filt = args[0].split("=", 1)[1]
        args = args[1:]
    main(args, filt)
