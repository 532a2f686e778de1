#!/bin/bash
# VALU-busy time of every kernel of ONE pair (PMC pass, kernels serialised): the issue-side floor of a pair
export TMPDIR=/tmp
root=$PWD; out=/tmp/valu_budget; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $out -o pmc -- python $root/bench.py --pmc-child --inflight 1 --no-cpu-baseline "$@" > $out/log 2>&1
cd $root
python - <<PY
import glob, sqlite3, collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for path in glob.glob("$out/**/*.db", recursive=True):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name_col = [c for c in cols if c in ("kernel_name", "name")][0]
    per = collections.defaultdict(float)
    for kn, cn, val, did in db.execute("select %s, counter_name, value, dispatch_id from counters_collection" % name_col):
        per[(kn, did, cn)] += val
    dur = {}
    for kn, t0, t1, did in db.execute("select name, start, end, dispatch_id from kernels"):
        dur[did] = (t1 - t0) / 1e6
    seen = set()
    for (kn, did, cn), v in per.items():
        k = kn.split("(")[0].replace("void ", "")
        a = agg[k]
        if cn == "SQ_ACTIVE_INST_VALU": a[1] += v * 4 / 1024 / 2.39e6     # ms of VALU-busy at 2.39 GHz, chip average
        if cn == "SQ_INSTS_VALU": a[3] += v
        if (kn, did) not in seen:
            seen.add((kn, did)); a[0] += 1; a[2] += dur.get(did, 0.0)
tot_busy = sum(a[1] for a in agg.values()); tot_dur = sum(a[2] for a in agg.values())
print("one pair: kernel time %.2f ms, VALU-busy %.2f ms (%.0f %%)" % (tot_dur, tot_busy, 100 * tot_busy / tot_dur))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-40s n %4d  kernel ms %7.3f  VALU-busy ms %7.3f (%.0f %%)" % (k[:40], a[0], a[2], a[1], 100 * a[1] / max(a[2], 1e-9)))
PY
