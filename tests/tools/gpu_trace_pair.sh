#!/bin/bash
# kernel trace of ONE pair alone (bench.py --pmc-child): per (kernel, grid) launch count, total / average duration, and the
# idle time of the GPU between consecutive kernels; bash tests/tools/gpu_trace_pair.sh [bench options]
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/trace_pair; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace -d $out -o tr -- python $root/bench.py --pmc-child --inflight 1 --no-cpu-baseline "$@" > $out/stdout.log 2>&1
cd $root
python - <<PY
import sqlite3, glob
db=sqlite3.connect(glob.glob("$out/**/*.db", recursive=True)[0]); cur=db.cursor()
cols=[r[1] for r in cur.execute("pragma table_info(kernels)")]
gx=[c for c in cols if c.lower() in ("grid_x","grid_size_x","grid_size")]
q="select name, start, end, %s from kernels order by start" % (", ".join(c for c in cols if "grid" in c.lower()) or "0")
rows=list(cur.execute(q))
print("columns:", [c for c in cols if "grid" in c.lower() or "workgroup" in c.lower()])
t0=rows[0][1]; t1=max(r[2] for r in rows)
agg={}; busy=0; last_end=rows[0][1]; gaps=0
for r in rows:
    n=r[0].split("(")[0].replace("void ",""); key=(n, tuple(r[3:]))
    a=agg.setdefault(key,[0,0.0,r[1]]); a[0]+=1; a[1]+=(r[2]-r[1])/1e3
    if r[1]>last_end: gaps+=(r[1]-last_end)/1e3
    last_end=max(last_end,r[2])
print("span %.3f ms, kernels %d, sum of kernel time %.3f ms, idle gaps %.3f ms" % ((t1-t0)/1e6, len(rows), sum(a[1] for a in agg.values())/1e3, gaps/1e3))
for key,a in sorted(agg.items(), key=lambda kv: kv[1][2]):
    print("%-46s grid %-22s n %4d total %9.1f us avg %8.1f" % (key[0][:46], str(key[1]), a[0], a[1], a[1]/a[0]))
PY
rm -rf $out
