#!/bin/bash
# kernel-trace stats of ONE pair of a config: per-kernel totals (which kernels make up a stage)
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/trace_stage; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace -d $out -o tr -- python $root/bench.py --no-cpu-baseline --measure-traffic 0 --steps 1 --warmup 0 --inflight 1 "$@" > $out/stdout.log 2>&1
cd $root
python - <<PY
import sqlite3, glob
db=sqlite3.connect(glob.glob("$out/**/*.db", recursive=True)[0]); cur=db.cursor()
rows=list(cur.execute("select name, start, end from kernels order by start"))
# last pair run = after the last k_pyr_down burst; take kernels after the last 'k_find_margin' launch
last=max(i for i,(n,s,e) in enumerate(rows) if "find_margin" in n)
agg={}
for n,s,e in rows[last:]:
    k=n.split("(")[0].replace("void ","")
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=(e-s)/1e6
for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:28]:
    print("%-60s %5d %8.3f ms" % (k[:60], c, t))
PY
rm -rf $out
