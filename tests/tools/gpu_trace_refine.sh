#!/bin/bash
# kernel trace of one C2 pair: per-sweep durations of the top-level refine kernels
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/trace_refine; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace -d $out -o tr -- python $root/bench.py --no-cpu-baseline --steps 1 --warmup 0 --inflight 1 "$@" > $out/stdout.log 2>&1
cd $root
python - <<PY
import sqlite3, glob
db=sqlite3.connect(glob.glob("$out/**/*.db", recursive=True)[0]); cur=db.cursor()
rows=list(cur.execute("select name, start, end from kernels order by start"))
# the LAST pair run of the process (the stage-split run / single runs follow the timed one): take the final 200 top-level launches
top=[(n,s,e) for n,s,e in rows if "k_refine_sweep<1" in n or "k_refine_skew<" in n and ", 1>" in n or "k_refine_first" in n]
sw=[(n,s,e) for n,s,e in rows if "k_refine_sweep<1" in n]
sk=[(n,s,e) for n,s,e in rows if "k_refine_skew<" in n and ", 1>" in n]
per=int("$NSW")
print("single sweeps, last pair:", [round((e-s)/1000) for n,s,e in sw[-per:]])
print("skew launches, last pair:", [round((e-s)/1000) for n,s,e in sk[-int("$NSK"):]])
first=[(e-s)/1000 for n,s,e in rows if "k_refine_first" in n]
print("k_refine_first:", [round(x) for x in first[-5:]])
allk={}
lo=sw[-per][1]
hi=sk[-1][2] if sk else sw[-1][2]
for n,s,e in rows:
    if s>=lo and e<=hi:
        allk.setdefault(n.split("(")[0],[0,0]); allk[n.split("(")[0]][0]+=1; allk[n.split("(")[0]][1]+=(e-s)/1e6
print("top level span ms", (hi-lo)/1e6, {k:(v[0],round(v[1],3)) for k,v in allk.items()})
PY
rm -rf $out
