#!/bin/bash
# kernel trace of one C2 pair: per-sweep durations of the top-level refine kernels
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/trace_refine; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace -d $out -o tr -- python $root/bench.py --no-cpu-baseline --steps 1 --warmup 0 --inflight 1 "$@" > $out/stdout.log 2>&1
cd $root
python - <<PY
import sqlite3, glob
db=sqlite3.connect(glob.glob("$out/*.db")[0]); cur=db.cursor()
rows=list(cur.execute("select name, start, end from kernels order by start"))
for pat in ("k_refine_sweep<1>", "k_refine_work"):
    seq=[(e-s)/1000 for n,s,e in rows if pat in n]
    top=seq[:149] if "sweep" in pat else seq
    print(pat, len(seq), "first 40:", [round(x) for x in top[:40]], "every 10th:", [round(x) for x in top[40:149:10]], "sum ms", round(sum(top)/1000,2))
# gaps between consecutive top-level sweeps
sw=[(s,e) for n,s,e in rows if "k_refine_sweep<1>" in n][:149]
print("span of the 149 sweeps ms:", (sw[-1][1]-sw[0][0])/1e6)
PY
