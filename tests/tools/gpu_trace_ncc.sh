#!/bin/bash
# kernel trace of one pair: the NCC kernels' launches (duration, grid)
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/trace_ncc; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace -d $out -o tr -- python $root/bench.py --no-cpu-baseline --steps 1 --warmup 0 --inflight 1 "$@" > $out/stdout.log 2>&1
cd $root
python - <<PY
import sqlite3, glob
db=sqlite3.connect(glob.glob("$out/*.db")[0]); cur=db.cursor()
rows=list(cur.execute("select name, start, end from kernels order by start"))
half=len(rows)//2
for pat in ("k_ncc_dot4", "k_ncc_wide", "k_ncc_rowgemm", "k_ncc_exact", "k_ncc_sparse", "k_hl_interval"):
    seq=[round((e-s)/1000) for n,s,e in rows[:half] if pat in n]
    print(pat, seq, "sum us", sum(seq))
PY
