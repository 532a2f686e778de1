#!/bin/bash
# the durations of every launch of the top level's time-skewed kernel in one pair alone, in launch order, for prebuilt
# libraries tests/_ab/<name>.so:  bash tests/tools/gpu_r06_trace.sh name1 name2 ...   (EXTRA: extra bench.py options)
export TMPDIR=/tmp RSM_AB_OLD_LIBRARY=1
root=$PWD
cp reconstruction_amd/librsm_mi355.so /tmp/keep.so
for n in "$@"; do
  [ -f tests/_ab/$n.so ] && cp tests/_ab/$n.so reconstruction_amd/librsm_mi355.so || cp /tmp/keep.so reconstruction_amd/librsm_mi355.so   # (a name without a file: the tree's library)
  out=/tmp/trace_r06; rm -rf $out; mkdir -p $out
  cd /tmp
  rocprofv3 --kernel-trace -d $out -o tr -- python $root/bench.py --pmc-child --inflight 1 --no-cpu-baseline --opt refine_split=0 $EXTRA > $out/log 2>&1
  cd $root
  python - "$n" $(find $out -name "*.db") <<'P'
import sqlite3, sys
name = sys.argv[1]
rows = []
for path in sys.argv[2:]:
    db = sqlite3.connect(path)
    try:
        for kn, t0, t1 in db.execute("select name, start, end from kernels order by start"):
            rows.append((kn, t0, t1))
    except Exception as e:
        print("no kernels table", e)
sk = [(t1 - t0) / 1e3 for kn, t0, t1 in rows if "k_refine_skew<4, 1" in kn]
sw = [(t1 - t0) / 1e3 for kn, t0, t1 in rows if "k_refine_sweep<1" in kn]
tot = sum(t1 - t0 for kn, t0, t1 in rows) / 1e6
print("[%s] all kernels %.2f ms; skew<4,1> launches %d, us each: %s" % (name, tot, len(sk), " ".join("%.0f" % v for v in sk)))
print("[%s] sweep<1> launches %d, us each: %s" % (name, len(sw), " ".join("%.0f" % v for v in sw)))
P
done
cp /tmp/keep.so reconstruction_amd/librsm_mi355.so
