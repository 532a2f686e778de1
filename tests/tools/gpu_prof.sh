#!/bin/bash
# rocprofv3 kernel-trace stats of one bench invocation -> gpurun_out/prof_<tag>/
tag=${1:-r01}; shift
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o $tag -- python bench.py --no-cpu-baseline "$@" > $out/bench_stdout.log 2>&1
echo "rocprof rc=$?" >> $out/bench_stdout.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f"
tail -2 $out/bench_stdout.log
# drop the big trace, keep the stats
find $out -name "*kernel_trace.csv" -size +20M -delete
