#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (three pairs in flight): per-kernel totals per pair
export TMPDIR=/tmp
root=$PWD; tag=${1:-r04}; shift; out=$root/gpurun_out/stats_$tag; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python $root/bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 "$@" > $out/bench_under_rocprof.log 2>&1
cd $root
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats.csv
rm -rf $out/stats
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/${tag}_kernel_stats.csv")))
d = json.loads([l for l in open("$out/bench_under_rocprof.log").read().splitlines() if l.startswith("{")][-1])
# pairs run by the command: warmup + steps (3 in flight each) + the stage-split run + the single-pair runs
F = d["config"]["pairs_in_flight"]; npairs = (d["warmup"] + d["steps"]) * F + 1 + max(2, d["steps"] // 2) + 2
print("value", d["value"], "ms/pair", d["ms_per_pair"], "pairs in the profiled command", npairs)
tot = 0
for r in rows[:22]:
    t = float(r["TotalDurationNs"]) / 1e6
    tot += t
    print("%-60s calls/pair %7.1f  ms/pair %7.3f  avg us %8.1f" % (r["Name"][:60], int(r["Calls"]) / npairs, t / npairs, float(r["AverageNs"]) / 1e3))
print("sum of all kernel time per pair: %.2f ms" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / npairs))
PY
