#!/bin/bash
# VALU counters of the dominant kernel for prebuilt libraries tests/_ab/<name>.so: bash tests/tools/gpu_r05_valu.sh name1 name2 ...
cp reconstruction_amd/librsm_mi355.so /tmp/keep.so
for n in "$@"; do
  cp tests/_ab/$n.so reconstruction_amd/librsm_mi355.so
  python -u bench.py --no-cpu-baseline --adapter-pairs 0 --steps 8 --warmup 2 $BENCH_OPTS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[$n]', 'value', d['value'], 'single', d['ms_single_pair'], 'skew alone', r['alone']['avg_launch_ms'], 'in flight', r['avg_launch_ms'], 'traffic', r['traffic'], json.dumps(r['valu']))"
done
cp /tmp/keep.so reconstruction_amd/librsm_mi355.so
