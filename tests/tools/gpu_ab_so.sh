#!/bin/bash
# same-box A/B of prebuilt libraries tests/_ab/<name>.so: bash tests/tools/gpu_ab_so.sh reps name1 name2 ...
reps=$1; shift
cp reconstruction_amd/librsm_mi355.so /tmp/keep.so
for r in $(seq $reps); do for n in "$@"; do
  cp tests/_ab/$n.so reconstruction_amd/librsm_mi355.so
  python -u bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 --steps 8 --warmup 2 $BENCH_OPTS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; r=d['roofline']
print('[$n]', 'value', d['value'], 'single', d['ms_single_pair'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'], 'skew alone', r['alone']['avg_launch_ms'], 'in flight', r['avg_launch_ms'])"
done; done
cp /tmp/keep.so reconstruction_amd/librsm_mi355.so
