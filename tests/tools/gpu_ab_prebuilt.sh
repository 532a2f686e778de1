#!/bin/bash
# same-box A/B of the prebuilt variants (tests/tools/build_variants.sh): bash tests/tools/gpu_ab_prebuilt.sh reps   (bench opts via $BENCH_OPTS)
reps=$1
cp reconstruction_amd/librsm_mi355.so /tmp/shipped.so
for r in $(seq $reps); do
  for v in reconstruction_amd/variants/v_*.so; do
    cp $v reconstruction_amd/librsm_mi355.so
    python -u bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 --steps 8 --warmup 2 $BENCH_OPTS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; r=d['roofline']
print('[%s]' % open('${v%.so}.txt').read().strip(), 'value', d['value'], 'single', d['ms_single_pair'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'], 'skew alone', r['alone']['avg_launch_ms'], 'in flight', r['avg_launch_ms'])"
  done
done
cp /tmp/shipped.so reconstruction_amd/librsm_mi355.so
