#!/bin/bash
# same-box A/B of compile-time variants: bash tests/tools/gpu_ab_build2.sh reps "EXTRA_A" "EXTRA_B" ...  (each: hipcc flags, "" = none; bench opts via $BENCH_OPTS)
reps=$1; shift
i=0
for e in "$@"; do
  i=$((i+1)); touch reconstruction_amd/csrc/k_refine.hip
  make -s -C reconstruction_amd/csrc EXTRA="$e" all 2>/dev/null || { echo "build failed: $e"; continue; }
  cp reconstruction_amd/librsm_mi355.so /tmp/variant_$i.so
done
for r in $(seq $reps); do
  i=0
  for e in "$@"; do
    i=$((i+1)); cp /tmp/variant_$i.so reconstruction_amd/librsm_mi355.so
    python -u bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 --steps 8 --warmup 2 $BENCH_OPTS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; r=d['roofline']
print('[%s]' % '$e', 'value', d['value'], 'single', d['ms_single_pair'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'], 'skew alone', r['alone']['avg_launch_ms'], 'in flight', r['avg_launch_ms'])"
  done
done
