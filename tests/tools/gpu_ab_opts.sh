#!/bin/bash
# same-box A/B of option sets: bash tests/tools/gpu_ab_opts.sh reps "optsA" "optsB" ...   (each a quoted list of name=value, "" = defaults)
reps=$1; shift
for i in $(seq $reps); do
  for o in "$@"; do
    args=""; for kv in $o; do args="$args --opt $kv"; done
    python -u bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 --steps 8 --warmup 2 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; r=d['roofline']
print('[%s]'.ljust(10) % '$o', 'value', d['value'], 'single', d['ms_single_pair'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'], 'skew alone', r['alone']['avg_launch_ms'], 'in flight', r['avg_launch_ms'])"
  done
done
