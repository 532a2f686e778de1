#!/bin/bash
# same-box A/B of option sets on another bench configuration: bash tests/tools/gpu_ab_cfg.sh config reps "optsA" "optsB" ...
cfg=$1; reps=$2; shift; shift
for i in $(seq $reps); do
  for o in "$@"; do
    args=""; for kv in $o; do args="$args --opt $kv"; done
    python -u bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 --config $cfg --steps 6 --warmup 2 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; r=d['roofline']
print('$cfg [%s]'.ljust(10) % '$o', 'value', d['value'], 'single', d['ms_single_pair'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'], 'skew alone', r['alone']['avg_launch_ms'], 'in flight', r['avg_launch_ms'])"
  done
done
