#!/bin/bash
# A/B of one rsm_set_option on the same box: bash tests/tools/gpu_ab_opt.sh name=value [reps]
mkdir -p gpurun_out
opt=$1; reps=${2:-3}
for i in $(seq $reps); do
  for o in "" "--opt $opt"; do
    python -u bench.py --no-cpu-baseline --steps 10 --warmup 2 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('$o'.ljust(24), d['ms_per_step'], 'init', s['initial_match'], 'rematch', s['rematch'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'])"
  done
done
