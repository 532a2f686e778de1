#!/bin/bash
# repeated A/B of bench options on one box: bash tests/tools/gpu_ab_rep.sh reps "optsA" "optsB" ...
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
reps=$1; shift
for i in $(seq $reps); do
  for o in "$@"; do
    python -u bench.py --no-cpu-baseline --measure-traffic 0 --steps 6 --warmup 2 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('$o'.ljust(60), d['value'], d['ms_per_pair'], 'single', d.get('ms_single_pair'), 'top', s['refine_sweep_top'], 'low', s['refine_sweep'])"
  done
done | tee gpurun_out/ab_rep.log
env | grep -i "rocp" | head
