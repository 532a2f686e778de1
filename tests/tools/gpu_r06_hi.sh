#!/bin/bash
# A/B of bench options on one box: bash tests/tools/gpu_r06_hi.sh "" "--opt refine_skew_hi_priority=1" ...
run() { python -u bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 --steps 8 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], 'ms/pair', d['ms_per_pair'], 'single', d['ms_single_pair'], 'skew alone', r['alone']['avg_launch_ms'], 'in flight', r['avg_launch_ms'])"; }
for rep in 1 2; do for o in "$@"; do echo "[$o]"; run $o; done; done
