#!/bin/bash
# bench the default configuration for several values of one option: bash tests/tools/gpu_sweep_opt.sh name v1 v2 ...
name=$1; shift
for v in "$@"; do
python -u bench.py --no-cpu-baseline --opt $name=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$name=$v', d['value'], d['ms_per_pair'], 'launch', r['avg_launch_ms'], 'alone', r['alone']['avg_launch_ms'])"
done
